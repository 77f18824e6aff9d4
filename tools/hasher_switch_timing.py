import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
import oracle_binding
from test_hasher_plug import placeholder_params
orc = oracle_binding.Oracle()
gpu = pkg.QpGpu(0)
pack, wires, pis = pkg.synth_circuit(8, num_wires=24, num_routed=16, num_public_inputs=1, seed=83)
def run(tag):
    t = time.perf_counter(); circ = pkg.Circuit(gpu, pack); t1 = time.perf_counter()
    p = circ.prove(wires, pis); t2 = time.perf_counter()
    p = circ.prove(wires, pis); t3 = time.perf_counter()
    oc = oracle_binding.OracleCircuit(orc, pack); t4 = time.perf_counter()
    q = oc.prove(wires, pis); t5 = time.perf_counter()
    print(f"{tag}: load {t1-t:.3f} prove1 {t2-t1:.3f} prove2 {t3-t2:.3f} | oracle load {t4-t3:.3f} prove {t5-t4:.3f} equal {p==q}", flush=True)
    circ.close(); oc.close()
run("poseidon")
prm = placeholder_params()
pkg.set_hasher_poseidon2(*prm); orc.select_poseidon2(*prm)
run("poseidon2")
pkg.set_hasher_poseidon(); orc.select_poseidon()
run("poseidon again")
