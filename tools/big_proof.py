#!/usr/bin/env python3
"""One proof at a larger degree (private-batch sized traces), GPU vs oracle bytes + verify. Usage: big_proof.py d"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
import oracle_binding
pkg = ge.load_package(); orc = oracle_binding.Oracle()
d = int(sys.argv[1])
pack, wires, pis = pkg.synth_circuit(d, seed=77)
gpu = pkg.QpGpu(0); circ = pkg.Circuit(gpu, pack)
t0 = time.perf_counter(); proof = circ.prove(wires, pis); t1 = time.perf_counter()
dw = gpu.to_device(wires)
circ.prove_dev(dw, pis); t2 = time.perf_counter()
for _ in range(3): circ.prove_dev(dw, pis)
t3 = time.perf_counter()
oc = oracle_binding.OracleCircuit(orc, pack)
t4 = time.perf_counter(); want = oc.prove(wires, pis); t5 = time.perf_counter()
print(f"d={d} proof {len(proof)} B; gpu first {1e3*(t1-t0):.1f} ms, steady {1e3*(t3-t2)/3:.2f} ms; oracle {t5-t4:.2f} s; equal={proof==want} verify={oc.verify(proof)}")
