#!/usr/bin/env python3
"""One proof at a larger degree (private-batch sized traces): GPU timing, the restated verifier's verdict and, with
--bytes, the byte comparison against the CPU restatement's prover (slow above 2^16 rows).
Usage: big_proof.py d [--routed R] [--zk] [--bytes]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
import oracle_binding
ap = argparse.ArgumentParser()
ap.add_argument("d", type=int); ap.add_argument("--routed", type=int, default=80)
ap.add_argument("--zk", action="store_true"); ap.add_argument("--bytes", action="store_true")
a = ap.parse_args()
pkg = ge.load_package(); orc = oracle_binding.Oracle()
t = time.perf_counter()
pack, wires, pis = pkg.synth_circuit(a.d, num_routed=a.routed, seed=77, poseidon=True, base_sum=True, ext_arith=True, recursion=True)
if a.zk:
    pack[14] = 1
print(f"synthetic circuit: {time.perf_counter()-t:.1f} s, pack {pack.nbytes/2**20:.0f} MiB, witness {wires.nbytes/2**20:.0f} MiB", flush=True)
gpu = pkg.QpGpu(0)
t = time.perf_counter(); circ = pkg.Circuit(gpu, pack); print(f"circuit load (constants/sigmas commitment, workspace): {time.perf_counter()-t:.2f} s", flush=True)
dw = gpu.to_device(wires)
circ.set_blinding_seed(5)
t0 = time.perf_counter(); proof = circ.prove_dev(dw, pis); t1 = time.perf_counter()
for _ in range(3):
    circ.set_blinding_seed(5); circ.prove_dev(dw, pis)
t2 = time.perf_counter()
print(f"d={a.d} proof {len(proof)} B; gpu first {1e3*(t1-t0):.1f} ms, steady {1e3*(t2-t1)/3:.2f} ms", flush=True)
t = time.perf_counter(); oc = oracle_binding.OracleCircuit(orc, pack); print(f"oracle circuit load: {time.perf_counter()-t:.1f} s", flush=True)
t = time.perf_counter(); v = oc.verify(proof); print(f"restated verifier: code {v} in {time.perf_counter()-t:.2f} s", flush=True)
bad = bytearray(proof); bad[len(bad) // 3] ^= 1
print("one flipped bit rejected:", oc.verify(bytes(bad)) != 0, flush=True)
if a.bytes:
    t = time.perf_counter(); want = oc.prove(wires, pis, seed=5); print(f"oracle prover {time.perf_counter()-t:.1f} s; bytes equal: {want == proof}", flush=True)
