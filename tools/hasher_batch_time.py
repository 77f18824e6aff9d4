#!/usr/bin/env python3
"""One worker, lockstep batch of 32 leaf-profile proofs (the bench shape) under either proof-system hasher: wall time per batch
and the proofs/s one worker delivers. The fork's proof-system hasher cannot be told from the reference, so the
Poseidon2 figure is the headline's counterpart should it turn out to be Poseidon2 (qp-poseidon-core's parameters).
usage: hasher_batch_time.py poseidon|poseidon2 [reps] [--check]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
kind = sys.argv[1] if len(sys.argv) > 1 else "poseidon"
reps = int(sys.argv[2]) if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else 5
check = "--check" in sys.argv
B = 32
qp = pkg.poseidon2_qp_params()
if kind == "poseidon2":
    pkg.set_hasher_poseidon2(*qp)     # the synthetic generator hashes public inputs and the circuit digest with the process default
gpu = pkg.QpGpu(0, hasher=qp if kind == "poseidon2" else None)
pack, wires, pis = pkg.synth_circuit(13, num_wires=135, num_routed=80, num_public_inputs=21, seed=1000, poseidon=True, base_sum=True, poseidon2=True)
d_w = gpu.to_device(wires)
cb = pkg.Circuit(gpu, pack, max_batch=B)
ptrs = [d_w.ptr] * B
proofs = cb.prove_batch_dev(ptrs, [pis] * B)
gpu.sync()
t0 = time.perf_counter()
for _ in range(reps):
    cb.prove_batch_dev(ptrs, [pis] * B)
gpu.sync()
ms = (time.perf_counter() - t0) / reps * 1e3
out = {"hasher": kind, "batch": B, "batch_ms": round(ms, 2), "proofs_per_s_one_worker": round(B / ms * 1e3, 1), "proof_bytes": len(proofs[0]),
       "mx": os.environ.get("QPGPU_MX", "1")}
if check:
    import oracle_binding
    orc = oracle_binding.Oracle()
    if kind == "poseidon2":
        orc.select_poseidon2(*qp)
    oc = oracle_binding.OracleCircuit(orc, pack)
    out["bytes_equal_oracle"] = bool(oc.prove(wires, pis) == proofs[0] and all(p == proofs[0] for p in proofs))
    oc.close()
print(json.dumps(out))
cb.close(); gpu.close()
if kind == "poseidon2":
    pkg.set_hasher_poseidon()
