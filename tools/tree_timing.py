#!/usr/bin/env python3
"""Where the 64-leaf aggregation tree's time goes on one GPU: per level, admission checks + verification / witness / proving."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
agg = pkg.aggregation
gpu = pkg.QpGpu(0)
rec = dict(poseidon=True, base_sum=True, ext_arith=True, recursion=True)
d, db = 13, 16
tleaf = pkg.synth_circuit(d, num_wires=135, num_routed=80, num_public_inputs=21, seed=1000, poseidon=True, base_sum=True)
tpriv = pkg.synth_circuit(db, num_wires=135, num_routed=60, num_public_inputs=176, seed=78, **rec); tpriv[0][14] = 1
tpub = pkg.synth_circuit(db, num_wires=135, num_routed=80, num_public_inputs=agg.public_batch_pi_len(8, 8), seed=77, **rec)
t = agg.AggregationTree(pkg, gpu, 0, 1, tleaf, tpriv, tpub, leaf_batch=64)
t.run(); t.run()
print("levels", t.times)
def clock(label, fn, reps=3):
    gpu.sync(); t0 = time.perf_counter()
    for _ in range(reps): r = fn()
    gpu.sync(); print(f"{label:44s} {(time.perf_counter() - t0) / reps * 1e3:8.2f} ms"); return r
leaves, batches, root = t.run()
lp = t.leaf
clock("leaf commit_many(32) [witness gen]", lambda: lp.commit_many([agg.leaf_public_inputs(i) for i in range(32)]))
clock("leaf prove_many(32)", lambda: (lp.commit_many([agg.leaf_public_inputs(i) for i in range(32)]), lp.prove_many())[1])
clock("leaf verify_many(64)", lambda: t.verifiers["leaf"].verify_many(leaves))
pp = t.private
clock("private batch_public_inputs x8 (checks+verify)", lambda: [pp.batch_public_inputs(leaves[8 * b:8 * b + 8]) for b in range(8)])
clock("private commit_many(8) (checks+verify+witness)", lambda: pp.commit_many([leaves[8 * b:8 * b + 8] for b in range(8)]))
clock("private commit_many + prove_many(8)", lambda: (pp.commit_many([leaves[8 * b:8 * b + 8] for b in range(8)]), pp.prove_many())[1])
pb = t.public
clock("public batch_public_inputs (checks+verify 8)", lambda: pb.batch_public_inputs(batches))
clock("public commit (checks+verify+witness)", lambda: pb.commit(batches))
clock("public commit + prove", lambda: (pb.commit(batches), pb.prove())[1])
clock("public verify root", lambda: t.verifiers["public"].verify_many([root]))
t.close(); gpu.close()
