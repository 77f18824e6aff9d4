"""The zero-knowledge private-batch circuit at several leaf counts on one GPU: circuit size, build time, and commit + s1 + prove
time of one batch (PrivateBatchProver over the restated leaf circuit; N - 2 real spends of one block + padding).
usage: python tools/private_batch_sizes.py [N ...]"""
import json, sys, time
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import __graft_entry__ as ge
import leaf_cases as lc
pkg = ge.load_package()
gpu = pkg.QpGpu(0)
L, R, A = pkg.leaf, pkg.recursion, pkg.aggregation
leaf = L.LeafCircuit()
lp = L.LeafProver(pkg, gpu, leaf)
for N in [int(a) for a in sys.argv[1:]] or [2, 4, 8, 16]:
    spends = lc.shared_tree_inputs(L, max(1, N - 2), depth=3, seed=N)
    proofs = [lp.prove(x)[0] for x in spends]
    t = time.perf_counter()
    pb = R.PrivateBatchProver(pkg, gpu, leaf, N, leaf_prover=lp)
    tb = time.perf_counter() - t
    pb.commit(proofs, seed=bytes(32)).prove()
    gpu.sync()
    t = time.perf_counter()
    proof = pb.commit(proofs, seed=bytes([1] * 32)).prove()
    dt = time.perf_counter() - t
    ok = pb.verifier.verify(proof)
    hdr, slots, nulls = A.parse_private_batch_public_inputs(A.proof_public_inputs(proof, A.private_batch_pi_len(N)))
    print(json.dumps({"N": N, "degree_bits": pb.circuit.info["degree_bits"], "rows_before_padding": pb.circuit.info["rows_before_padding"], "rows_blinding": pb.circuit.info["rows_blinding"],
                      "build_s": round(tb, 2), "commit_s1_prove_ms": round(dt * 1e3, 1), "proof_bytes": len(proof), "verified": bool(ok),
                      "paid_exit_slots": sum(1 for s in slots if s[0]), "nullifiers": len(set(nulls))}))
    pb.close()
