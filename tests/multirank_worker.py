"""One rank of the multi-rank GPU tests (tests/test_multirank_gpu.py starts WORLD_SIZE fresh interpreters of this file).

BASELINE configs[3]: 8 independent leaf proofs sharded over the ranks, proof bytes gathered; configs[4]: the 64-leaf
aggregation tree at small degrees. Every proof is produced by the HIP path through the C ABI and compared byte for byte
with the CPU oracle's proof of the same circuit, witness and public inputs (the oracle is the checker only).
Rehearsal backend: gloo, so two ranks can share the one GPU of a test box; bench.py uses RCCL for the same calls.

usage: multirank_worker.py {leaves|tree} <result.json>   (RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT from the env)
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    mode, out_path = sys.argv[1], sys.argv[2]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    import oracle_binding
    pkg = ge.load_package()
    agg = pkg.aggregation
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ndev = torch.cuda.device_count()
    gpu = pkg.QpGpu(rank % max(ndev, 1))
    orc = oracle_binding.Oracle()
    res = {"rank": rank, "ok": False}

    def oracle_equal(pack, wires, pis, proof, seed=0):
        oc = oracle_binding.OracleCircuit(orc, pack)
        try:
            return proof == oc.prove(wires, pis, seed) and oc.verify(proof) == 0
        finally:
            oc.close()

    if mode == "leaves":
        # ---- configs[3]: 8 proofs with different witnesses, proof i on rank i % ... (contiguous shards), gathered ----
        d = int(os.environ.get("QP_TEST_LEAF_BITS", "10"))
        pack, wires, _ = pkg.synth_circuit(d, num_wires=135, num_routed=80, num_public_inputs=21, seed=4242, poseidon=True, base_sum=True)
        lo, hi = pkg.sharding.shard_range(8, world, rank)
        prover = agg.TemplateProver(gpu, pack, wires)
        mine, checks = [], []
        for i in range(lo, hi):
            pis = prover.commit(agg.leaf_public_inputs(i))
            w = prover.witness()
            proof = prover.prove()
            checks.append(oracle_equal(pack, w, pis, proof))
            mine.append(proof)
        gathered = pkg.sharding.gather_proof_bytes(mine, dist)
        flat = [p for r in gathered for p in r]
        ok = all(checks) and len(flat) == 8 and [len(r) for r in gathered] == [pkg.sharding.shard_range(8, world, r)[1] - pkg.sharding.shard_range(8, world, r)[0] for r in range(world)]
        if rank == 0:
            # rank 0 rebuilds every proof independently: witness on its own GPU, proof by the oracle
            for i in range(8):
                pis = prover.commit(agg.leaf_public_inputs(i))
                ok = ok and oracle_equal(pack, prover.witness(), pis, flat[i])
                ok = ok and np.array_equal(agg.proof_public_inputs(flat[i], 21), agg.leaf_public_inputs(i))
            ok = ok and len(set(flat)) == 8                      # different witnesses, different proofs
        prover.close()
        res.update(ok=bool(ok), proofs=len(flat), proof_bytes=len(flat[0]))
    elif mode == "tree":
        # ---- configs[4]: 64 leaves -> 8 zero-knowledge private batches -> 1 public batch, small degrees ----
        dl, db = int(os.environ.get("QP_TEST_LEAF_BITS", "7")), int(os.environ.get("QP_TEST_BATCH_BITS", "8"))
        rec = dict(poseidon=True, base_sum=True, ext_arith=True, recursion=True)
        leaf = pkg.synth_circuit(dl, num_wires=135, num_routed=80, num_public_inputs=21, seed=11, poseidon=True, base_sum=True, poseidon2=True)
        priv = pkg.synth_circuit(db, num_wires=135, num_routed=60, num_public_inputs=21 * 8 + 8, seed=12, **rec)
        priv[0][14] = 1                                          # standard_recursion_zk_config: salted leaves
        pub = pkg.synth_circuit(db, num_wires=135, num_routed=80, num_public_inputs=agg.public_batch_pi_len(8, 8), seed=13, **rec)
        tree = agg.AggregationTree(pkg, gpu, rank, world, leaf, priv, pub)
        keep = {}
        SEED = 7000
        ADDRESS = b"".join(v.to_bytes(8, "little") for v in (0xA661, 2, 3, 4))
        exchange = os.environ.get("QP_TEST_EXCHANGE", "all")     # "root": proof bytes travel to the consuming rank only
        leaves, batches, root = tree.run(dist, None, blinding_seed=SEED, keep=keep, shuffle_seed=bytes(range(32)), aggregator_address=ADDRESS, exchange=exchange)
        ok = len(leaves) == 64 and len(batches) == 8
        if exchange == "root" and world > 1:
            mine = tree.plan["ranks"][rank]
            ok = ok and all((leaves[i] is not None) == (i in mine["leaves"]) for i in range(64))
            ok = ok and all((batches[b] is not None) == (rank == tree.plan["root"] or b in mine["private_batches"]) for b in range(8))
        # every proof this rank produced, against the oracle
        for (i, pis, w) in keep["leaf"]:
            ok = ok and oracle_equal(leaf[0], w, pis, leaves[i])
        for (b, pis, w) in keep["private"]:
            ok = ok and oracle_equal(priv[0], w, pis, batches[b], SEED + b)
        if rank == tree.plan["root"]:
            (_, pis, w), = keep["public"]
            ok = ok and oracle_equal(pub[0], w, pis, root)
            # the root's public inputs are the public-batch circuit's: aggregator address, the common block, and every batch
            # proof's exit slots and nullifiers forwarded in rank order; a batch's are its 8 leaves' (any slot order):
            # amounts merged per exit account, nullifiers sorted
            rp = agg.proof_public_inputs(root, pis.size)
            nb = 21 * 8 + 8
            ok = ok and rp[:4].tolist() == [0xA661, 2, 3, 4] and tuple(rp[6:10].tolist()) == agg.TEST_BLOCK_HASH and int(rp[10]) == 42 and int(rp[11]) == 128
            for b in range(8):
                bp = agg.proof_public_inputs(batches[b], nb)
                ok = ok and int(bp[0]) == 16 and tuple(bp[3:7].tolist()) == agg.TEST_BLOCK_HASH
                ok = ok and np.array_equal(rp[12 + 80 * b:12 + 80 * (b + 1)], bp[8:88])
                ok = ok and np.array_equal(rp[12 + 640 + 32 * b:12 + 640 + 32 * (b + 1)], bp[88:120])
                want, nulls = {}, []
                for j in range(8):
                    lp = agg.leaf_public_inputs(8 * b + j)
                    if leaves[8 * b + j] is not None:
                        ok = ok and np.array_equal(agg.proof_public_inputs(leaves[8 * b + j], 21), lp)
                    nulls.append(tuple(lp[4:8].tolist()))
                    for acct, amt in ((tuple(lp[8:12].tolist()), int(lp[1])), (tuple(lp[12:16].tolist()), int(lp[2]))):
                        want[acct] = want.get(acct, 0) + amt
                got = {tuple(bp[9 + 5 * k:13 + 5 * k].tolist()): int(bp[8 + 5 * k]) for k in range(16) if int(bp[8 + 5 * k])}
                ok = ok and got == {a: v for a, v in want.items() if v}
                ok = ok and [tuple(bp[88 + 4 * k:92 + 4 * k].tolist()) for k in range(8)] == sorted(nulls)
        else:
            ok = ok and root is None
        tree.close()
        res.update(ok=bool(ok), leaves=len(leaves), batches=len(batches), root_bytes=len(root) if root else 0)
    elif mode == "attest":
        # ---- configs[4] with circuits that check their inner proofs: 8 leaves (restated leaf circuit, from CircuitInputs) -> 4
        # first-level wrappers of 2 -> 1 second-level wrapper, batches round-robin over the ranks, first-level proofs to rank 0 ----
        import leaf_cases as lc
        L = pkg.leaf
        # (each level with its layer's own logic: private batch over the leaves, public batch over the first level)
        tree = pkg.recursion.AttestingTree(pkg, gpu, per_batch=2, batches=4, rank=rank, world=world, aggregator_address=bytes([6] * 32))
        sp = lc.shared_tree_inputs(L, 5, seed=60)
        dm = lc.dummy_inputs(L)
        xs = [sp[0], sp[1], dm, sp[2], sp[3], dm, dm, sp[4]]
        leaves, level1, root = tree.run(xs, dist, None)
        ok = len(leaves) == 2 * len(tree.my_batches) and all(tree.leaf_ver.verify(p) for p in leaves)
        if rank == 0:
            ok = ok and len(level1) == 4 and all(tree.w1_ver.verify(p) for p in level1) and tree.w2_ver.verify(root)
            oc = oracle_binding.OracleCircuit(orc, tree.w2.pack)
            ok = ok and oc.verify(root) == 0
            oc.close()
            # the root carries the PublicBatchPublicInputs of the eight leaves, whichever rank proved which batch
            want = tree.expected_root_public_inputs(np.stack([tree.leaf.commit(x)[2] for x in xs]))
            ok = ok and np.array_equal(np.frombuffer(root[-8 * want.size:], dtype=np.uint64), want)
            hdr, slots, nulls = pkg.aggregation.parse_public_batch_public_inputs(want, 4, 2)
            ok = ok and hdr["total_exit_slots"] == 16 and sum(s[0] for s in slots) == 2 * 297 + 2 * 297 + 297 and len(set(nulls)) == 8
        else:
            ok = ok and root is None
        tree.close()
        res.update(ok=bool(ok), root_bytes=len(root) if root else 0)
    else:
        raise SystemExit("unknown mode " + mode)
    gpu.close()
    dist.barrier()
    dist.destroy_process_group()
    with open(out_path + ".%d" % rank, "w") as f:
        json.dump(res, f)
    if not res["ok"]:
        raise SystemExit(3)


if __name__ == "__main__":
    main()
