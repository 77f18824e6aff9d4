"""The N>1 path on CPU: world_size-2 gloo processes shard a set of proofs, gather the proof bytes the way
bench.py --gpus N does (same function), and agree on the aggregation-tree schedule."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import __graft_entry__ as ge


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = ge.load_package()
    sh = pkg.sharding
    lo, hi = sh.shard_range(5, world, rank)                 # 5 proofs over 2 ranks: 3 + 2
    rng = np.random.default_rng(100)
    all_proofs = [rng.integers(0, 256, 1000 + 37 * i, dtype=np.uint8).tobytes() for i in range(5)]
    mine = all_proofs[lo:hi]
    gathered = sh.gather_proof_bytes(mine, dist)
    flat = [p for r in gathered for p in r]
    ok = flat == all_proofs and [len(r) for r in gathered] == [3, 2]
    # a rank with nothing to contribute still takes part
    g2 = sh.gather_proof_bytes(mine if rank == 0 else [], dist)
    ok = ok and g2 == [all_proofs[0:3], []] if rank == 0 else ok and g2[1] == []
    # cached layout: metadata exchanged on the first call only; a changed size is refused rather than mis-sliced
    layout = {}
    for _ in range(3):
        ok = ok and [p for r in sh.gather_proof_bytes(mine, dist, None, layout) for p in r] == all_proofs
    ok = ok and layout["counts"] == [3, 2]
    try:   # only rank 0 breaks the layout: both ranks must raise, neither may hang in the collective
        sh.gather_proof_bytes(([mine[0] + b"x"] + mine[1:]) if rank == 0 else mine, dist, None, dict(layout))
        ok = False
    except ValueError:
        pass
    ok = ok and [p for r in sh.gather_proof_bytes(mine, dist, None, layout) for p in r] == all_proofs   # still usable
    # the bench's per-step exchange: fixed-size proofs written into the sender's block in place, one collective, no decoding
    count, plen = 4, 777
    blk = sh.ProofBlockGather(count, plen, dist, torch.device("cpu"), blocks=3)
    want = [[np.random.default_rng(50 * r + i).integers(0, 256, plen, dtype=np.uint8) for i in range(count)] for r in range(world)]
    for step in range(5):                                   # ring of 3 blocks reused
        b = step % 3
        for i in range(count):
            blk.slot(b, i)[:] = want[rank][i] ^ np.uint8(step)
        got = blk.gather(b)
        ok = ok and tuple(got.shape) == (world, count, plen)
        ok = ok and all(np.array_equal(got[r, i].numpy(), want[r][i] ^ np.uint8(step)) for r in range(world) for i in range(count))
    # gather to the consuming rank only: the root gets everything, the others nothing
    g3 = sh.gather_proof_bytes(mine, dist, root=1)
    ok = ok and ((g3 is None) if rank != 1 else [p for r in g3 for p in r] == all_proofs)
    lay3 = {}
    for _ in range(2):
        g3 = sh.gather_proof_bytes(mine, dist, None, lay3, root=0)
        ok = ok and ((g3 is None) if rank != 0 else [len(r) for r in g3] == [3, 2] and [p for r in g3 for p in r] == all_proofs)
    blk = sh.ProofBlockGather(count, plen, dist, torch.device("cpu"), blocks=2, root=0)
    for step in range(3):
        b = step % 2
        for i in range(count):
            blk.slot(b, i)[:] = want[rank][i] ^ np.uint8(7 + step)
        got = blk.gather(b)
        if rank == 0:
            ok = ok and tuple(got.shape) == (world, count, plen)
            ok = ok and all(np.array_equal(got[r, i].numpy(), want[r][i] ^ np.uint8(7 + step)) for r in range(world) for i in range(count))
        else:
            ok = ok and got is None
    ret[rank] = ok
    dist.barrier()
    dist.destroy_process_group()


def test_gather_two_ranks_gloo():
    world, port = 2, _free_port()
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert ret[0] and ret[1]


def test_schedule_and_shards(pkg):
    sh = pkg.sharding
    assert [sh.shard_range(64, 8, r) for r in range(8)] == [(8 * r, 8 * r + 8) for r in range(8)]
    assert [sh.shard_range(5, 2, r) for r in range(2)] == [(0, 3), (3, 5)]
    assert sh.shard_range(3, 8, 7) == (3, 3)
    plan = sh.aggregation_schedule(64, 8, 8)              # BASELINE config 5
    assert plan["root"] == 0
    leaves = sorted(l for r in plan["ranks"].values() for l in r["leaves"])
    assert leaves == list(range(64))
    assert all(len(r["leaves"]) == 8 and len(r["private_batches"]) == 1 for r in plan["ranks"].values())
    plan2 = sh.aggregation_schedule(64, 8, 2)
    assert [len(plan2["ranks"][r]["private_batches"]) for r in range(2)] == [4, 4]
    # single process: gather is the identity
    assert sh.gather_proof_bytes([b"ab", b"c"]) == [[b"ab", b"c"]]
    one = sh.ProofBlockGather(2, 5, None, torch.device("cpu"))
    one.slot(0, 0)[:] = 7; one.slot(0, 1)[:] = 9
    assert one.gather(0).tolist() == [[[7] * 5, [9] * 5]]


def _tree_worker(rank, world, port, ret):
    """The aggregation-tree dataflow of bench.py's `aggregation_tree` leg with stand-in proofs (sha256 of the inputs):
    leaves per rank -> gather -> private batches per rank -> gather -> root proves the public batch."""
    import hashlib
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sh = ge.load_package().sharding
    plan = sh.aggregation_schedule(64, 8, world)
    mine = plan["ranks"][rank]
    leaf = lambda i: hashlib.sha256(b"leaf%d" % i).digest() * 4
    g0 = sh.gather_proof_bytes([leaf(i) for i in mine["leaves"]], dist)
    all_leaves = [p for r in g0 for p in r]
    ok = all_leaves == [leaf(i) for i in range(64)]                       # rank order = leaf order
    batch = lambda b: hashlib.sha256(b"".join(all_leaves[8 * b:8 * b + 8])).digest() * 5
    g1 = sh.gather_proof_bytes([batch(b) for b in mine["private_batches"]], dist)
    all_batches = [p for r in g1 for p in r]
    ok = ok and all_batches == [batch(b) for b in range(8)]
    root = hashlib.sha256(b"".join(all_batches)).digest() if rank == plan["root"] else None
    want = hashlib.sha256(b"".join(hashlib.sha256(b"".join(leaf(i) for i in range(8 * b, 8 * b + 8))).digest() * 5 for b in range(8))).digest()
    ret[rank] = ok and (root is None or root == want)
    dist.barrier()
    dist.destroy_process_group()


def test_aggregation_tree_dataflow_four_ranks_gloo():
    world, port = 4, _free_port()
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_tree_worker, args=(world, port, ret), nprocs=world, join=True)
    assert all(ret[r] for r in range(world))
