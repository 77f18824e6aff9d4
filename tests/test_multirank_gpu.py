"""BASELINE configs[3] and configs[4] through the HIP path with several ranks.

Each rank is a fresh interpreter (tests/multirank_worker.py) started with subprocess — never a fork of this process, which
may already hold a GPU context — with the torchrun environment (RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT).
The ranks share the box's one GPU and gather over gloo; bench.py runs the same calls over RCCL with one GPU per rank.
Every proof of every level is compared byte for byte with the CPU oracle inside the workers.
"""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "multirank_worker.py")


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _run_ranks(mode, world, tmp_path, timeout=900, extra_env=None):
    port = _free_port()
    out = str(tmp_path / ("result_" + mode + ".json"))
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable, WORKER, mode, out], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = []
    try:
        for p in procs:
            logs.append(p.communicate(timeout=timeout)[0])
    finally:
        for p in procs:            # exactly the processes started here
            if p.poll() is None:
                p.kill()
    for rank, p in enumerate(procs):
        assert p.returncode == 0, "rank %d failed:\n%s" % (rank, logs[rank][-3000:])
    return [json.load(open(out + ".%d" % r)) for r in range(world)]


@pytest.mark.gpu
def test_eight_leaf_proofs_over_two_ranks(tmp_path):
    """configs[3]: 8 independent leaf proofs (different witnesses), sharded 4 + 4, gathered in rank order, each byte-equal
    to the oracle's proof."""
    res = _run_ranks("leaves", 2, tmp_path)
    assert all(r["ok"] for r in res) and res[0]["proofs"] == 8


@pytest.mark.gpu
def test_aggregation_tree_over_two_ranks(tmp_path):
    """configs[4]: 64 leaves -> 8 zero-knowledge private batches -> 1 public batch; every level consumes the previous level's
    gathered proof bytes; every proof byte-equal to the oracle's; the root holds the 8 batch proofs in rank order."""
    res = _run_ranks("tree", 2, tmp_path)
    assert all(r["ok"] for r in res)
    assert res[0]["leaves"] == 64 and res[0]["batches"] == 8 and res[0]["root_bytes"] > 0 and res[1]["root_bytes"] == 0


@pytest.mark.gpu
def test_aggregation_tree_gathers_to_the_consuming_rank_only(tmp_path):
    """The same tree with exchange="root" (what bench.py times): the leaf level exchanges nothing — a rank's private batches
    consume its own leaves — and the private-batch proofs travel to the root rank alone; every proof still byte-equal to the
    oracle's and the root's public inputs unchanged."""
    res = _run_ranks("tree", 2, tmp_path, extra_env={"QP_TEST_EXCHANGE": "root"})
    assert all(r["ok"] for r in res)
    assert res[0]["root_bytes"] > 0 and res[1]["root_bytes"] == 0


@pytest.mark.gpu
def test_leaf_proofs_single_rank_matches(tmp_path):
    """The same 8 proofs from one rank: sharding does not change any proof."""
    res = _run_ranks("leaves", 1, tmp_path)
    assert res[0]["ok"] and res[0]["proofs"] == 8


@pytest.mark.gpu
def test_attesting_tree_over_two_ranks(tmp_path):
    """The two-level tree whose wrappers check the Merkle half of their inner proofs, sharded: a rank proves its batches' leaves
    and first-level wrappers, the first-level proofs travel to rank 0 (the one exchange), rank 0 proves the root; the library's
    and the oracle's verifiers accept it and its public inputs are the eight leaves' in order."""
    res = _run_ranks("attest", 2, tmp_path)
    assert all(r["ok"] for r in res) and res[0]["root_bytes"] > 0 and res[1]["root_bytes"] == 0
