"""The RCCL leg of the multi-GPU path on the one GPU a test box has: a single-rank "nccl" process group runs the very
collectives bench.py --gpus N issues per step (all_reduce of the layout flag, all_gather of counts, lengths and the padded proof
payload) on device tensors. The multi-rank logic is covered with gloo (tests/test_sharding_gloo.py, tests/test_multirank_gpu.py);
this covers the backend those tests cannot use with two ranks on one device."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_gather_proof_bytes_over_rccl_single_rank():
    code = textwrap.dedent("""
        import os, sys
        sys.path.insert(0, %r)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29531", RANK="0", WORLD_SIZE="1", QPGPU_FORCE_COLLECTIVE="1")
        import torch, torch.distributed as dist
        import __graft_entry__ as ge
        pkg = ge.load_package()
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        dist.init_process_group(backend="nccl", device_id=dev)
        proofs = [bytes([i]) * (1000 + 7 * i) for i in range(5)]
        layout = {}
        for _ in range(2):                     # second call: cached layout, flag all_reduce + payload all_gather only
            got = pkg.sharding.gather_proof_bytes(proofs, dist, dev, layout)
            assert got == [proofs], "round trip over RCCL differs"
        # the bench's per-step exchange: pinned block -> device -> all_gather -> pinned host tensor
        blk = pkg.sharding.ProofBlockGather(3, 4096, dist, dev, blocks=2)
        for step in range(3):
            for i in range(3):
                blk.slot(step %% 2, i)[:] = (17 * step + i) %% 251
            got = blk.gather(step %% 2)
            assert tuple(got.shape) == (1, 3, 4096) and all(int(got[0, i, 0]) == (17 * step + i) %% 251 and int(got[0, i, -1]) == (17 * step + i) %% 251 for i in range(3))
        # what bench.py issues by default from round 3 on: gather to the consuming rank (dist.gather over RCCL, device tensors)
        got = pkg.sharding.gather_proof_bytes(proofs, dist, dev, {}, root=0)
        assert got == [proofs], "gather-to-root over RCCL differs"
        blk = pkg.sharding.ProofBlockGather(3, 4096, dist, dev, blocks=2, root=0)
        for step in range(3):
            for i in range(3):
                blk.slot(step %% 2, i)[:] = (29 * step + i) %% 251
            got = blk.gather(step %% 2)
            assert tuple(got.shape) == (1, 3, 4096) and all(int(got[0, i, 7]) == (29 * step + i) %% 251 for i in range(3))
        t = torch.ones(4, device=dev); dist.all_reduce(t); dist.barrier(); torch.cuda.synchronize(dev)
        dist.destroy_process_group()
        print("rccl ok")
    """ % ROOT)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "rccl ok" in r.stdout, r.stdout[-500:] + r.stderr[-1500:]
