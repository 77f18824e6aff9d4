"""Multi-GPU sharding of independent proofs.

The path shards across proofs: leaf proofs are independent (each is build_fresh().commit().prove(),
reference wormhole/aggregator/benches/aggregator.rs:45-49); private batches are independent of each other;
only the single public-batch proof is a join. One process per GPU; the only exchange is a gather of proof
bytes (about 130 KB per leaf proof) to the rank that proves the next level, an all_gather over RCCL/xGMI on
GPUs (gloo in the CPU tests). Nothing else crosses ranks.
"""
import numpy as np


def shard_range(num_items, world, rank):
    """Contiguous block partition: rank r gets items [lo, hi). BASELINE config 5: 64 leaves -> 8 per GPU."""
    base, rem = divmod(num_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def aggregation_schedule(num_leaves=64, leaves_per_private_batch=8, world=8):
    """Who proves what in the recursive tree (reference call stack SURVEY 3.4: 64 leaf proofs ->
    num_leaves/N private batches of N -> one public batch). Returns a dict rank -> {"leaves", "private_batches"}
    plus "root" (the rank that proves the public batch after the second gather)."""
    if num_leaves % leaves_per_private_batch:
        raise ValueError("num_leaves must be a multiple of leaves_per_private_batch")
    n_batches = num_leaves // leaves_per_private_batch
    plan = {"root": 0, "ranks": {}}
    for r in range(world):
        blo, bhi = shard_range(n_batches, world, r)
        leaves = list(range(blo * leaves_per_private_batch, bhi * leaves_per_private_batch))
        plan["ranks"][r] = {"leaves": leaves, "private_batches": list(range(blo, bhi))}
    return plan


def gather_proof_bytes(proofs, dist=None, device=None, layout=None, root=None):
    """All ranks contribute a list of proofs (bytes); every rank receives the list of all ranks' lists, in rank
    order. Fixed-size padded uint8 buffers + a length vector, one all_gather each (payload is latency-bound).
    `layout`: a dict the caller keeps between calls when every call has the same proof counts and sizes on every rank
    (proofs of one circuit have a fixed size): the two metadata collectives then run once, not per call.
    `root`: gather to that rank only (the proof bytes of a level go to the rank that proves the next one);
    every other rank gets None and neither receives nor decodes anything."""
    import os
    if dist is None or not dist.is_initialized():
        return [list(proofs)]
    import torch
    if dist.get_world_size() == 1 and os.environ.get("QPGPU_FORCE_COLLECTIVE") != "1":   # the test hook runs the collectives on one rank
        return [list(proofs)]
    world = dist.get_world_size()
    dev = device if device is not None else torch.device("cpu")
    if layout is not None and "counts" in layout:
        counts, all_lens, max_count, max_len = layout["counts"], layout["all_lens"], layout["max_count"], layout["max_len"]
        bad = len(proofs) != counts[dist.get_rank()] or any(len(p) != n for p, n in zip(proofs, all_lens[dist.get_rank()]))
        # every rank must leave together: a rank raising on its own would strand the others inside all_gather
        flag = torch.tensor([1 if bad else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if int(flag.item()):
            raise ValueError("gather_proof_bytes: proofs do not match the cached layout" + ("" if bad else " on another rank"))
    else:
        lens = torch.tensor([len(p) for p in proofs], dtype=torch.int64, device=dev)
        count = torch.tensor([len(proofs)], dtype=torch.int64, device=dev)
        counts_t = [torch.zeros_like(count) for _ in range(world)]
        dist.all_gather(counts_t, count)
        counts = [int(c.item()) for c in counts_t]
        max_count = max(counts)
        lens_pad = torch.zeros(max_count, dtype=torch.int64, device=dev)
        lens_pad[:len(proofs)] = lens
        all_lens_t = [torch.zeros_like(lens_pad) for _ in range(world)]
        dist.all_gather(all_lens_t, lens_pad)
        all_lens = [[int(x) for x in l.cpu().tolist()] for l in all_lens_t]
        max_len = max([max(l) if l else 0 for l in all_lens] + [0])
        if layout is not None:
            layout.update(counts=counts, all_lens=all_lens, max_count=max_count, max_len=max_len)
    staged = np.zeros(max_count * max_len, dtype=np.uint8)         # assembled on the host, one copy to the device
    for i, p in enumerate(proofs):
        staged[i * max_len:i * max_len + len(p)] = np.frombuffer(p, dtype=np.uint8)
    payload = torch.from_numpy(staged).to(dev)
    if root is not None:
        recv = [torch.zeros_like(payload) for _ in range(world)] if dist.get_rank() == root else None
        dist.gather(payload, gather_list=recv, dst=root)
        if recv is None:
            return None
    else:
        recv = [torch.zeros_like(payload) for _ in range(world)]
        dist.all_gather(recv, payload)
    out = []
    for r in range(world):
        buf = recv[r].cpu().numpy()
        out.append([buf[i * max_len:i * max_len + all_lens[r][i]].tobytes() for i in range(counts[r])])
    return out


class LocalProofBlocks:
    """The one-rank form of ProofBlockGather without torch: `blocks` blocks of `count` proof slots in plain host memory (a one-rank
    run has nothing to send, and importing torch would put a second ROCm runtime into the process)."""

    def __init__(self, count, proof_len, blocks=1):
        self.count, self.proof_len = count, proof_len
        self.send = [np.zeros((count, proof_len), dtype=np.uint8) for _ in range(blocks)]      # zeros: every page touched from this thread

    def slot(self, block, i):
        return self.send[block][i]

    def gather(self, block):
        return self.send[block][None]


class ProofBlockGather:
    """The per-step exchange of fixed-size proofs (every rank contributes `count` proofs of `proof_len` bytes, e.g. one bench
    step): the proving workers write straight into a pinned host block (`slot(i)` is proof i's output buffer), the block goes to
    the device in one copy, one all_gather, and the result comes back in one copy into a pinned host tensor
    [world][count][proof_len] — no per-proof staging or slicing on the host (with 8 ranks x 192 proofs of 130 KB the general
    gather_proof_bytes spends ~160 ms per step in host copies, which is most of a step). `blocks`: how many steps' blocks are
    kept (a ring; the bench submits steps ahead of the one it collects)."""

    def __init__(self, count, proof_len, dist, device, blocks=1, root=None):
        """root: None = every rank receives every rank's block (all_gather); a rank number = only that rank does (gather):
        the next aggregation level runs on one rank, and at 8 ranks x 192 proofs the all_gather form moves 8 x 25 MB into
        every rank and a 200 MB device-to-host copy per rank and step for bytes only the root reads."""
        import torch
        self.count, self.proof_len, self.dist, self.device = count, proof_len, dist, device
        self.root = root
        import os
        self.world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
        self.force = os.environ.get("QPGPU_FORCE_COLLECTIVE") == "1" and dist is not None and dist.is_initialized()   # test hook
        cuda = getattr(device, "type", "cpu") == "cuda"
        mk = (lambda *shape: torch.empty(*shape, dtype=torch.uint8).pin_memory()) if cuda else (lambda *shape: torch.empty(*shape, dtype=torch.uint8))
        self.send = [mk(count, proof_len) for _ in range(blocks)]
        self.rank = dist.get_rank() if dist is not None and dist.is_initialized() else 0
        self.receives = root is None or self.rank == root
        self.recv_host = mk(self.world if self.receives else 1, count, proof_len)
        for t in self.send + [self.recv_host]:      # touch every page now, from this thread: left to the proving threads' first
            t.zero_()                                # writes, the two-rank gloo rehearsal of the bench lost 13 % of its rate
        self.send_np = [t.numpy() for t in self.send]
        self.recv_dev = torch.empty(self.world if self.receives else 1, count, proof_len, dtype=torch.uint8, device=device) if cuda else self.recv_host
        self.send_dev = torch.empty(count, proof_len, dtype=torch.uint8, device=device) if cuda else None

    def slot(self, block, i):
        """numpy view of proof i's bytes in block `block` (hand it to the prover as its output buffer)"""
        return self.send_np[block][i]

    def gather(self, block):
        """all ranks' blocks -> uint8 host tensor [world][count][proof_len] (valid until the next gather); with a root, None on
        the other ranks"""
        import torch
        if self.world == 1 and not self.force:
            self.recv_host[0].copy_(self.send[block])
            return self.recv_host
        if self.root is not None:
            if self.send_dev is not None:
                self.send_dev.copy_(self.send[block], non_blocking=True)
                self.dist.gather(self.send_dev, gather_list=list(self.recv_dev.unbind(0)) if self.receives else None, dst=self.root)
                if self.receives:
                    self.recv_host.copy_(self.recv_dev, non_blocking=True)
                torch.cuda.current_stream(self.device).synchronize()
            else:
                self.dist.gather(self.send[block], gather_list=list(self.recv_host.unbind(0)) if self.receives else None, dst=self.root)
            return self.recv_host if self.receives else None
        if self.send_dev is not None:
            self.send_dev.copy_(self.send[block], non_blocking=True)
            self.dist.all_gather(list(self.recv_dev.unbind(0)), self.send_dev)
            self.recv_host.copy_(self.recv_dev, non_blocking=True)
            torch.cuda.current_stream(self.device).synchronize()
        else:
            self.dist.all_gather(list(self.recv_host.unbind(0)), self.send[block])
        return self.recv_host
