"""Aggregation front-end: the data flow of the reference's batch provers on top of the GPU prover.

Reference flow:
  PrivateBatchProver::commit(leaf proofs) -> fill_private_batch_witness -> prove
      wormhole/aggregator/src/private_batch/prover/lib.rs:244-343, witness.rs:15-77
  PublicBatchProver::commit(private-batch proofs) -> prove
      wormhole/aggregator/src/public_batch/prover/lib.rs:268-305, aggregator.rs:187-227
A batch proof's public inputs are a function of its inner proofs' public inputs, and its witness is regenerated
from them; the proof bytes of level k are the input of level k+1.

The recursive wrapper circuits themselves need the Rust CircuitBuilder (circuit-pack exporter), so the
circuit proven here is the synthetic stand-in of the level's size and gate mix ("shape-equivalent"). What is
real: inner proofs are parsed from their bytes, the batch's public inputs are the inner public inputs in slot order
(padded with the dummy template's, as the reference pads a short batch, private_batch/prover/lib.rs:283-300) followed by
the level's 8 trailing words, the witness is generated on the device from a PartialWitness (stage s1, with plonky2's
"set twice with different values" check), and the proof is produced by the same qpgpu_prove path as every other proof.
What is not: the inner proofs are not verified in-circuit (no recursive verifier gates are wired to them).
"""
import numpy as np

from .binding import Circuit, pack_header, pack_public_input_cells

LEAF_PUBLIC_INPUTS = 21          # wormhole/inputs/src/lib.rs:33
BATCH_TRAILER_WORDS = 8          # private_batch/circuit/constants.rs:92-94: 21 * N + 8


def proof_public_inputs(proof, num_public_inputs):
    """The public inputs of a serialized ProofWithPublicInputs: its last num_public_inputs little-endian u64 words
    (stage s12 order: proof, then public inputs; no length prefixes)."""
    if len(proof) < 8 * num_public_inputs:
        raise ValueError("proof shorter than its public inputs")
    return np.frombuffer(proof, dtype="<u8", offset=len(proof) - 8 * num_public_inputs, count=num_public_inputs).copy()


def batch_public_inputs(inner_proofs, inner_num_public_inputs, slots, dummy_public_inputs, trailer):
    """Public inputs of a batch over `slots` inner proofs: slot i carries inner proof i's public inputs, missing slots the
    dummy template's; then the level's trailer words."""
    if len(inner_proofs) > slots:
        raise ValueError("more inner proofs than slots")
    rows = [proof_public_inputs(p, inner_num_public_inputs) for p in inner_proofs]
    rows += [np.asarray(dummy_public_inputs, dtype=np.uint64)] * (slots - len(rows))
    out = np.concatenate(rows + [np.asarray(trailer, dtype=np.uint64)])
    return out.astype(np.uint64)


class TemplateProver:
    """commit(public inputs) -> prove() for one circuit on one GPU: the PartialWitness is the circuit's template witness
    (its free cells other than the public-input cells) plus the given public inputs; the full witness is generated on the
    device. Stands in for WormholeProver::commit / prove (wormhole/prover/src/lib.rs:156-175) on a synthetic leaf circuit."""

    def __init__(self, gpu, pack, template_wires, max_batch=1):
        self.gpu, self.pack = gpu, pack
        self.hdr = pack_header(pack)
        self.max_batch = max_batch
        self.circ = Circuit(gpu, pack, max_batch=max_batch)
        nw, n = self.hdr["num_wires"], 1 << self.hdr["degree_bits"]
        pi_cells = pack_public_input_cells(pack)
        if pi_cells is None:
            raise ValueError("circuit pack carries no public-input cell trailer")
        mask = self.circ.witness_free_mask(nw, n)
        col, row = np.nonzero(mask == 1)
        cells = row.astype(np.uint64) * np.uint64(nw) + col.astype(np.uint64)
        keep = ~np.isin(cells, pi_cells)
        self.cells = cells[keep]
        self.values = np.asarray(template_wires, dtype=np.uint64)[col[keep], row[keep]]
        self.mat_bytes = nw * n * 8
        self.d_wires = gpu.alloc(self.mat_bytes * max_batch)
        self.d_template = None          # the template's PartialWitness expanded on the device (first commit)
        self.pis = None
        self.batch_pis = None

    def close(self):
        if self.circ is not None:
            self.d_wires.free()
            if self.d_template is not None:
                self.d_template.free()
            self.circ.close()
            self.circ = None

    def proof_size(self):
        return self.circ.proof_size()

    def commit(self, public_inputs, extra_cells=(), extra_values=()):
        """Generate the witness for these public inputs on the device (extra assignments are added to the PartialWitness;
        one that contradicts a generated value raises QpGpuError(-4), plonky2's "set twice with different values")."""
        pis = np.ascontiguousarray(public_inputs, dtype=np.uint64)
        if pis.size != self.hdr["num_public_inputs"]:
            raise ValueError("wrong number of public inputs")
        if len(extra_cells):
            cells = np.concatenate([self.cells, np.asarray(extra_cells, dtype=np.uint64)])
            values = np.concatenate([self.values, np.asarray(extra_values, dtype=np.uint64)])
            self.circ.generate_witness_partial_dev(cells, values, pis, self.d_wires)
        else:
            self._expand(1, pis)
            self.circ.generate_witness_dev(self.d_wires, pis)
        self.pis = pis
        return pis

    def _expand(self, nb, pis):
        """nb copies of the template's expanded PartialWitness in d_wires. The template's free cells are part of the circuit
        stand-in, not of a proof's input: they go through the PartialWitness entry (upload, conflict check) once per prover;
        a commit then only rewrites the public inputs and regenerates what depends on them."""
        if self.d_template is None:
            self.d_template = self.gpu.alloc(self.mat_bytes)
            self.circ.generate_witness_partial_dev(self.cells, self.values, pis, self.d_template)
        for b in range(nb):
            self.gpu._check(self.gpu.lib.qpgpu_memcpy_d2d(self.gpu.ctx, self.d_wires.ptr + b * self.mat_bytes, self.d_template.ptr, self.mat_bytes))

    def commit_many(self, public_inputs_list):
        """Witnesses for up to max_batch public-input vectors at once: the PartialWitness of the first is expanded on the
        device, copied, and one batched generation pass (stage s1, all dependency levels walked once) rewrites the public-input
        cells and everything that depends on them in every copy."""
        nb = len(public_inputs_list)
        if nb == 0 or nb > self.max_batch:
            raise ValueError("batch size outside 1..max_batch")
        pis = np.ascontiguousarray(np.stack([np.asarray(p, dtype=np.uint64) for p in public_inputs_list]))
        self._expand(nb, pis[0])
        self.circ.generate_witness_dev(self.d_wires, pis, batch=nb)
        self.batch_pis = pis
        return pis

    def prove_many(self):
        """The committed batch in lockstep (qpgpu_prove_batch_dev); returns the list of proofs."""
        if self.batch_pis is None:
            raise RuntimeError("prove_many() before commit_many()")
        pis = self.batch_pis
        self.batch_pis = None
        return self.circ.prove_batch_dev([self.d_wires.ptr + b * self.mat_bytes for b in range(len(pis))], list(pis))

    def witness(self, b=0):
        """The committed full witness [num_wires, n] of batch slot b (host copy; tests compare it with the oracle's view)."""
        nw, n = self.hdr["num_wires"], 1 << self.hdr["degree_bits"]
        return self.d_wires.download(count=(b + 1) * self.mat_bytes // 8)[b * nw * n:].reshape(nw, n)

    def prove(self, out=None):
        if self.pis is None:
            raise RuntimeError("prove() before commit()")
        proof = self.circ.prove_dev(self.d_wires, self.pis, out)
        self.pis = None
        return proof


class BatchProver(TemplateProver):
    """commit(inner proofs) -> prove(): one batch level on one GPU.

    pack / template_wires / template_pis: the level's circuit and one satisfying witness of it (the dummy-proof template the
    reference generates at build time, dummy_proof.rs:104-115)."""

    def __init__(self, gpu, pack, template_wires, template_pis, inner_num_public_inputs, slots, max_batch=1):
        super().__init__(gpu, pack, template_wires, max_batch=max_batch)
        self.inner_npis, self.slots = inner_num_public_inputs, slots
        if self.hdr["num_public_inputs"] != slots * inner_num_public_inputs + BATCH_TRAILER_WORDS:
            self.close()
            raise ValueError("batch circuit public-input count does not match slots * inner + 8")
        self.template_pis = np.asarray(template_pis, dtype=np.uint64)

    def commit(self, inner_proofs, trailer=None):
        """Parse the inner proofs, derive the batch's public inputs, generate the witness on the device."""
        dummy = self.template_pis[:self.inner_npis]
        tr = self.template_pis[-BATCH_TRAILER_WORDS:] if trailer is None else trailer
        return super().commit(batch_public_inputs(inner_proofs, self.inner_npis, self.slots, dummy, tr))

    def commit_many(self, inner_proofs_list, trailer=None):
        """Several batches of this level at once (one witness-generation pass, proven in lockstep by prove_many)."""
        dummy = self.template_pis[:self.inner_npis]
        tr = self.template_pis[-BATCH_TRAILER_WORDS:] if trailer is None else trailer
        return super().commit_many([batch_public_inputs(ps, self.inner_npis, self.slots, dummy, tr) for ps in inner_proofs_list])


def leaf_public_inputs(index, count=LEAF_PUBLIC_INPUTS):
    """Deterministic stand-in public inputs for synthetic leaf `index` (canonical field elements)."""
    x = (np.arange(count, dtype=np.uint64) + np.uint64(1 + index * 1000003)) * np.uint64(0x9E3779B97F4A7C15)
    x ^= x >> np.uint64(31)
    return (x % np.uint64(0xFFFFFFFF00000001)).astype(np.uint64)


class AggregationTree:
    """BASELINE configs[4] on this rank: `num_leaves` leaf proofs sharded over the ranks, one private batch (zero-knowledge)
    per `slots` leaves, one public batch over the private batches on the root rank; the proof bytes of every level are
    gathered (sharding.gather_proof_bytes: RCCL on GPUs, gloo in rehearsals) and consumed by the next level.
    Reference call stack SURVEY.md 3.4; partitioning SURVEY.md 8e."""

    def __init__(self, pkg, gpu, rank, world, leaf, private, public, num_leaves=64, slots=8, leaf_batch=8, private_batch=8):
        """leaf / private / public: (pack, template_wires, template_pis) of the level's circuit (public only on the root).
        leaf_batch / private_batch: how many proofs of a level this rank proves in lockstep."""
        from . import sharding
        self.sharding, self.rank, self.world, self.slots = sharding, rank, world, slots
        self.plan = sharding.aggregation_schedule(num_leaves, slots, world)
        self.mine = self.plan["ranks"][rank]
        self.num_batches = num_leaves // slots
        self.leaf_batch = max(1, min(leaf_batch, len(self.mine["leaves"]))) if self.mine["leaves"] else 1
        self.leaf = TemplateProver(gpu, leaf[0], leaf[1], max_batch=self.leaf_batch)
        self.private_batch = max(1, min(private_batch, len(self.mine["private_batches"])))
        self.private = BatchProver(gpu, private[0], private[1], private[2], LEAF_PUBLIC_INPUTS, slots, max_batch=self.private_batch)
        self.times = {}
        self.public = None
        if rank == self.plan["root"]:
            self.public = BatchProver(gpu, public[0], public[1], public[2], LEAF_PUBLIC_INPUTS * slots + BATCH_TRAILER_WORDS, self.num_batches)

    def close(self):
        for p in (self.leaf, self.private, self.public):
            if p is not None:
                p.close()

    def run(self, dist=None, device=None, blinding_seed=None, keep=None):
        """One pass over the tree. Returns (all leaf proofs, all private-batch proofs, root proof or None).
        blinding_seed: makes the zero-knowledge level reproducible (tests); keep: a dict that receives, per level, this
        rank's (index, public inputs, full witness) triples for an external checker (costs a device download each)."""
        import time
        d = dist if self.world > 1 else None
        t0 = time.perf_counter()
        mine_leaf = []
        ids = list(self.mine["leaves"])
        for k in range(0, len(ids), self.leaf_batch):   # this rank's leaves, leaf_batch at a time in lockstep
            chunk = ids[k:k + self.leaf_batch]
            pis = self.leaf.commit_many([leaf_public_inputs(i) for i in chunk])
            if keep is not None:
                all_w = self.leaf.d_wires.download(count=len(chunk) * self.leaf.mat_bytes // 8).reshape(len(chunk), self.leaf.hdr["num_wires"], -1)
                for j, i in enumerate(chunk):
                    keep.setdefault("leaf", []).append((i, pis[j].copy(), all_w[j].copy()))
            mine_leaf += self.leaf.prove_many()
        leaves = [p for r in self.sharding.gather_proof_bytes(mine_leaf, d, device) for p in r]
        t1 = time.perf_counter()
        mine_priv = []
        # this rank's private batches in lockstep, in runs of consecutive batch numbers (proof j of a run is blinded with
        # seed + first + j, the same salts the one-at-a-time order would draw)
        pb = list(self.mine["private_batches"])
        k = 0
        while k < len(pb):
            run = [pb[k]]
            while k + len(run) < len(pb) and len(run) < self.private_batch and pb[k + len(run)] == run[-1] + 1:
                run.append(pb[k + len(run)])
            pis = self.private.commit_many([leaves[b * self.slots:(b + 1) * self.slots] for b in run])
            if blinding_seed is not None:
                self.private.circ.set_blinding_seed(blinding_seed + run[0])
            if keep is not None:
                for j, b in enumerate(run):
                    keep.setdefault("private", []).append((b, pis[j].copy(), self.private.witness(j)))
            mine_priv += self.private.prove_many()
            k += len(run)
        batches = [p for r in self.sharding.gather_proof_bytes(mine_priv, d, device) for p in r]
        t2 = time.perf_counter()
        root = None
        if self.public is not None:
            pis = self.public.commit(batches)
            if keep is not None:
                keep.setdefault("public", []).append((0, pis.copy(), self.public.witness()))
            root = self.public.prove()
        t3 = time.perf_counter()
        self.times = {"leaf_level_s": round(t1 - t0, 4), "private_level_s": round(t2 - t1, 4), "public_level_s": round(t3 - t2, 4)}
        return leaves, batches, root
