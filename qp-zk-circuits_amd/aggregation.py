"""Aggregation front-end: the data flow of the reference's batch provers on top of the GPU prover.

Reference flow:
  PrivateBatchProver::commit(leaf proofs) -> fill_private_batch_witness -> prove
      wormhole/aggregator/src/private_batch/prover/lib.rs:244-343, witness.rs:15-77
  PublicBatchProver::commit(private-batch proofs) -> prove
      wormhole/aggregator/src/public_batch/prover/lib.rs:268-305, aggregator.rs:187-227
A batch proof's public inputs are a function of its inner proofs' public inputs, and its witness is regenerated
from them; the proof bytes of level k are the input of level k+1.

THIS module is round 3's form of the two levels: the circuit proven is a SYNTHETIC stand-in of the level's size and gate mix
("shape-equivalent") — kept for the bench's `aggregation_tree` leg and as the home of the host-side helpers
(public-input layouts, parsers, outputs). The levels with circuits that verify their inner proofs (recursive verifier + the
layer's own constraints, restated on the library's builder) are in recursion.py: PrivateBatchProver / PublicBatchProver /
ProvingContext / AttestingTree. What is real here: inner proofs are parsed from their bytes; the reference's admission checks run on them (include/qpgpu_batch.h:
counts, asset / block / fee consistency, duplicate nullifiers, all-dummy, the padding templates' sentinels; the
cryptographic half through the caller's verifier); short batches are padded with the dummy template, private batches
shuffled uniformly and given one dummy-nullifier preimage per slot; the batch's public inputs are exactly what the
wrapper circuit would emit (layout 21 N + 8 / 12 + 14 M N: references from the first non-dummy slot, exit accounts
merged, dummy nullifiers hashed, the nullifier region sorted); the witness is generated on the device from a
PartialWitness (stage s1); and the proof is produced by the same qpgpu_prove path as every other proof.
What is not, in the stand-ins of this module: the inner proofs are not verified in-circuit (recursion.py's circuits do that).
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


# ---- include/qpgpu_batch.h: public-input layouts, commit preflights, padding / shuffle, wrapper-circuit outputs ----
_ERR_CAP = 400
_batch_lib = None


def _lib():
    global _batch_lib
    if _batch_lib is None:
        import ctypes
        from .binding import load_library
        L = load_library()
        vp, sz, cp = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p
        for name, argt in {"qpgpu_private_batch_preflight": [vp, sz, sz, cp], "qpgpu_dummy_leaf_template_check": [vp, sz, cp],
                           "qpgpu_private_batch_arrange": [sz, sz, cp, vp, vp, cp], "qpgpu_private_batch_outputs": [vp, sz, vp, vp, cp],
                           "qpgpu_public_batch_preflight": [vp, sz, sz, sz, cp], "qpgpu_dummy_private_batch_template_check": [vp, sz, cp],
                           "qpgpu_public_batch_outputs": [vp, sz, sz, cp, vp, cp],
                           "qpgpu_private_batch_public_inputs_parse": [vp, sz, vp, vp, vp, cp],
                           "qpgpu_public_batch_public_inputs_parse": [vp, sz, ctypes.c_uint64, ctypes.c_uint64, vp, vp, vp, cp],
                           "qpgpu_poseidon2_hash_bytes": [vp, sz, vp, sz, cp]}.items():
            getattr(L, name).argtypes = argt
            getattr(L, name).restype = ctypes.c_int
        L.qpgpu_private_batch_pi_len.argtypes = [sz]; L.qpgpu_private_batch_pi_len.restype = sz
        L.qpgpu_public_batch_pi_len.argtypes = [sz, sz]; L.qpgpu_public_batch_pi_len.restype = sz
        _batch_lib = L
    return _batch_lib


def _call(fn, *args):
    """Reference behaviour: an admission failure is an error with the reference's text (anyhow -> ValueError); inputs
    the wrapper circuit cannot satisfy are QpGpuError(-4)."""
    import ctypes
    from .binding import QpGpuError
    err = ctypes.create_string_buffer(_ERR_CAP)
    rc = fn(*args, err)
    if rc == -4:
        raise QpGpuError(rc, err.value.decode())
    if rc != 0:
        raise ValueError(err.value.decode())


def private_batch_pi_len(num_leaf_proofs):
    return int(_lib().qpgpu_private_batch_pi_len(num_leaf_proofs))


def public_batch_pi_len(num_private_batch_proofs, num_leaf_proofs):
    return int(_lib().qpgpu_public_batch_pi_len(num_private_batch_proofs, num_leaf_proofs))


def private_batch_outputs(leaf_rows, preimages):
    """What the private-batch circuit writes into its public inputs for these slots (circuit_logic.rs:170-523)."""
    rows = np.ascontiguousarray(leaf_rows, dtype=np.uint64).reshape(-1, LEAF_PUBLIC_INPUTS)
    pre = np.ascontiguousarray(preimages, dtype=np.uint64).reshape(rows.shape[0], 4)
    out = np.zeros(private_batch_pi_len(rows.shape[0]), dtype=np.uint64)
    _call(_lib().qpgpu_private_batch_outputs, rows.ctypes.data, rows.shape[0], pre.ctypes.data, out.ctypes.data)
    return out


def public_batch_outputs(inner_rows, num_leaf_proofs, aggregator_address):
    rows = np.ascontiguousarray(inner_rows, dtype=np.uint64).reshape(-1, private_batch_pi_len(num_leaf_proofs))
    out = np.zeros(public_batch_pi_len(rows.shape[0], num_leaf_proofs), dtype=np.uint64)
    _call(_lib().qpgpu_public_batch_outputs, rows.ctypes.data, rows.shape[0], num_leaf_proofs, bytes(aggregator_address), out.ctypes.data)
    return out


def _parse(fn, hdr_fields, pis, n_slots, n_nulls, *counts):
    import ctypes
    u32, b32 = ctypes.c_uint32, ctypes.c_uint8 * 32

    class Hdr(ctypes.Structure):
        _fields_ = [(k, b32 if k in ("block_hash", "aggregator_address") else u32) for k in hdr_fields]

    class Slot(ctypes.Structure):
        _fields_ = [("summed_output_amount", u32), ("exit_account", b32)]

    pis = np.ascontiguousarray(pis, dtype=np.uint64)
    hdr = Hdr(); slots = (Slot * max(n_slots, 1))(); nulls = ctypes.create_string_buffer(32 * max(n_nulls, 1))
    _call(fn, pis.ctypes.data, pis.size, *counts, ctypes.byref(hdr), slots, nulls)
    h = {k: (bytes(getattr(hdr, k)) if k in ("block_hash", "aggregator_address") else int(getattr(hdr, k))) for k in hdr_fields}
    return (h, [(int(slots[i].summed_output_amount), bytes(slots[i].exit_account)) for i in range(n_slots)],
            [nulls.raw[32 * i:32 * i + 32] for i in range(n_nulls)])


def parse_private_batch_public_inputs(pis):
    """PrivateBatchPublicInputs::try_from_u64_slice: (header dict, [(summed_output_amount, exit_account)] x 2N, [nullifier] x N);
    ValueError with the reference's message for a malformed slice."""
    n = (len(pis) - 8) // LEAF_PUBLIC_INPUTS if len(pis) >= 8 else 0
    return _parse(_lib().qpgpu_private_batch_public_inputs_parse, ("num_exit_slots", "asset_id", "volume_fee_bps", "block_hash", "block_number", "n_leaf"),
                  pis, 2 * n, n)


def parse_public_batch_public_inputs(pis, num_private_batch_proofs, num_leaf_proofs):
    """PublicBatchPublicInputs::try_from_u64_slice."""
    m, n = num_private_batch_proofs, num_leaf_proofs
    return _parse(_lib().qpgpu_public_batch_public_inputs_parse, ("aggregator_address", "asset_id", "volume_fee_bps", "block_hash", "block_number", "total_exit_slots"),
                  pis, m * 2 * n, m * n, m, n)


def dummy_nullifier(preimage):
    """hash_dummy_nullifier_pre_image (private_batch/circuit/circuit_logic.rs:479-488): H(H(preimage)) with the fork's
    Poseidon2Hash::hash_no_pad, as the 32 bytes the parsers hand out."""
    import ctypes
    pre = np.ascontiguousarray(preimage, dtype=np.uint64)
    out = ctypes.create_string_buffer(32)
    if _lib().qpgpu_poseidon2_hash_bytes(None, 0, pre.ctypes.data, 4, out) != 0:
        raise ValueError("dummy_nullifier: hash failed")
    inner = np.frombuffer(out.raw, dtype=np.uint64).copy()
    if _lib().qpgpu_poseidon2_hash_bytes(None, 0, inner.ctypes.data, 4, out) != 0:
        raise ValueError("dummy_nullifier: hash failed")
    return out.raw


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
            self.d_wires.free(scrub=True)          # witnesses: zeroed before release
            if self.d_template is not None:
                self.d_template.free(scrub=True)
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


class MemoVerifier:
    """verify(list of proofs) -> list of bool over a verifier's verify_many, remembering verdicts: a level that admits its
    inner proofs batch by batch can have all of them verified in one call (all host cores busy once) beforehand."""

    def __init__(self, verify_many):
        self.verify_many, self.seen = verify_many, {}

    def __call__(self, proofs):
        todo = [p for p in proofs if p not in self.seen]
        if todo:
            for p, ok in zip(todo, self.verify_many(todo)):
                self.seen[p] = ok
        return [self.seen[p] for p in proofs]

    def forget(self):
        self.seen.clear()


class PrivateBatchProver(TemplateProver):
    """PrivateBatchProver::{commit, prove, aggregate} (private_batch/prover/lib.rs:244-343) on one GPU.

    commit: the reference's admission checks on the supplied leaf proofs (qpgpu_private_batch_preflight, then the
    cryptographic `verify_leaf(list of proofs) -> list of bool`, e.g. binding.Verifier.verify_many, if one is given), padding with the dummy leaf template, a uniform shuffle,
    one dummy-nullifier preimage per slot; the batch's public inputs are what the private-batch circuit would emit for
    those slots (qpgpu_private_batch_outputs) and the witness is regenerated from them on the device."""

    def __init__(self, gpu, pack, template_wires, dummy_leaf_proof, num_leaf_proofs, verify_leaf=None, max_batch=1):
        super().__init__(gpu, pack, template_wires, max_batch=max_batch)
        self.num_leaf_proofs, self.verify_leaf = num_leaf_proofs, verify_leaf
        try:
            if self.hdr["num_public_inputs"] != private_batch_pi_len(num_leaf_proofs):
                raise ValueError("private-batch circuit public-input count is not 21 * N + 8")
            # verify_dummy_leaf_template: the sentinel, then the cryptographic check
            self.dummy_pis = proof_public_inputs(dummy_leaf_proof, LEAF_PUBLIC_INPUTS)
            _call(_lib().qpgpu_dummy_leaf_template_check, self.dummy_pis.ctypes.data, self.dummy_pis.size)
            if verify_leaf is not None and not verify_leaf([dummy_leaf_proof])[0]:
                raise ValueError("dummy leaf proof template failed verification")
        except Exception:
            self.close()
            raise
        self.dummy_leaf_proof = dummy_leaf_proof
        self.arrangement = None       # of the last commit: (slot_source, preimages) per batch

    def _slots(self, leaf_proofs, seed):
        """-> (rows of 21 felts in slot order, preimages) after the admission checks, padding and shuffle."""
        N = self.num_leaf_proofs
        rows = np.stack([proof_public_inputs(p, LEAF_PUBLIC_INPUTS) for p in leaf_proofs]) if len(leaf_proofs) else np.zeros((0, LEAF_PUBLIC_INPUTS), dtype=np.uint64)
        rows = np.ascontiguousarray(rows, dtype=np.uint64)
        _call(_lib().qpgpu_private_batch_preflight, rows.ctypes.data if rows.size else None, rows.shape[0], N)
        if self.verify_leaf is not None:
            for i, ok in enumerate(self.verify_leaf(list(leaf_proofs))):
                if not ok:
                    raise ValueError("leaf proof %d failed verification against the pinned leaf verifier" % i)
        src = np.zeros(N, dtype=np.uint32)
        pre = np.zeros(4 * N, dtype=np.uint64)
        _call(_lib().qpgpu_private_batch_arrange, rows.shape[0], N, seed, src.ctypes.data, pre.ctypes.data)
        slot_rows = np.stack([self.dummy_pis if k == 0xFFFFFFFF else rows[k] for k in src.tolist()])
        return slot_rows, pre.reshape(N, 4), src

    def batch_public_inputs(self, leaf_proofs, seed=None):
        slot_rows, pre, src = self._slots(leaf_proofs, seed)
        return private_batch_outputs(slot_rows, pre), (src, pre)

    def commit(self, leaf_proofs, seed=None):
        pis, arr = self.batch_public_inputs(leaf_proofs, seed)
        self.arrangement = [arr]
        return super().commit(pis)

    def commit_many(self, leaf_proofs_list, seed=None):
        """Several private batches at once (one witness-generation pass, proven in lockstep by prove_many); with a seed,
        batch k is arranged from seed with its last byte increased by k."""
        out, arrs = [], []
        for k, ps in enumerate(leaf_proofs_list):
            sk = None if seed is None else bytes(seed[:31]) + bytes([(seed[31] + k) & 0xFF])
            pis, arr = self.batch_public_inputs(ps, sk)
            out.append(pis); arrs.append(arr)
        self.arrangement = arrs
        return super().commit_many(out)

    def prove_dummy_template(self, blinding_seed=None):
        """generate_dummy_private_batch_proof (private_batch/circuit/build.rs:165-193): the all-dummy private-batch proof the
        public level pads with; built from explicit dummy leaves, never through commit (which refuses an all-dummy batch)."""
        N = self.num_leaf_proofs
        pre = np.zeros(4 * N, dtype=np.uint64); src = np.zeros(N, dtype=np.uint32)
        _call(_lib().qpgpu_private_batch_arrange, 1, N, None, src.ctypes.data, pre.ctypes.data)     # only the preimages are used
        pis = private_batch_outputs(np.stack([self.dummy_pis] * N), pre.reshape(N, 4))
        TemplateProver.commit(self, pis)
        if blinding_seed is not None:
            self.circ.set_blinding_seed(blinding_seed)
        return self.prove()


class PublicBatchProver(TemplateProver):
    """PublicBatchProver::{commit, prove} (public_batch/prover/lib.rs:268-305): admission checks on the supplied private-batch
    proofs, order-preserving padding with the dummy private-batch template (no shuffle), public inputs as the public-batch
    circuit emits them."""

    def __init__(self, gpu, pack, template_wires, dummy_private_batch_proof, num_private_batch_proofs, num_leaf_proofs, verify_inner=None):
        super().__init__(gpu, pack, template_wires)
        self.M, self.N, self.verify_inner = num_private_batch_proofs, num_leaf_proofs, verify_inner
        self.inner_len = private_batch_pi_len(num_leaf_proofs)
        try:
            if self.hdr["num_public_inputs"] != public_batch_pi_len(self.M, self.N):
                raise ValueError("public-batch circuit public-input count is not 12 + 14 * M * N")
            self.dummy_pis = proof_public_inputs(dummy_private_batch_proof, self.inner_len)
            _call(_lib().qpgpu_dummy_private_batch_template_check, self.dummy_pis.ctypes.data, self.dummy_pis.size)
            if verify_inner is not None and not verify_inner([dummy_private_batch_proof])[0]:
                raise ValueError("dummy private-batch proof template failed verification")
        except Exception:
            self.close()
            raise

    def batch_public_inputs(self, inner_proofs, aggregator_address=bytes(32)):
        rows = np.stack([proof_public_inputs(p, self.inner_len) for p in inner_proofs]) if len(inner_proofs) else np.zeros((0, self.inner_len), dtype=np.uint64)
        rows = np.ascontiguousarray(rows, dtype=np.uint64)
        _call(_lib().qpgpu_public_batch_preflight, rows.ctypes.data if rows.size else None, rows.shape[0], self.inner_len, self.M)
        if self.verify_inner is not None:
            for i, ok in enumerate(self.verify_inner(list(inner_proofs))):
                if not ok:
                    raise ValueError("private-batch proof %d failed verification against the pinned private-batch verifier" % i)
        padded = np.concatenate([rows] + [self.dummy_pis[None, :]] * (self.M - rows.shape[0]))
        return public_batch_outputs(padded, self.N, aggregator_address)

    def commit(self, inner_proofs, aggregator_address=bytes(32)):
        return super().commit(self.batch_public_inputs(inner_proofs, aggregator_address))


TEST_BLOCK_HASH = (0x0B10C0DE, 0x2222, 0x3333, 0x4444)


def leaf_public_inputs(index, count=LEAF_PUBLIC_INPUTS):
    """Deterministic stand-in public inputs for synthetic leaf `index`. With the leaf layout (21 felts,
    wormhole/inputs/src/lib.rs:25-33) they are well-formed leaf public inputs of one block: native asset, two outputs to
    accounts drawn from a small set (so that batches have exit accounts to merge), fee 10 bps, a distinct nullifier."""
    x = (np.arange(count, dtype=np.uint64) + np.uint64(1 + index * 1000003)) * np.uint64(0x9E3779B97F4A7C15)
    x ^= x >> np.uint64(31)
    x = (x % np.uint64(0xFFFFFFFF00000001)).astype(np.uint64)
    if count != LEAF_PUBLIC_INPUTS:
        return x
    p = np.zeros(LEAF_PUBLIC_INPUTS, dtype=np.uint64)
    p[1], p[2], p[3] = 1000 + index, index % 7, 10
    p[4:8] = x[4:8]
    a1, a2 = index % 5, (index + 1) % 5
    p[8:12] = [0xACC0 + a1, 1, 2, 3]
    p[12:16] = [0xACC0 + a2, 1, 2, 3]
    p[16:20] = TEST_BLOCK_HASH
    p[20] = 42
    return p


class AggregationTree:
    """BASELINE configs[4] on this rank: `num_leaves` leaf proofs sharded over the ranks, one private batch (zero-knowledge)
    per `slots` leaves, one public batch over the private batches on the root rank; the proof bytes of every level are
    gathered (sharding.gather_proof_bytes: RCCL on GPUs, gloo in rehearsals) and consumed by the next level.
    Reference call stack SURVEY.md 3.4; partitioning SURVEY.md 8e."""

    def __init__(self, pkg, gpu, rank, world, leaf, private, public, num_leaves=64, slots=8, leaf_batch=8, private_batch=8, verify="product"):
        """leaf / private / public: (pack, template_wires, ...) of the level's circuit (public only on the root).
        leaf_batch / private_batch: how many proofs of a level this rank proves in lockstep.
        verify: the cryptographic half of the admission checks and the root's self-verification. "product" (default): the
        library's host verifier (binding.Verifier, verifier data = the GPU circuit handles' constants/sigmas caps), inner
        proofs verified on all host cores; None: skipped; or {"leaf": fn, "private": fn, "public": fn} with
        fn(list of proofs) -> list of bool. Construction proves the two padding templates, as the reference's artifact
        build does (dummy_proof.rs:104-115, private_batch/circuit/build.rs:165-193)."""
        from . import sharding
        from .binding import Verifier
        self.sharding, self.rank, self.world, self.slots = sharding, rank, world, slots
        self.plan = sharding.aggregation_schedule(num_leaves, slots, world)
        self.mine = self.plan["ranks"][rank]
        self.num_batches = num_leaves // slots
        self.verifiers = {}
        product = verify == "product"
        fns = {} if product or verify is None else dict(verify)
        hk = gpu.lib.qpgpu_ctx_get_hasher(gpu.ctx) if product else 0
        self.leaf_batch = max(1, min(leaf_batch, len(self.mine["leaves"]))) if self.mine["leaves"] else 1
        self.leaf = TemplateProver(gpu, leaf[0], leaf[1], max_batch=self.leaf_batch)
        if product:
            self.verifiers["leaf"] = Verifier(leaf[0], circuit=self.leaf.circ, hasher=hk)
            fns["leaf"] = MemoVerifier(self.verifiers["leaf"].verify_many)
        self.leaf.commit(np.zeros(LEAF_PUBLIC_INPUTS, dtype=np.uint64))
        self.dummy_leaf_proof = self.leaf.prove()
        self.private_batch = max(1, min(private_batch, len(self.mine["private_batches"])))
        self.private = PrivateBatchProver(gpu, private[0], private[1], self.dummy_leaf_proof, slots, fns.get("leaf"), max_batch=self.private_batch)
        self.times = {}
        self.public = None
        self.verify_root = None
        if rank == self.plan["root"]:
            if product:
                self.verifiers["private"] = Verifier(private[0], circuit=self.private.circ, hasher=hk)
                fns["private"] = self.verifiers["private"].verify_many
            self.dummy_private_batch_proof = self.private.prove_dummy_template()
            self.public = PublicBatchProver(gpu, public[0], public[1], self.dummy_private_batch_proof, self.num_batches, slots, fns.get("private"))
            if product:
                self.verifiers["public"] = Verifier(public[0], circuit=self.public.circ, hasher=hk)
                fns["public"] = self.verifiers["public"].verify_many
            self.verify_root = fns.get("public")

    def close(self):
        for p in (self.leaf, self.private, self.public):
            if p is not None:
                p.close()
        for v in self.verifiers.values():
            v.close()
        self.verifiers = {}

    def run(self, dist=None, device=None, blinding_seed=None, keep=None, shuffle_seed=None, aggregator_address=bytes(32), exchange="all"):
        """One pass over the tree. Returns (all leaf proofs, all private-batch proofs, root proof or None).
        exchange: "all" = every level's proof bytes reach every rank (all_gather); "root" = what the next level needs and no
        more: a rank's private batches consume the leaves the same rank proved, so the leaf level exchanges
        nothing, and the private-batch proofs are gathered to the root rank only — the returned lists then hold None for proofs
        this rank neither made nor received.
        blinding_seed / shuffle_seed (32 bytes): make the zero-knowledge salts / the private batches' slot order and dummy
        preimages reproducible (tests; default: operating-system entropy); keep: a dict that receives, per level, this
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
        if exchange == "root" and d is not None:
            leaves = [None] * (self.num_batches * self.slots)
            for i, p_ in zip(ids, mine_leaf):
                leaves[i] = p_
        else:
            leaves = [p for r in self.sharding.gather_proof_bytes(mine_leaf, d, device) for p in r]
        t1 = time.perf_counter()
        mine_priv = []
        # this rank's private batches in lockstep, in runs of consecutive batch numbers (proof j of a run is blinded with
        # seed + first + j, the same salts the one-at-a-time order would draw)
        pb = list(self.mine["private_batches"])
        if isinstance(self.private.verify_leaf, MemoVerifier):     # all of this rank's leaves in one pass over the host cores
            self.private.verify_leaf.forget()
            self.private.verify_leaf([p for b in pb for p in leaves[b * self.slots:(b + 1) * self.slots]])
        k = 0
        while k < len(pb):
            run = [pb[k]]
            while k + len(run) < len(pb) and len(run) < self.private_batch and pb[k + len(run)] == run[-1] + 1:
                run.append(pb[k + len(run)])
            sk = None if shuffle_seed is None else bytes(shuffle_seed[:30]) + bytes([run[0] & 0xFF, 0])
            pis = self.private.commit_many([leaves[b * self.slots:(b + 1) * self.slots] for b in run], seed=sk)
            if blinding_seed is not None:
                self.private.circ.set_blinding_seed(blinding_seed + run[0])
            if keep is not None:
                for j, b in enumerate(run):
                    keep.setdefault("private", []).append((b, pis[j].copy(), self.private.witness(j)))
            mine_priv += self.private.prove_many()
            k += len(run)
        if exchange == "root" and d is not None:
            got = self.sharding.gather_proof_bytes(mine_priv, d, device, root=self.plan["root"])
            if got is not None:
                batches = [p for r in got for p in r]
            else:
                batches = [None] * self.num_batches
                for b, p_ in zip(pb, mine_priv):
                    batches[b] = p_
        else:
            batches = [p for r in self.sharding.gather_proof_bytes(mine_priv, d, device) for p in r]
        t2 = time.perf_counter()
        root = None
        if self.public is not None:
            pis = self.public.commit(batches, aggregator_address)
            if keep is not None:
                keep.setdefault("public", []).append((0, pis.copy(), self.public.witness()))
            root = self.public.prove()
            # ProvingContext::prove_batch verifies what it has just proven before handing it on (aggregator.rs:224-225)
            if self.verify_root is not None and not self.verify_root([root])[0]:
                raise ValueError("public-batch proof failed self-verification")
        t3 = time.perf_counter()
        self.times = {"leaf_level_s": round(t1 - t0, 4), "private_level_s": round(t2 - t1, 4), "public_level_s": round(t3 - t2, 4)}
        return leaves, batches, root
