"""The recursive circuits and their provers over the C ABI (include/qpgpu_batch.h: qpgpu_wrapper_circuit_build; csrc/wrapper_circuit.cpp)
— ctypes only. WrapperCircuit = add_recursive_verifiers (wormhole/aggregator/src/common/recursive.rs:74-102) with the flags' parts
(in-circuit transcript, the arithmetic half of verify_proof, the private- / public-batch logic, zero-knowledge blinding) and
fill_private_batch_witness (private_batch/prover/witness.rs:15-77) as its commit(); PrivateBatchProver / PublicBatchProver /
ProvingContext carry the reference's new / commit / prove / aggregate / prove_batch (private_batch/prover/lib.rs:244-343,
public_batch/prover/lib.rs:268-305, aggregator.rs:158-248); AttestingTree is BASELINE configs[4] on them."""
import ctypes

import numpy as np

from .binding import QpGpuError, load_library

INFO_FIELDS = ("degree_bits", "rows_before_padding", "targets_per_proof", "query_rounds", "rows_poseidon", "rows_random_access", "rows_base_sum",
               "rows_arithmetic", "rows_constant", "public_inputs", "rows_verification", "rows_blinding")
NO_CELL = 0xFFFFFFFFFFFFFFFF


def _lib():
    L = load_library()
    if not getattr(L, "_rec_sigs", False):
        c = ctypes
        vp, sz, cp = c.c_void_p, c.c_size_t, c.c_char_p
        L.qpgpu_wrapper_circuit_build.restype = c.c_int
        L.qpgpu_wrapper_circuit_build.argtypes = [vp, sz, vp, sz, c.c_uint, c.c_uint, c.c_uint, c.c_int, c.c_uint, vp, sz, c.POINTER(sz), vp, sz, c.POINTER(sz), vp, cp]
        L.qpgpu_batch_fill_proof_targets.argtypes = [vp, sz, vp, vp, sz, sz, vp, sz, sz, cp, vp, vp, sz, c.POINTER(sz), cp]
        L.qpgpu_proof_target_count.argtypes = [vp, sz]; L.qpgpu_proof_target_count.restype = sz
        L.qpgpu_verifier_query_indices.restype = c.c_int
        L.qpgpu_verifier_query_indices.argtypes = [vp, cp, sz, vp, sz, cp]
        L.qpgpu_random_field_elements.argtypes = [cp, vp, sz, cp]
        L.qpgpu_leaf_map_targets.restype = sz
        L.qpgpu_leaf_map_targets.argtypes = [vp, vp, sz, vp, sz, vp, vp]
        L._rec_sigs = True
    return L


class WrapperCircuit:
    """`num_proofs` proof targets of the inner circuit + the in-circuit Merkle checks of their query rounds. Host only.
    verifier: a binding.Verifier of the INNER circuit (its constants/sigmas cap becomes constants of the wrapper; it also replays
    the transcript for the query indices)."""

    FLAGS = {None: 0, "private_batch": 2, "public_batch": 4}          # QPGPU_WRAPPER_PRIVATE_BATCH / _PUBLIC_BATCH

    def __init__(self, inner_pack, verifier, num_proofs, num_routed_wires=80, min_degree_bits=0, inner_hasher=0, transcript=True, logic=None, verify=False, zero_knowledge=False):
        """transcript=True (QPGPU_WRAPPER_TRANSCRIPT): the inner proofs' Fiat-Shamir transcripts are replayed in-circuit, the query
        indices are derived there and the proof-of-work response is range-checked; False: the query indices are witness inputs.
        logic: None (inner public inputs forwarded), "private_batch" (build_private_batch_constraints over leaf proofs) or
        "public_batch" (build_public_batch_constraints over private-batch proofs): the layer's own constraints and public inputs.
        verify=True (QPGPU_WRAPPER_VERIFY): the arithmetic half of verify_proof in-circuit too (openings against the vanishing
        polynomial at zeta, FRI consistency) — the wrapper then enforces everything the host verifier checks on an inner proof.
        zero_knowledge=True (QPGPU_WRAPPER_ZERO_KNOWLEDGE): the circuit is built with CircuitBuilder::blind's rows and proven with
        salted Merkle leaves; commit draws the blinding rows' random wires (blinding_seed: 32 bytes for reproducible tests, None:
        operating-system entropy)."""
        L = _lib()
        self.logic, self.verify, self.zero_knowledge = logic, verify, zero_knowledge
        self.inner_pack = np.ascontiguousarray(inner_pack, dtype=np.uint64)
        self.verifier, self.num_proofs, self.transcript = verifier, num_proofs, transcript
        cap_h = int(self.inner_pack[11])
        cap = np.empty(4 << cap_h, dtype=np.uint64)
        if L.qpgpu_verifier_constants_sigmas_cap(verifier.h, cap.ctypes.data, cap.size) != 0:
            raise QpGpuError(-1, "verifier has no constants/sigmas cap")
        n, m = ctypes.c_size_t(), ctypes.c_size_t()
        err = ctypes.create_string_buffer(200)
        args = (self.inner_pack.ctypes.data, self.inner_pack.size, cap.ctypes.data, cap.size, num_proofs, num_routed_wires, min_degree_bits, inner_hasher,
                (1 if transcript else 0) | self.FLAGS[logic] | (8 if verify else 0) | (16 if zero_knowledge else 0))
        rc = L.qpgpu_wrapper_circuit_build(*args, None, 0, ctypes.byref(n), None, 0, ctypes.byref(m), None, err)
        if rc != 0:
            raise QpGpuError(rc, err.value.decode())
        self.pack = np.empty(n.value, dtype=np.uint64)
        self.target_map = np.empty(m.value, dtype=np.uint64)
        info = np.zeros(len(INFO_FIELDS), dtype=np.uint64)
        rc = L.qpgpu_wrapper_circuit_build(*args, self.pack.ctypes.data, self.pack.size, ctypes.byref(n), self.target_map.ctypes.data, self.target_map.size,
                                           ctypes.byref(m), info.ctypes.data, err)
        if rc != 0:
            raise QpGpuError(rc, err.value.decode())
        self.info = {k: int(v) for k, v in zip(INFO_FIELDS, info) if k != "_"}
        self.T, self.Q = self.info["targets_per_proof"], self.info["query_rounds"]
        self.blinding_cells = self.target_map[num_proofs * (self.T + 4 + self.Q):].copy()       # cells, not logical ids
        self.target_map = self.target_map[:num_proofs * (self.T + 4 + self.Q)]

    def query_indices(self, proof):
        out = np.empty(self.Q, dtype=np.uint64)
        err = ctypes.create_string_buffer(200)
        rc = _lib().qpgpu_verifier_query_indices(self.verifier.h, proof, len(proof), out.ctypes.data, out.size, err)
        if rc != 0:
            raise ValueError(err.value.decode())
        return out

    def commit(self, proofs, preimages=None, query_indices=None, aggregator_address=None, public_inputs=None, blinding_seed=None, device_blinding=False, derive_public_inputs=False):
        """fill_private_batch_witness / fill_public_batch_witness + the query indices: (cells, values, public_inputs) of the
        wrapper's PartialWitness. Raises ValueError with the reference's message for a malformed proof. preimages: the dummy-nullifier
        preimages (private batch, N x 4 felts); aggregator_address: 32 bytes (public batch). With a batch logic the public inputs
        are the ones the layer's circuit computes (aggregation.private_batch_outputs / public_batch_outputs on the host; QpGpuError
        -4 there when the slots violate a constraint the host restatement sees); public_inputs overrides them (tests: what the
        CIRCUIT says to slots the host restatement refuses). device_blinding=True (zero-knowledge circuits): the blinding cells are
        appended to the cell list WITHOUT values — Circuit.generate_witness_partial_batch_blinded_dev draws them on the device
        (n_blinding = self.blinding_cells.size). derive_public_inputs=True: the third element is None — the public inputs are
        what the circuit computes (generate_wrapper_witnesses reads them back from the device witness), as in plonky2's prove()."""
        L = _lib()
        N = self.num_proofs
        pre = np.zeros(4 * N, dtype=np.uint64) if preimages is None else np.ascontiguousarray(preimages, dtype=np.uint64).reshape(-1).copy()
        if self.logic == "public_batch":
            addr = bytes(32) if aggregator_address is None else bytes(aggregator_address)
            pre[:4] = np.frombuffer(addr, dtype=np.uint64)
        bufs = [ctypes.create_string_buffer(bytes(p), len(p)) for p in proofs]
        ptrs = (ctypes.c_void_p * len(proofs))(*[ctypes.addressof(b) for b in bufs])
        lens = (ctypes.c_size_t * len(proofs))(*[len(p) for p in proofs])
        cnt = ctypes.c_size_t(); err = ctypes.create_string_buffer(400)
        total = N * (self.T + 4)
        t = np.empty(total + N * self.Q, dtype=np.uint32); v = np.empty(total + N * self.Q, dtype=np.uint64)
        rc = L.qpgpu_batch_fill_proof_targets(self.inner_pack.ctypes.data, self.inner_pack.size, ptrs, lens, len(proofs), N, pre.ctypes.data, pre.size // 4, N,
                                              b"leaf proof", t.ctypes.data, v.ctypes.data, total, ctypes.byref(cnt), err)
        if rc != 0:
            raise ValueError(err.value.decode())
        assert cnt.value == total
        for i, p in enumerate(proofs):
            if self.transcript:       # derived in-circuit: the targets have no cell, any value is dropped by the target map
                qi = np.zeros(self.Q, dtype=np.uint64)
            else:
                qi = self.query_indices(p) if query_indices is None else np.asarray(query_indices[i], dtype=np.uint64)
            t[total + i * self.Q:total + (i + 1) * self.Q] = total + i * self.Q + np.arange(self.Q, dtype=np.uint32)
            v[total + i * self.Q:total + (i + 1) * self.Q] = qi
        cells = np.empty(t.size, dtype=np.uint64); vals = np.empty(t.size, dtype=np.uint64)
        k = L.qpgpu_leaf_map_targets(t.ctypes.data, v.ctypes.data, t.size, self.target_map.ctypes.data, self.target_map.size, cells.ctypes.data, vals.ctypes.data)
        npis = int(self.inner_pack[9])
        pis = np.concatenate([np.frombuffer(p[-8 * npis:], dtype=np.uint64) if npis else np.zeros(0, dtype=np.uint64) for p in proofs])
        if derive_public_inputs:
            pis = None
        elif public_inputs is not None:
            pis = np.ascontiguousarray(public_inputs, dtype=np.uint64)
        elif self.logic == "private_batch":
            from . import aggregation
            pis = aggregation.private_batch_outputs(pis.reshape(N, npis), pre.reshape(N, 4))
        elif self.logic == "public_batch":
            from . import aggregation
            pis = aggregation.public_batch_outputs(pis.reshape(N, npis), (npis - 8) // 21, addr)
        cells, vals = cells[:k].copy(), vals[:k].copy()
        if self.blinding_cells.size and device_blinding:
            cells = np.concatenate([cells, self.blinding_cells])
        elif self.blinding_cells.size:        # RandomValueGenerator: one fresh field element per blinding wire, per proof
            rnd = np.empty(self.blinding_cells.size, dtype=np.uint64)
            if L.qpgpu_random_field_elements(blinding_seed, rnd.ctypes.data, rnd.size, err) != 0:
                raise QpGpuError(-1, err.value.decode())
            cells, vals = np.concatenate([cells, self.blinding_cells]), np.concatenate([vals, rnd])
        return cells, vals, pis


def generate_wrapper_witnesses(circ, wrapper, commits, d_wires, seeds=None):
    """Stage s1 for a lockstep batch of one wrapper circuit's commits: values from the host, the blinding wires of a
    zero-knowledge wrapper drawn on the device when the commits were made with device_blinding=True. Returns the status list."""
    cells = commits[0][0]
    vals = np.stack([c[1] for c in commits])
    pis = None if commits[0][2] is None else np.stack([c[2] for c in commits])
    nb = cells.size - vals.shape[1]
    if nb:
        return circ.generate_witness_partial_batch_blinded_dev(cells, vals, pis, d_wires, nb, seeds)
    return circ.generate_witness_partial_batch_dev(cells, vals, pis, d_wires)


def _ensure_proof_public_input_len(proofs, proof_bytes, expected_len, label):
    """ensure_proof_public_input_len (wormhole/aggregator/src/common/utils.rs:540-556) at the API boundary, for EVERY proof of a batch (a
    full batch too: aggregator_tests.rs:395-412). A serialized proof of a circuit has one length; the public inputs are its tail, so a
    proof whose public-input list is short or long is a proof of another length."""
    for p in proofs:
        if len(p) != proof_bytes:
            got = expected_len + (len(p) - proof_bytes) // 8 if (len(p) - proof_bytes) % 8 == 0 else -1
            raise ValueError("%s public input length mismatch: expected %d, got %s (%d bytes where the circuit's proofs have %d)"
                             % (label, expected_len, got if got >= 0 else "a ragged tail", len(p), proof_bytes))


class PrivateBatchProver:
    """PrivateBatchProver::{new, commit, prove} (wormhole/aggregator/src/private_batch/prover/lib.rs:244-343) over the restated
    circuits: the private-batch circuit is a WrapperCircuit over the leaf circuit with the layer's logic and the complete
    in-circuit verifier; the dummy leaf the slots are padded with is proven here from build_dummy_circuit_inputs (generate_dummy_proof,
    dummy_proof.rs:104-115). commit: the reference's admission checks (qpgpu_private_batch_preflight, then every supplied proof
    against the leaf verifier), padding + uniform shuffle + one dummy-nullifier preimage per slot (qpgpu_private_batch_arrange),
    fill_private_batch_witness (+ the blinding rows' random wires of the zero-knowledge circuit). prove: stage s1 and s2..s12 on
    the device (Merkle leaves salted)."""

    def __init__(self, pkg, gpu, leaf_circuit, num_leaf_proofs, verify=True, leaf_prover=None, zero_knowledge=True, num_routed_wires=60):
        """zero_knowledge / num_routed_wires: wormhole_private_batch_circuit_config (common/src/circuit.rs:396-402) — the one
        zero-knowledge layer of the stack, 60 routed wires."""
        from . import aggregation, leaf as leaf_mod
        self.pkg, self.gpu, self.N, self.A = pkg, gpu, num_leaf_proofs, aggregation
        self.own_leaf_prover = leaf_prover is None
        self.leaf_prover = leaf_prover or leaf_mod.LeafProver(pkg, gpu, leaf_circuit)
        self.leaf_verifier = pkg.Verifier(leaf_circuit.pack, circuit=self.leaf_prover.circ)
        self.circuit = WrapperCircuit(leaf_circuit.pack, self.leaf_verifier, num_leaf_proofs, num_routed_wires=num_routed_wires, logic="private_batch", verify=verify,
                                      zero_knowledge=zero_knowledge)
        self.circ = pkg.Circuit(gpu, self.circuit.pack)
        self.verifier = pkg.Verifier(self.circuit.pack, circuit=self.circ)
        self.d_wires = gpu.alloc(8 * (135 << self.circuit.info["degree_bits"]))
        # verify_dummy_leaf_template: the sentinel, then the cryptographic check
        self.dummy_leaf_proof = self.leaf_prover.prove(leaf_mod.dummy_circuit_inputs())[0]
        dp = aggregation.proof_public_inputs(self.dummy_leaf_proof, 21)
        aggregation._call(aggregation._lib().qpgpu_dummy_leaf_template_check, dp.ctypes.data, dp.size)
        if not self.leaf_verifier.verify(self.dummy_leaf_proof):
            raise ValueError("dummy leaf proof template failed verification")
        self.committed, self.arrangement = None, None

    def close(self):
        self.d_wires.free(scrub=True)
        for x in (self.verifier, self.circ, self.leaf_verifier):
            x.close()
        if self.own_leaf_prover:
            self.leaf_prover.close()

    def _fill(self, slot_proofs, preimages, blinding_seed=None):
        self.committed = self.circuit.commit(slot_proofs, preimages=preimages, device_blinding=True, derive_public_inputs=True)
        self.blinding_seed = blinding_seed
        return self

    def commit(self, leaf_proofs, seed=None):
        """leaf_proofs: 1..N serialized leaf proofs. seed: 32 bytes for a reproducible arrangement and blinding (None: OS entropy).
        ValueError with the reference's message for a batch the preflight refuses or a proof the leaf verifier rejects."""
        A, N = self.A, self.N
        _ensure_proof_public_input_len(leaf_proofs, len(self.dummy_leaf_proof), 21, "leaf proof")
        rows = np.stack([A.proof_public_inputs(p, 21) for p in leaf_proofs]) if len(leaf_proofs) else np.zeros((0, 21), dtype=np.uint64)
        rows = np.ascontiguousarray(rows, dtype=np.uint64)
        A._call(A._lib().qpgpu_private_batch_preflight, rows.ctypes.data if rows.size else None, rows.shape[0], N)
        for i, ok in enumerate(self.leaf_verifier.verify_many(list(leaf_proofs))):
            if not ok:
                raise ValueError("leaf proof %d failed verification against the pinned leaf verifier" % i)
        src = np.zeros(N, dtype=np.uint32); pre = np.zeros(4 * N, dtype=np.uint64)
        A._call(A._lib().qpgpu_private_batch_arrange, rows.shape[0], N, seed, src.ctypes.data, pre.ctypes.data)
        self.arrangement = (src, pre.reshape(N, 4))
        return self._fill([self.dummy_leaf_proof if k == 0xFFFFFFFF else leaf_proofs[k] for k in src.tolist()], pre.reshape(N, 4),
                          None if seed is None else bytes(b ^ 0x5A for b in seed))

    def prove(self):
        if self.committed is None:
            raise ValueError("prove() before commit()")
        commit = self.committed
        self.committed = None
        st = generate_wrapper_witnesses(self.circ, self.circuit, [commit], self.d_wires, self.blinding_seed)
        if any(st):
            raise QpGpuError(-4, self.gpu.last_error())
        # the public inputs are the circuit's: read out of the witness, as ProverCircuitData::prove does
        return self.circ.prove_dev(self.d_wires, self.circ.witness_public_inputs_dev(self.d_wires)[0])

    def aggregate(self, leaf_proofs, seed=None):
        """PrivateBatchProver::aggregate (private_batch/prover/lib.rs:336-343): commit + prove."""
        return self.commit(leaf_proofs, seed=seed).prove()

    def prove_dummy_template(self):
        """generate_dummy_private_batch_proof (private_batch/circuit/build.rs:165-193): the all-dummy private-batch proof the
        public level pads with — built from explicit dummy leaves, never through commit (which refuses an all-dummy batch)."""
        A, N = self.A, self.N
        src = np.zeros(N, dtype=np.uint32); pre = np.zeros(4 * N, dtype=np.uint64)
        A._call(A._lib().qpgpu_private_batch_arrange, 1, N, None, src.ctypes.data, pre.ctypes.data)      # only the preimages are used
        return self._fill([self.dummy_leaf_proof] * N, pre.reshape(N, 4)).prove()


class PublicBatchProver:
    """PublicBatchProver::{new, commit, prove} (wormhole/aggregator/src/public_batch/prover/lib.rs:268-305): the public-batch
    circuit is a WrapperCircuit over the private-batch circuit with the layer's logic and the complete in-circuit verifier;
    admission checks on the supplied private-batch proofs, order-preserving padding with the all-dummy private-batch template
    (no shuffle), the aggregator address as a witness input that becomes the first four public inputs."""

    def __init__(self, pkg, gpu, private_prover, num_private_batch_proofs, verify=True):
        from . import aggregation
        self.pkg, self.gpu, self.M, self.N, self.A = pkg, gpu, num_private_batch_proofs, private_prover.N, aggregation
        self.private_verifier = private_prover.verifier
        self.inner_len = aggregation.private_batch_pi_len(self.N)
        self.circuit = WrapperCircuit(private_prover.circuit.pack, self.private_verifier, self.M, logic="public_batch", verify=verify)
        self.circ = pkg.Circuit(gpu, self.circuit.pack)
        self.verifier = pkg.Verifier(self.circuit.pack, circuit=self.circ)
        self.d_wires = gpu.alloc(8 * (135 << self.circuit.info["degree_bits"]))
        # verify_dummy_private_batch_template: the sentinel, then the cryptographic check
        self.dummy_private_batch_proof = private_prover.prove_dummy_template()
        dp = aggregation.proof_public_inputs(self.dummy_private_batch_proof, self.inner_len)
        aggregation._call(aggregation._lib().qpgpu_dummy_private_batch_template_check, dp.ctypes.data, dp.size)
        if not self.private_verifier.verify(self.dummy_private_batch_proof):
            raise ValueError("dummy private-batch proof template failed verification")
        self.committed = None

    def close(self):
        self.d_wires.free(scrub=True)
        self.verifier.close(); self.circ.close()

    def commit(self, private_batch_proofs, aggregator_address=bytes(32)):
        A = self.A
        _ensure_proof_public_input_len(private_batch_proofs, len(self.dummy_private_batch_proof), self.inner_len, "private-batch proof")
        rows = np.stack([A.proof_public_inputs(p, self.inner_len) for p in private_batch_proofs]) if len(private_batch_proofs) else np.zeros((0, self.inner_len), dtype=np.uint64)
        rows = np.ascontiguousarray(rows, dtype=np.uint64)
        A._call(A._lib().qpgpu_public_batch_preflight, rows.ctypes.data if rows.size else None, rows.shape[0], self.inner_len, self.M)
        for i, ok in enumerate(self.private_verifier.verify_many(list(private_batch_proofs))):
            if not ok:
                raise ValueError("private-batch proof %d failed verification against the pinned private-batch verifier" % i)
        padded = list(private_batch_proofs) + [self.dummy_private_batch_proof] * (self.M - len(private_batch_proofs))
        self.committed = self.circuit.commit(padded, aggregator_address=aggregator_address, derive_public_inputs=True)
        return self

    def prove(self):
        if self.committed is None:
            raise ValueError("prove() before commit()")
        commit = self.committed
        self.committed = None
        st = generate_wrapper_witnesses(self.circ, self.circuit, [commit], self.d_wires)
        if any(st):
            raise QpGpuError(-4, self.gpu.last_error())
        return self.circ.prove_dev(self.d_wires, self.circ.witness_public_inputs_dev(self.d_wires)[0])


class ProvingContext:
    """ProvingContext (wormhole/aggregator/src/aggregator.rs:158-248): what a miner's proving worker holds — the public-batch
    prover pinned at construction and the aggregator's address. prove_batch: admission checks, order-preserving padding, proof,
    then the proof is checked against the pinned verifier AND its exposed aggregator address before it is handed back."""

    def __init__(self, public_prover, aggregator_address):
        self.prover, self.aggregator_address = public_prover, bytes(aggregator_address)

    def prove_batch(self, private_batch_proofs):
        proof = self.prover.commit(list(private_batch_proofs), aggregator_address=self.aggregator_address).prove()
        self.verify(proof)
        return proof

    def verify(self, proof):
        from . import aggregation
        n = aggregation.public_batch_pi_len(self.prover.M, self.prover.N)
        if len(proof) < 8 * n:
            raise ValueError("public-batch proof: shorter than its public inputs")
        pis = aggregation.proof_public_inputs(proof, n)
        got = pis[:4].astype("<u8").tobytes()
        if got != self.aggregator_address:
            raise ValueError("public-batch proof aggregator address %s does not match configured aggregator address %s" % (got.hex(), self.aggregator_address.hex()))
        if not self.prover.verifier.verify(proof):
            raise ValueError("public-batch aggregated proof verification failed: " + getattr(self.prover.verifier, "reason", ""))


class AttestingTree:
    """BASELINE configs[4]'s shape with circuits that check something: `batches` x `per_batch` leaf proofs of the restated
    Wormhole leaf circuit (from CircuitInputs), one first-level wrapper per `per_batch` leaves, one second-level wrapper over the
    first-level proofs (wormhole/aggregator/src/aggregator.rs:187-227's two layers). Every wrapper is a WrapperCircuit: it checks
    the Merkle half of each inner proof in-circuit, replays the inner proofs' transcripts in-circuit (query indices derived, proof
    of work checked), with verify (the default) evaluates the openings against the vanishing polynomial and the FRI consistency
    arithmetic in-circuit as well — everything VerifierCircuitData::verify checks — and, with batch_logic (the default), carries its
    layer's own constraints: the first level is the private-batch circuit's logic over its leaves' public inputs, the second the
    public-batch circuit's over the first level's — the root proof's public inputs are a PublicBatchPublicInputs. The leaves
    of one tree must then be what the reference's layers accept (real spends of ONE block, dummies elsewhere). One lockstep batch
    per level and rank."""

    def __init__(self, pkg, gpu, per_batch=8, batches=8, leaf_min_degree_bits=0, rank=0, world=1, batch_logic=True, aggregator_address=bytes(32), seed=1, verify=True, zero_knowledge=False):
        """rank / world: with several ranks (one per GPU) a rank proves the leaves and the first-level wrapper of batches
        b = rank, rank + world, ..; the first-level proofs travel to rank 0, which proves the second level."""
        self.pkg, self.gpu, self.per_batch, self.batches = pkg, gpu, per_batch, batches
        self.rank, self.world = rank, world
        self.my_batches = list(range(rank, batches, world))
        L = pkg.leaf
        self.leaf = L.LeafCircuit(min_degree_bits=leaf_min_degree_bits)
        n_leaves = per_batch * max(1, len(self.my_batches))
        self.leaf_circ = pkg.Circuit(gpu, self.leaf.pack, max_batch=n_leaves)
        self.leaf_ver = pkg.Verifier(self.leaf.pack, circuit=self.leaf_circ)
        self.batch_logic, self.aggregator_address, self.verify = batch_logic, bytes(aggregator_address), verify
        self.seed = seed
        # zero_knowledge: the first level as the reference configures its private layer (blinding rows, salted leaves, 60 routed wires)
        self.zero_knowledge = zero_knowledge
        self.w1 = WrapperCircuit(self.leaf.pack, self.leaf_ver, per_batch, num_routed_wires=60 if zero_knowledge else 80, logic="private_batch" if batch_logic else None,
                                 verify=verify, zero_knowledge=zero_knowledge)
        self.w1_circ = pkg.Circuit(gpu, self.w1.pack, max_batch=max(1, len(self.my_batches)))
        self.w1_ver = pkg.Verifier(self.w1.pack, circuit=self.w1_circ)
        self.w2 = WrapperCircuit(self.w1.pack, self.w1_ver, batches, logic="public_batch" if batch_logic else None, verify=verify)
        self.w2_circ = pkg.Circuit(gpu, self.w2.pack)
        self.w2_ver = pkg.Verifier(self.w2.pack, circuit=self.w2_circ)
        self.words = [135 << c.info["degree_bits"] for c in (self.leaf, self.w1, self.w2)]
        self.d_wires = gpu.alloc(8 * max(n_leaves * self.words[0], max(1, len(self.my_batches)) * self.words[1], self.words[2]))
        self.times = {}

    def preimages(self, batch):
        """The dummy-nullifier preimages of first-level batch `batch` (per_batch x 4 felts). The reference draws them from the
        operating system per proof (private_batch/prover/lib.rs); here they are a function of (seed, batch) so that a test can
        recompute the public inputs any rank's batch must carry."""
        return np.random.default_rng([self.seed, batch]).integers(0, 1 << 63, (self.per_batch, 4), dtype=np.uint64)

    def expected_root_public_inputs(self, leaf_public_inputs):
        """The public inputs the root proof must carry for these leaves (batches * per_batch rows of 21 felts, in leaf order),
        from the HOST restatements of the two layers' logic (aggregation.private_batch_outputs per batch, then
        public_batch_outputs) — or, without batch_logic, the leaves' public inputs forwarded."""
        rows = np.ascontiguousarray(leaf_public_inputs, dtype=np.uint64).reshape(self.batches, self.per_batch, 21)
        if not self.batch_logic:
            return rows.reshape(-1)
        from . import aggregation
        inner = np.stack([aggregation.private_batch_outputs(rows[b], self.preimages(b)) for b in range(self.batches)])
        return aggregation.public_batch_outputs(inner, self.per_batch, self.aggregator_address)

    def close(self):
        self.d_wires.free(scrub=True)
        for x in (self.leaf_ver, self.w1_ver, self.w2_ver, self.leaf_circ, self.w1_circ, self.w2_circ):
            x.close()

    def _level(self, circ, words, cells, values, pis):
        st = circ.generate_witness_partial_batch_dev(cells, values, pis, self.d_wires)
        if any(st):
            raise QpGpuError(-4, self.gpu.last_error())
        nb = len(values)
        return circ.prove_batch_dev([self.d_wires.ptr + 8 * k * words for k in range(nb)], list(pis))

    def run(self, inputs, dist=None, device=None):
        """inputs: batches * per_batch LeafInputs (every rank holds the list; a rank reads its own batches' inputs). Returns
        (this rank's leaf proofs in batch order, the first-level proofs — all of them on rank 0, this rank's otherwise —, the root
        proof on rank 0 / None elsewhere). dist / device: torch.distributed and the device of its tensors when world > 1."""
        import time
        assert len(inputs) == self.per_batch * self.batches
        t0 = time.perf_counter()
        mine = [x for b in self.my_batches for x in inputs[b * self.per_batch:(b + 1) * self.per_batch]]
        leaves, level1 = [], []
        if mine:
            com = [self.leaf.commit(x) for x in mine]
            leaves = self._level(self.leaf_circ, self.words[0], com[0][0], np.stack([c[1] for c in com]), np.stack([c[2] for c in com]))
        t1 = time.perf_counter()
        if mine:
            pre = [self.preimages(b) for b in self.my_batches]
            # fill_private_batch_witness per batch on host threads (the ctypes calls release the interpreter lock): 24 664 assignments
            # per inner proof and, for the zero-knowledge circuit, ~0.6 M blinding values per batch from ChaCha20
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(max_workers=min(8, len(self.my_batches))) as ex:
                com1 = list(ex.map(lambda k: self.w1.commit(leaves[k * self.per_batch:(k + 1) * self.per_batch], preimages=pre[k], device_blinding=True, derive_public_inputs=True), range(len(self.my_batches))))
            st = generate_wrapper_witnesses(self.w1_circ, self.w1, com1, self.d_wires)
            if any(st):
                raise QpGpuError(-4, self.gpu.last_error())
            pis1 = self.w1_circ.witness_public_inputs_dev(self.d_wires, len(com1))        # the circuit's own outputs, as prove() reads them
            level1 = self.w1_circ.prove_batch_dev([self.d_wires.ptr + 8 * k * self.words[1] for k in range(len(com1))], list(pis1))
        if self.world > 1:       # the one exchange of the tree: first-level proof bytes to the rank that proves the second level
            from . import sharding
            got = sharding.gather_proof_bytes(level1, dist, device, root=0)
            if got is not None:
                level1 = [None] * self.batches
                for r, lst in enumerate(got):
                    for k, p in enumerate(lst):
                        level1[r + k * self.world] = p
        t2 = time.perf_counter()
        root = None
        if self.rank == 0:
            c2 = self.w2.commit(level1, aggregator_address=self.aggregator_address, derive_public_inputs=True)
            st = generate_wrapper_witnesses(self.w2_circ, self.w2, [c2], self.d_wires)
            if any(st):
                raise QpGpuError(-4, self.gpu.last_error())
            root = self.w2_circ.prove_dev(self.d_wires, self.w2_circ.witness_public_inputs_dev(self.d_wires)[0])
        t3 = time.perf_counter()
        self.times = {"leaf_level_s": round(t1 - t0, 4), "first_level_s": round(t2 - t1, 4), "second_level_s": round(t3 - t2, 4)}
        return leaves, level1, root
