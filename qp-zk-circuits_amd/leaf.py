"""Host-side mirror of the reference's leaf prover surface over include/qpgpu_leaf.h (ctypes only; the work is in libqpgpu.so):

    WormholeCircuit::new(config).build_prover()   wormhole/circuit/src/circuit.rs:115-152,210-212   -> LeafCircuit(...)
    WormholeProver::commit(&inputs)               wormhole/prover/src/lib.rs:156-163,187-221        -> LeafCircuit.commit(inputs)
    WormholeProver::prove()                       wormhole/prover/src/lib.rs:171-175                -> LeafProver.prove(inputs)

`CircuitInputs` is `LeafInputs` (the C struct qpgpu_leaf_inputs). The circuit is the reference's statement restated on the
library's native builder (csrc/builder.hpp, csrc/leaf_circuit.cpp); see the header for what cannot match the fork offline.
"""
import ctypes

import numpy as np

from .binding import QpGpuError, load_library

MAX_DEPTH, DIGEST_LEN, LT_COUNT, PUBLIC_INPUTS = 16, 110, 299, 21
HASH_HINTS = 12 * 61 + 4 * 16          # QPGPU_LEAF_HASH_HINTS: 61 sponge states + the Merkle walk's running hash per level
FRAGMENT_FULL, FRAGMENT_BLOCK_HEADER, FRAGMENT_UNSPENDABLE_ACCOUNT, FRAGMENT_NULLIFIER, FRAGMENT_FAKE_LEAF = 0, 1, 2, 3, 4
NO_CELL = 0xFFFFFFFFFFFFFFFF
INFO_FIELDS = ("degree_bits", "rows_before_padding", "gates_after_targets", "gates_unspendable_account", "gates_zk_merkle_proof",
               "gates_block_number_range_check", "gates_connect_shared_targets", "rows_arithmetic", "rows_base_sum", "rows_poseidon2",
               "rows_poseidon", "rows_constant", "rows_public_input", "rows_noop", "free_standing_generators", "selector_polynomials")


class LeafInputs(ctypes.Structure):
    """qpgpu_leaf_inputs = CircuitInputs (wormhole/circuit/src/inputs.rs:30-83)."""
    _fields_ = [("asset_id", ctypes.c_uint32), ("output_amount_1", ctypes.c_uint32), ("output_amount_2", ctypes.c_uint32),
                ("volume_fee_bps", ctypes.c_uint32),
                ("nullifier", ctypes.c_uint8 * 32), ("exit_account_1", ctypes.c_uint8 * 32), ("exit_account_2", ctypes.c_uint8 * 32),
                ("block_hash", ctypes.c_uint8 * 32), ("block_number", ctypes.c_uint32),
                ("secret", ctypes.c_uint8 * 32), ("transfer_count", ctypes.c_uint64),
                ("unspendable_account", ctypes.c_uint8 * 32), ("parent_hash", ctypes.c_uint8 * 32), ("state_root", ctypes.c_uint8 * 32),
                ("extrinsics_root", ctypes.c_uint8 * 32), ("digest", ctypes.c_uint8 * DIGEST_LEN), ("input_amount", ctypes.c_uint32),
                ("zk_tree_root", ctypes.c_uint8 * 32), ("zk_merkle_depth", ctypes.c_uint32),
                ("zk_merkle_siblings", ctypes.c_uint8 * (MAX_DEPTH * 3 * 32)), ("zk_merkle_positions", ctypes.c_uint8 * MAX_DEPTH)]

    def set32(self, name, data):
        data = bytes(data)
        assert len(data) == 32, name
        ctypes.memmove(getattr(self, name), data, 32)
        return self

    def get32(self, name):
        return bytes(getattr(self, name))

    def copy(self):
        other = LeafInputs()
        ctypes.memmove(ctypes.byref(other), ctypes.byref(self), ctypes.sizeof(LeafInputs))
        return other


def _lib():
    L = load_library()
    if not getattr(L, "_leaf_sigs", False):
        c = ctypes
        L.qpgpu_leaf_circuit_build.restype = c.c_int
        L.qpgpu_leaf_circuit_build.argtypes = [c.c_uint, c.c_uint, c.c_int, c.c_void_p, c.c_void_p, c.c_size_t, c.POINTER(c.c_size_t),
                                               c.c_void_p, c.c_void_p, c.c_char_p]
        L.qpgpu_leaf_commit.restype = c.c_int
        L.qpgpu_leaf_commit.argtypes = [c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p, c.c_size_t, c.POINTER(c.c_size_t), c.c_void_p, c.c_char_p]
        L.qpgpu_leaf_check_constraints.restype = c.c_int
        L.qpgpu_leaf_check_constraints.argtypes = [c.c_void_p, c.c_char_p]
        L.qpgpu_leaf_unspendable_account.argtypes = [c.c_void_p, c.c_size_t, c.c_char_p, c.c_void_p]
        L.qpgpu_leaf_nullifier.argtypes = [c.c_void_p, c.c_size_t, c.c_char_p, c.c_uint64, c.c_void_p]
        L.qpgpu_leaf_block_hash.argtypes = [c.c_void_p, c.c_size_t, c.c_char_p, c.c_uint32, c.c_char_p, c.c_char_p, c.c_char_p, c.c_char_p, c.c_void_p]
        L.qpgpu_zk_leaf_hash.argtypes = [c.c_char_p, c.c_uint64, c.c_uint32, c.c_uint32, c.c_void_p]
        L.qpgpu_zk_proof_from_unsorted.argtypes = [c.c_char_p, c.c_char_p, c.c_size_t, c.c_void_p, c.c_void_p, c.c_void_p, c.c_char_p]
        L.qpgpu_leaf_circuit_hash_hint_cells.restype = c.c_int
        L.qpgpu_leaf_circuit_hash_hint_cells.argtypes = [c.c_uint, c.c_int, c.c_void_p, c.c_void_p, c.c_size_t, c.POINTER(c.c_size_t), c.c_char_p]
        L.qpgpu_leaf_hash_hints.restype = c.c_int
        L.qpgpu_leaf_hash_hints.argtypes = [c.c_void_p, c.c_void_p, c.c_size_t, c.POINTER(c.c_size_t), c.c_char_p]
        L._leaf_sigs = True
    return L


# ---- the hash-deriving helpers of qpgpu_leaf.h (KAT-pinned Poseidon2 sponge), as bytes -> bytes ----
def unspendable_account(secret):
    out = ctypes.create_string_buffer(32)
    assert _lib().qpgpu_leaf_unspendable_account(None, 0, bytes(secret), out) == 0
    return out.raw


def nullifier(secret, transfer_count):
    out = ctypes.create_string_buffer(32)
    assert _lib().qpgpu_leaf_nullifier(None, 0, bytes(secret), transfer_count, out) == 0
    return out.raw


def block_hash(parent_hash, block_number, state_root, extrinsics_root, zk_tree_root, digest):
    out = ctypes.create_string_buffer(32)
    assert _lib().qpgpu_leaf_block_hash(None, 0, bytes(parent_hash), block_number, bytes(state_root), bytes(extrinsics_root), bytes(zk_tree_root),
                                        bytes(digest), out) == 0
    return out.raw


def zk_leaf_hash(to_account, transfer_count, asset_id, input_amount):
    out = ctypes.create_string_buffer(32)
    assert _lib().qpgpu_zk_leaf_hash(bytes(to_account), transfer_count, asset_id, input_amount, out) == 0
    return out.raw


def zk_proof_from_unsorted(leaf_hash, unsorted_siblings):
    """ZkMerkleProof::from_unsorted: (sorted siblings [depth][3][32] as bytes, positions, root)."""
    depth = len(unsorted_siblings)
    flat = b"".join(bytes(s) for lvl in unsorted_siblings for s in lvl)
    so = ctypes.create_string_buffer(max(96 * depth, 1)); po = ctypes.create_string_buffer(max(depth, 1)); root = ctypes.create_string_buffer(32)
    err = ctypes.create_string_buffer(160)
    if _lib().qpgpu_zk_proof_from_unsorted(bytes(leaf_hash), flat, depth, so, po, root, err) != 0:
        raise ValueError(err.value.decode())
    return so.raw[:96 * depth], list(po.raw[:depth]), root.raw


def dummy_circuit_inputs():
    """build_dummy_circuit_inputs (wormhole/aggregator/src/dummy_proof.rs:58-84,125-170): the CircuitInputs of the dummy leaf the
    batch layers pad with, and the reference bench's input (wormhole/prover/benches/prover.rs:31-42). Zero block hash, outputs,
    nullifier and exit accounts (the sentinel), depth-0 Merkle proof; secret / transfer count / state root / digest are the
    reference's DEFAULT_* constants, the unspendable account is derived from the secret."""
    x = LeafInputs()
    secret = bytes.fromhex("4c8587bd422e01d961acdc75e7d66f6761b7af7c9b1864a492f369c9d6724f05")
    x.asset_id, x.output_amount_1, x.output_amount_2, x.volume_fee_bps, x.block_number = 0, 0, 0, 10, 0
    x.transfer_count, x.input_amount, x.zk_merkle_depth = 4, 100, 0
    for name in ("nullifier", "exit_account_1", "exit_account_2", "block_hash", "parent_hash", "extrinsics_root", "zk_tree_root"):
        x.set32(name, bytes(32))
    x.set32("secret", secret).set32("unspendable_account", unspendable_account(secret))
    x.set32("state_root", bytes.fromhex("ae6e4ff0dca1ef5ede9dccc84365cecfab4e431c6f3086216bc3b819cdf0a893"))
    digest = bytes.fromhex("0806706f775f80e9b6b76b9e017313db7efd561ed0b046152db4e5093e5b040635f53430267be105706f775f0101") + bytes(61) + bytes.fromhex("124fe2")
    assert len(digest) == 110
    ctypes.memmove(x.digest, digest, 110)
    return x


class LeafCircuit:
    """WormholeCircuit::new(config) -> build_prover(): the circuit pack and the wire cell of every logical target. Host only."""

    def __init__(self, fragment=FRAGMENT_FULL, min_degree_bits=0, inner_hasher=0, p2_layout=None):
        L = _lib()
        n = ctypes.c_size_t()
        err = ctypes.create_string_buffer(160)
        lay = None if p2_layout is None else np.ascontiguousarray(p2_layout, dtype=np.uint64)
        layp = None if lay is None else lay.ctypes.data
        rc = L.qpgpu_leaf_circuit_build(fragment, min_degree_bits, inner_hasher, layp, None, 0, ctypes.byref(n), None, None, err)
        if rc != 0:
            raise QpGpuError(rc, err.value.decode())
        self.pack = np.empty(n.value, dtype=np.uint64)
        self.target_map = np.empty(LT_COUNT, dtype=np.uint64)
        info = np.zeros(len(INFO_FIELDS), dtype=np.uint64)
        rc = L.qpgpu_leaf_circuit_build(fragment, min_degree_bits, inner_hasher, layp, self.pack.ctypes.data, self.pack.size, ctypes.byref(n),
                                        self.target_map.ctypes.data, info.ctypes.data, err)
        if rc != 0:
            raise QpGpuError(rc, err.value.decode())
        self.info = {k: int(v) for k, v in zip(INFO_FIELDS, info)}
        self.fragment = fragment
        self._build_args = (min_degree_bits, inner_hasher, layp, lay)
        self._hint_cells = None

    @property
    def hash_hint_cells(self):
        """The cells of the 61 Poseidon2 rows' outputs, call sites in tag order (qpgpu_leaf_circuit_hash_hint_cells); full circuit only."""
        if self._hint_cells is None:
            if self.fragment != FRAGMENT_FULL:
                raise ValueError("hash hints exist for the full leaf circuit only")
            cells = np.empty(HASH_HINTS, dtype=np.uint64)
            n = ctypes.c_size_t(); err = ctypes.create_string_buffer(160)
            rc = _lib().qpgpu_leaf_circuit_hash_hint_cells(self._build_args[0], self._build_args[1], self._build_args[2], cells.ctypes.data, cells.size, ctypes.byref(n), err)
            if rc != 0:
                raise QpGpuError(rc, err.value.decode())
            self._hint_cells = cells[:n.value].copy()
        return self._hint_cells

    def commit(self, inputs, hash_hints=False):
        """WormholeProver::commit: (cells, values, public_inputs[21]); raises ValueError with the reference's message. hash_hints=True
        appends the 732 sponge-state elements of the circuit's hash call sites and the Merkle walk's 16 running hashes, computed on the host (qpgpu_leaf_hash_hints): the same witness,
        with the 61 hash rows generated side by side and checked instead of one after the other."""
        cells = np.empty(LT_COUNT, dtype=np.uint64); values = np.empty(LT_COUNT, dtype=np.uint64); pis = np.empty(PUBLIC_INPUTS, dtype=np.uint64)
        n = ctypes.c_size_t(); err = ctypes.create_string_buffer(160)
        rc = _lib().qpgpu_leaf_commit(ctypes.byref(inputs), self.target_map.ctypes.data, cells.ctypes.data, values.ctypes.data, LT_COUNT,
                                      ctypes.byref(n), pis.ctypes.data, err)
        if rc != 0:
            raise ValueError(err.value.decode())
        if hash_hints:
            hv = np.empty(HASH_HINTS, dtype=np.uint64)
            hn = ctypes.c_size_t()
            if _lib().qpgpu_leaf_hash_hints(ctypes.byref(inputs), hv.ctypes.data, hv.size, ctypes.byref(hn), err) != 0:
                raise ValueError(err.value.decode())
            return np.concatenate([cells[:n.value], self.hash_hint_cells]), np.concatenate([values[:n.value], hv[:hn.value]]), self.public_inputs(pis)
        return cells[:n.value].copy(), values[:n.value].copy(), self.public_inputs(pis)

    def public_inputs(self, pis21):
        """The circuit's own public inputs out of the leaf's 21 (a fragment circuit registers only some of them)."""
        if self.fragment == FRAGMENT_FULL:
            return pis21
        if self.fragment == FRAGMENT_BLOCK_HEADER:
            return np.ascontiguousarray(pis21[16:21])
        if self.fragment == FRAGMENT_NULLIFIER:
            return np.ascontiguousarray(pis21[4:8])
        return np.zeros(0, dtype=np.uint64)


class LeafProver:
    """WormholeProver over a loaded LeafCircuit: commit (host) -> stage s1 on the device -> stages s2..s12."""

    def __init__(self, pkg, gpu, circuit, witness_check=False, hash_hints=False):
        self.pkg, self.gpu, self.circuit, self.hash_hints = pkg, gpu, circuit, hash_hints
        self.circ = pkg.Circuit(gpu, circuit.pack)
        if witness_check:
            self.circ.set_witness_check(True)
        h = pkg.pack_header(circuit.pack)
        self.shape = (h["num_wires"], 1 << h["degree_bits"])
        self.d_wires = gpu.alloc(self.shape[0] * self.shape[1] * 8)

    def generate_witness(self, inputs):
        cells, values, pis = self.circuit.commit(inputs, hash_hints=self.hash_hints)
        self.circ.generate_witness_partial_dev(cells, values, pis, self.d_wires)
        return pis

    def prove(self, inputs):
        pis = self.generate_witness(inputs)
        return self.circ.prove_dev(self.d_wires, pis), pis

    def witness(self):
        return self.d_wires.download().reshape(self.shape)

    def close(self):
        self.d_wires.free(scrub=True)
        self.circ.close()
