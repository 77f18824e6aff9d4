"""Host-side harness of the wrapper circuit that checks the Merkle half of its inner proofs (include/qpgpu_batch.h,
qpgpu_wrapper_circuit_build; csrc/wrapper_circuit.cpp) — ctypes only. Mirrors, as far as it goes, the reference's
add_recursive_verifiers + fill_private_batch_witness (wormhole/aggregator/src/common/recursive.rs:74-102,
private_batch/prover/witness.rs:15-77)."""
import ctypes

import numpy as np

from .binding import QpGpuError, load_library

INFO_FIELDS = ("degree_bits", "rows_before_padding", "targets_per_proof", "query_rounds", "rows_poseidon", "rows_random_access", "rows_base_sum",
               "rows_arithmetic", "rows_constant", "public_inputs", "rows_verification", "_")
NO_CELL = 0xFFFFFFFFFFFFFFFF


def _lib():
    L = load_library()
    if not getattr(L, "_rec_sigs", False):
        c = ctypes
        vp, sz, cp = c.c_void_p, c.c_size_t, c.c_char_p
        L.qpgpu_wrapper_circuit_build.restype = c.c_int
        L.qpgpu_wrapper_circuit_build.argtypes = [vp, sz, vp, sz, c.c_uint, c.c_uint, c.c_uint, c.c_int, vp, sz, c.POINTER(sz), vp, sz, c.POINTER(sz), vp, cp]
        L.qpgpu_batch_fill_proof_targets.argtypes = [vp, sz, vp, vp, sz, sz, vp, sz, sz, cp, vp, vp, sz, c.POINTER(sz), cp]
        L.qpgpu_proof_target_count.argtypes = [vp, sz]; L.qpgpu_proof_target_count.restype = sz
        L.qpgpu_verifier_query_indices.restype = c.c_int
        L.qpgpu_verifier_query_indices.argtypes = [vp, cp, sz, vp, sz, cp]
        L.qpgpu_leaf_map_targets.restype = sz
        L.qpgpu_leaf_map_targets.argtypes = [vp, vp, sz, vp, sz, vp, vp]
        L._rec_sigs = True
    return L


class WrapperCircuit:
    """`num_proofs` proof targets of the inner circuit + the in-circuit Merkle checks of their query rounds. Host only.
    verifier: a binding.Verifier of the INNER circuit (its constants/sigmas cap becomes constants of the wrapper; it also replays
    the transcript for the query indices)."""

    def __init__(self, inner_pack, verifier, num_proofs, num_routed_wires=80, min_degree_bits=0, inner_hasher=0):
        L = _lib()
        self.inner_pack = np.ascontiguousarray(inner_pack, dtype=np.uint64)
        self.verifier, self.num_proofs = verifier, num_proofs
        cap_h = int(self.inner_pack[11])
        cap = np.empty(4 << cap_h, dtype=np.uint64)
        if L.qpgpu_verifier_constants_sigmas_cap(verifier.h, cap.ctypes.data, cap.size) != 0:
            raise QpGpuError(-1, "verifier has no constants/sigmas cap")
        n, m = ctypes.c_size_t(), ctypes.c_size_t()
        err = ctypes.create_string_buffer(200)
        args = (self.inner_pack.ctypes.data, self.inner_pack.size, cap.ctypes.data, cap.size, num_proofs, num_routed_wires, min_degree_bits, inner_hasher)
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

    def query_indices(self, proof):
        out = np.empty(self.Q, dtype=np.uint64)
        err = ctypes.create_string_buffer(200)
        rc = _lib().qpgpu_verifier_query_indices(self.verifier.h, proof, len(proof), out.ctypes.data, out.size, err)
        if rc != 0:
            raise ValueError(err.value.decode())
        return out

    def commit(self, proofs, preimages=None, query_indices=None):
        """fill_private_batch_witness + the query indices: (cells, values, public_inputs) of the wrapper's PartialWitness.
        Raises ValueError with the reference's message for a malformed proof."""
        L = _lib()
        N = self.num_proofs
        pre = np.zeros(4 * N, dtype=np.uint64) if preimages is None else np.ascontiguousarray(preimages, dtype=np.uint64)
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
            qi = self.query_indices(p) if query_indices is None else np.asarray(query_indices[i], dtype=np.uint64)
            t[total + i * self.Q:total + (i + 1) * self.Q] = total + i * self.Q + np.arange(self.Q, dtype=np.uint32)
            v[total + i * self.Q:total + (i + 1) * self.Q] = qi
        cells = np.empty(t.size, dtype=np.uint64); vals = np.empty(t.size, dtype=np.uint64)
        k = L.qpgpu_leaf_map_targets(t.ctypes.data, v.ctypes.data, t.size, self.target_map.ctypes.data, self.target_map.size, cells.ctypes.data, vals.ctypes.data)
        npis = int(self.inner_pack[9])
        pis = np.concatenate([np.frombuffer(p[-8 * npis:], dtype=np.uint64) if npis else np.zeros(0, dtype=np.uint64) for p in proofs])
        return cells[:k].copy(), vals[:k].copy(), pis
