"""ctypes binding of include/qpgpu.h. Thin: argument marshalling and error mapping only."""
import ctypes
import os

import numpy as np

P = 0xFFFFFFFF00000001
MULT_GEN = 14293326489335486720

NTT_FORWARD = 0
NTT_INVERSE = 1
NTT_OUT_BITREV = 2

_HERE = os.path.dirname(os.path.abspath(__file__))


def lib_path():
    return os.path.join(_HERE, "libqpgpu.so")


class QpGpuError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"qpgpu error {code}: {msg}")
        self.code = code


_lib = None


def load_library():
    """Load libqpgpu.so. Raises if it has not been built (no CPU fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise FileNotFoundError(f"{path} missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
    lib = ctypes.CDLL(path)
    c = ctypes
    vp, u64p = c.c_void_p, c.c_void_p
    sigs = {
        "qpgpu_version": (c.c_char_p, []),
        "qpgpu_ctx_pci_bus_id": (c.c_int, [c.c_void_p, c.c_char_p, c.c_size_t]),
        "qpgpu_verifier_create": (c.c_int, [vp, c.c_size_t, vp, c.c_size_t, c.c_int, vp, c.c_size_t, c.POINTER(vp), c.c_char_p]),
        "qpgpu_verifier_free": (None, [vp]),
        "qpgpu_verifier_proof_size": (c.c_size_t, [vp]),
        "qpgpu_verifier_constants_sigmas_cap": (c.c_int, [vp, vp, c.c_size_t]),
        "qpgpu_verifier_verify": (c.c_int, [vp, c.c_char_p, c.c_size_t, c.c_char_p]),
        "qpgpu_verifier_verify_many": (c.c_int, [vp, vp, vp, c.c_size_t, c.c_uint, vp, c.c_char_p]),
        "qpgpu_ctx_create": (c.c_int, [c.c_int, c.POINTER(vp)]),
        "qpgpu_ctx_destroy": (None, [vp]),
        "qpgpu_last_error": (c.c_char_p, [vp]),
        "qpgpu_ctx_set_stream": (c.c_int, [vp, vp]),
        "qpgpu_sync": (c.c_int, [vp]),
        "qpgpu_profile_enable": (c.c_int, [vp, c.c_int]),
        "qpgpu_profile_read": (c.c_int, [vp, c.c_char_p, c.POINTER(c.c_double), c.POINTER(c.c_uint64)]),
        "qpgpu_malloc": (c.c_int, [vp, c.c_size_t, c.POINTER(vp)]),
        "qpgpu_free": (c.c_int, [vp, vp]),
        "qpgpu_free_scrubbed": (c.c_int, [vp, vp, c.c_size_t]),
        "qpgpu_memcpy_h2d": (c.c_int, [vp, vp, vp, c.c_size_t]),
        "qpgpu_memcpy_d2h": (c.c_int, [vp, vp, vp, c.c_size_t]),
        "qpgpu_memcpy_d2d": (c.c_int, [vp, vp, vp, c.c_size_t]),
        "qpgpu_ntt_batch": (c.c_int, [vp, u64p, c.c_uint, c.c_size_t, c.c_int, c.c_uint64]),
        "qpgpu_ntt_batch_dev": (c.c_int, [vp, u64p, u64p, c.c_uint, c.c_size_t, c.c_int, c.c_uint64]),
        "qpgpu_lde_batch_dev": (c.c_int, [vp, u64p, u64p, c.c_uint, c.c_uint, c.c_size_t, c.c_int, c.c_uint64]),
        "qpgpu_poseidon_permute_dev": (c.c_int, [vp, u64p, c.c_size_t]),
        "qpgpu_poseidon2_hash_pad10_dev": (c.c_int, [vp, u64p, c.c_size_t, u64p, c.c_size_t, c.c_size_t, u64p]),
        "qpgpu_poseidon2_qp_params": (c.c_size_t, [u64p, c.c_size_t]),
        "qpgpu_poseidon2_hash_pad10": (c.c_int, [u64p, c.c_size_t, u64p, c.c_size_t, u64p]),
        "qpgpu_crash_trace_armed": (c.c_int, []),
        "qpgpu_circuit_num_public_inputs": (c.c_size_t, [vp]),
        "qpgpu_merkle_digest_count": (c.c_size_t, [c.c_uint, c.c_uint]),
        "qpgpu_merkle_build_dev": (c.c_int, [vp, u64p, c.c_uint64, c.c_uint32, c.c_uint, c.c_uint, u64p, u64p]),
        "qpgpu_merkle_build_rows_dev": (c.c_int, [vp, u64p, c.c_uint32, c.c_uint, c.c_uint, u64p, u64p]),
        "qpgpu_circuit_load": (c.c_int, [vp, u64p, c.c_size_t, c.POINTER(vp)]),
        "qpgpu_circuit_load_batch": (c.c_int, [vp, u64p, c.c_size_t, c.c_uint, c.POINTER(vp)]),
        "qpgpu_circuit_max_batch": (c.c_uint, [vp]),
        "qpgpu_circuit_scrub": (c.c_int, [vp]),
        "qpgpu_prove_batch_dev": (c.c_int, [vp, vp, c.c_uint32, vp, vp, c.c_size_t, vp]),
        "qpgpu_ctx_set_hasher": (c.c_int, [vp, c.c_int, u64p, c.c_size_t]),
        "qpgpu_ctx_get_hasher": (c.c_int, [vp]),
        "qpgpu_ctx_challenger_observe": (c.c_int, [vp, vp, u64p, c.c_size_t]),
        "qpgpu_ctx_challenger_get": (c.c_int, [vp, vp, c.POINTER(c.c_uint64)]),
        "qpgpu_pool_create_batched": (c.c_int, [c.c_int, u64p, c.c_size_t, c.c_uint, c.c_uint, c.POINTER(vp)]),
        "qpgpu_circuit_free": (None, [vp]),
        "qpgpu_circuit_constants_sigmas_cap": (c.c_int, [vp, u64p, c.c_size_t]),
        "qpgpu_proof_size": (c.c_size_t, [vp]),
        "qpgpu_circuit_set_blinding_seed": (c.c_int, [vp, c.c_uint64]),
        "qpgpu_circuit_set_witness_check": (c.c_int, [vp, c.c_int]),
        "qpgpu_circuit_gate_rows": (c.c_int, [vp, c.c_uint, vp, c.c_size_t, c.POINTER(c.c_size_t)]),
        "qpgpu_prove": (c.c_int, [vp, u64p, u64p, vp, c.c_size_t, c.POINTER(c.c_size_t)]),
        "qpgpu_prove_dev": (c.c_int, [vp, u64p, u64p, vp, c.c_size_t, c.POINTER(c.c_size_t)]),
        "qpgpu_poseidon_constants": (c.c_size_t, [u64p, u64p, c.c_size_t]),
        "qpgpu_pool_create": (c.c_int, [c.c_int, u64p, c.c_size_t, c.c_uint, c.POINTER(vp)]),
        "qpgpu_pool_destroy": (None, [vp]),
        "qpgpu_pool_proof_size": (c.c_size_t, [vp]),
        "qpgpu_pool_workers": (c.c_uint, [vp]),
        "qpgpu_pool_last_error": (c.c_char_p, [vp]),
        "qpgpu_pool_submit": (c.c_int, [vp, u64p, u64p, vp, c.c_size_t, c.POINTER(c.c_uint64)]),
        "qpgpu_pool_wait": (c.c_int, [vp, c.c_uint64, c.POINTER(c.c_size_t)]),
        "qpgpu_pool_set_witness_check": (c.c_int, [vp, c.c_int]),
        "qpgpu_pool_create_multi": (c.c_int, [vp, c.c_uint, u64p, c.c_size_t, c.c_uint, c.c_uint, c.c_uint, c.POINTER(vp)]),
        "qpgpu_pool_devices": (c.c_uint, [vp]),
        "qpgpu_pool_serialized": (c.c_int, [vp]),
        "qpgpu_pool_set_partial_cells": (c.c_int, [vp, u64p, c.c_size_t]),
        "qpgpu_pool_set_partial_cells_blinded": (c.c_int, [vp, u64p, c.c_size_t, c.c_size_t]),
        "qpgpu_pool_submit_on": (c.c_int, [vp, c.c_uint, u64p, u64p, vp, c.c_size_t, c.POINTER(c.c_uint64)]),
        "qpgpu_pool_submit_host": (c.c_int, [vp, u64p, u64p, vp, c.c_size_t, c.POINTER(c.c_uint64)]),
        "qpgpu_pool_submit_partial": (c.c_int, [vp, u64p, u64p, vp, c.c_size_t, c.POINTER(c.c_uint64)]),
        "qpgpu_set_hasher": (c.c_int, [c.c_int, u64p, c.c_size_t]),
        "qpgpu_get_hasher": (c.c_int, []),
        "qpgpu_witness_info": (c.c_int, [vp, c.POINTER(c.c_uint64), c.POINTER(c.c_uint64), c.POINTER(c.c_uint64)]),
        "qpgpu_witness_free_mask": (c.c_int, [vp, vp, c.c_size_t]),
        "qpgpu_generate_witness_dev": (c.c_int, [vp, u64p, u64p]),
        "qpgpu_generate_witness": (c.c_int, [vp, u64p, u64p]),
        "qpgpu_generate_witness_batch_dev": (c.c_int, [vp, u64p, c.c_uint32, u64p]),
        "qpgpu_generate_witness_partial_dev": (c.c_int, [vp, u64p, u64p, c.c_size_t, u64p, u64p]),
        "qpgpu_generate_witness_partial_batch_dev": (c.c_int, [vp, u64p, c.c_size_t, u64p, u64p, c.c_uint32, u64p, vp]),
        "qpgpu_generate_witness_partial_batch_blinded_dev": (c.c_int, [vp, u64p, c.c_size_t, c.c_size_t, u64p, c.c_char_p, u64p, c.c_uint32, u64p, vp]),
        "qpgpu_witness_public_inputs_dev": (c.c_int, [vp, u64p, c.c_uint32, u64p]),
        "qpgpu_witness_partial_prepare": (c.c_int, [vp, u64p, c.c_size_t, c.c_uint32]),
        "qpgpu_oracle_commit": (c.c_int, [vp, u64p, c.c_uint32, c.c_uint, c.c_uint, c.c_uint, c.c_uint, c.c_uint64, c.c_uint32,
                                          c.POINTER(vp)]),
        "qpgpu_oracle_free": (None, [vp]),
        "qpgpu_oracle_cap": (c.c_int, [vp, u64p, c.c_size_t]),
        "qpgpu_oracle_eval": (c.c_int, [vp, u64p, c.c_uint32, c.c_uint32, u64p]),
        "qpgpu_oracle_read": (c.c_int, [vp, c.c_uint, c.c_uint32, c.c_uint32, u64p]),
        "qpgpu_oracle_device_ptrs": (c.c_int, [vp, c.POINTER(vp), c.POINTER(vp), c.POINTER(vp)]),
        "qpgpu_challenger_init": (None, [vp]),
        "qpgpu_challenger_observe": (None, [vp, u64p, c.c_size_t]),
        "qpgpu_challenger_get": (c.c_uint64, [vp]),
        "qpgpu_fri_proof_size": (c.c_size_t, [c.POINTER(vp), c.c_uint32, vp]),
        "qpgpu_fri_prove": (c.c_int, [vp, c.POINTER(vp), c.c_uint32, vp, c.c_uint32, vp, vp, vp, c.c_size_t,
                                      c.POINTER(c.c_size_t)]),
        "qpgpu_synth_pack_words": (c.c_size_t, [c.c_uint, c.c_uint, c.c_uint]),
        "qpgpu_synth_pack_words_ex": (c.c_size_t, [c.c_uint, c.c_uint, c.c_uint, c.c_uint]),
        "qpgpu_synth_p2_sites": (c.c_size_t, [c.c_uint, c.c_uint, c.c_uint, u64p, c.c_size_t]),
        "qpgpu_synth_circuit_ex": (c.c_int, [c.c_uint, c.c_uint, c.c_uint, c.c_uint, c.c_uint64, c.c_uint, u64p, c.c_size_t,
                                             c.POINTER(c.c_size_t), u64p, u64p]),
        "qpgpu_synth_circuit": (c.c_int, [c.c_uint, c.c_uint, c.c_uint, c.c_uint, c.c_uint64, u64p, c.c_size_t,
                                          c.POINTER(c.c_size_t), u64p, u64p]),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def exported_symbols():
    """Names include/qpgpu.h declares; tests check each resolves in the built library."""
    import glob
    import re
    names = set()
    for hdr in glob.glob(os.path.join(os.path.dirname(_HERE), "include", "*.h")):
        names |= set(re.findall(r"\b(qpgpu_[a-z0-9_]+)\s*\(", open(hdr).read()))
    return sorted(names)


def set_hasher_poseidon():
    """Process default (what NEW contexts and the context-free helpers use): plonky2's Poseidon."""
    load_library().qpgpu_set_hasher(0, None, 0)


def _p2_block(rc_ext, rc_int, diag_m1, m4):
    return np.concatenate([np.asarray(rc_ext, dtype=np.uint64).ravel(), np.asarray(rc_int, dtype=np.uint64).ravel(),
                           np.asarray(diag_m1, dtype=np.uint64).ravel(), np.asarray(m4, dtype=np.uint64).ravel()])


def set_hasher_poseidon2(rc_ext, rc_int, diag_m1, m4):
    """Process default: Poseidon2 with the given parameters (external round constants [8][12], internal [22], diagonal
    [12], 4x4 block). Contexts that already exist keep their hasher; QpGpu(hasher=...) sets one context's."""
    flat = _p2_block(rc_ext, rc_int, diag_m1, m4)
    rc = load_library().qpgpu_set_hasher(1, flat.ctypes.data, flat.size)
    if rc != 0:
        raise QpGpuError(rc, "qpgpu_set_hasher: bad Poseidon2 parameter block")


def poseidon2_qp_params():
    """qp-poseidon-core 3.1.0's Poseidon2 parameter set as the library carries it (pinned by the reference's seven known-answer
    vectors): (rc_ext[8,12], rc_int[22], diag_m1[12], m4[4,4]) — the arguments of set_hasher_poseidon2 / QpGpu(hasher=...)."""
    flat = np.empty(146, dtype=np.uint64)
    n = load_library().qpgpu_poseidon2_qp_params(flat.ctypes.data, flat.size)
    assert n == 146
    return flat[:96].reshape(8, 12).copy(), flat[96:118].copy(), flat[118:130].copy(), flat[130:146].reshape(4, 4).copy()


def poseidon_constants():
    """(round_constants[360], fast_partial[flat]) as derived by the library at start-up. Host only."""
    lib = load_library()
    n = lib.qpgpu_poseidon_constants(None, None, 0)
    rc = np.empty(360, dtype=np.uint64); fp = np.empty(n, dtype=np.uint64)
    lib.qpgpu_poseidon_constants(rc.ctypes.data, fp.ctypes.data, n)
    return rc, fp


def synth_flags(poseidon=False, base_sum=False, ext_arith=False, recursion=False, hints=False, poseidon2=False, p2_alt_layout=False):
    return ((1 if poseidon else 0) | (2 if base_sum else 0) | (4 if ext_arith else 0) | (8 if recursion else 0) | (16 if hints else 0) |
            (64 if poseidon2 else 0) | (128 if p2_alt_layout else 0))


def synth_p2_sites(degree_bits, num_public_inputs=21, **kw):
    """Hash sites of a poseidon2=True synthetic circuit: list of (preimage length, gate rows, first slot); see include/qpgpu.h."""
    out = np.zeros(3 * 64, dtype=np.uint64)
    n = load_library().qpgpu_synth_p2_sites(degree_bits, num_public_inputs, synth_flags(**kw), out.ctypes.data, out.size)
    return [tuple(int(x) for x in out[3 * i:3 * i + 3]) for i in range(n)]


def synth_circuit(degree_bits, num_wires=135, num_routed=80, num_public_inputs=21, seed=1, poseidon=False, base_sum=False,
                  ext_arith=False, recursion=False, hints=False, poseidon2=False, p2_alt_layout=False):
    """Synthetic satisfied circuit: returns (pack_words, wires[num_wires, n], public_inputs). Host only.
    poseidon=True adds PoseidonGate rows (needs 135 wires), base_sum=True BaseSumGate<2> rows, ext_arith=True
    ArithmeticExtensionGate and MulExtensionGate rows, recursion=True Reducing / ReducingExtension / RandomAccess /
    Exponentiation / PoseidonMds rows, hints=True free-standing witness generators (a hint trailer in the pack)."""
    lib = load_library()
    flags = synth_flags(poseidon, base_sum, ext_arith, recursion, hints, poseidon2, p2_alt_layout)
    # with Poseidon rows the pack also carries the public-input cell trailer (2 + num_public_inputs words, circuit.hpp)
    words = lib.qpgpu_synth_pack_words_ex(degree_bits, num_wires, num_routed, flags) + 2 + num_public_inputs
    pack = np.empty(words, dtype=np.uint64)
    wires = np.empty((num_wires, 1 << degree_bits), dtype=np.uint64)
    pis = np.empty(num_public_inputs, dtype=np.uint64)
    got = ctypes.c_size_t()
    rc = lib.qpgpu_synth_circuit_ex(degree_bits, num_wires, num_routed, num_public_inputs, seed, flags, pack.ctypes.data,
                                    words, ctypes.byref(got), wires.ctypes.data, pis.ctypes.data)
    if rc != 0 or got.value > words:
        raise QpGpuError(rc, f"synth_circuit failed (words {got.value} vs {words})")
    return pack[:got.value].copy() if got.value != words else pack, wires, pis


def pack_header(pack_words):
    """The fixed header of a circuit pack (csrc/circuit.hpp) as a dict. Host only."""
    names = ["degree_bits", "num_wires", "num_routed_wires", "num_constants", "num_selectors", "num_challenges",
             "quotient_degree_factor", "num_partial_products", "num_public_inputs", "rate_bits", "cap_height",
             "proof_of_work_bits", "num_query_rounds", "zero_knowledge", "num_gate_constraints", "num_gates", "num_arity_rounds"]
    return {k: int(v) for k, v in zip(names, pack_words[1:18])}


def pack_public_input_cells(pack_words):
    """Cells (row * num_wires + wire) of the public inputs from the pack's "PUBI1" trailer, or None."""
    h = pack_header(pack_words)
    pos = 18 + h["num_arity_rounds"] + 8 * h["num_gates"] + h["num_routed_wires"] + 4
    pos += (h["num_selectors"] + h["num_constants"] + h["num_routed_wires"]) << h["degree_bits"]
    while pos + 2 <= len(pack_words):
        magic, cnt = int(pack_words[pos]), int(pack_words[pos + 1])
        if magic == 0x31544E4948:
            pos += 2 + 8 * cnt
        elif magic == 0x3149425550:
            return np.array(pack_words[pos + 2:pos + 2 + cnt], dtype=np.uint64)
        else:
            break
    return None


P2_LAYOUT_FIELDS = ("w_input", "w_output", "w_swap", "w_delta", "w_full0", "w_partial", "w_full1", "first_round_wires", "constraint_order", "end_wire")
P2_NO_SWAP = 0xFFFFFFFF


def pack_trailers(pack_words):
    """(magic, first word index, count) of every optional trailer of a circuit pack, in order."""
    h = pack_header(pack_words)
    pos = 18 + h["num_arity_rounds"] + 8 * h["num_gates"] + h["num_routed_wires"] + 4
    pos += (h["num_selectors"] + h["num_constants"] + h["num_routed_wires"]) << h["degree_bits"]
    out = []
    while pos + 2 <= len(pack_words):
        magic, cnt = int(pack_words[pos]), int(pack_words[pos + 1])
        per = {0x31544E4948: 8, 0x3149425550: 1, 0x314C473250: 1}.get(magic)
        if per is None:
            break
        out.append((magic, pos + 2, cnt))
        pos += 2 + per * cnt
    return out


def pack_p2_layout(pack_words):
    """Wire layout of the Poseidon2 gate from the pack's "P2GL1" trailer as a dict, or None when the pack has none (the
    library then assumes the default layout, csrc/circuit.hpp)."""
    for magic, at, cnt in pack_trailers(pack_words):
        if magic == 0x314C473250 and cnt == len(P2_LAYOUT_FIELDS):
            return {k: int(v) for k, v in zip(P2_LAYOUT_FIELDS, pack_words[at:at + cnt])}
    return None


def p2_site_cells(pack_words, site):
    """Cells of one Poseidon2 hash site of a synthetic circuit (synth_p2_sites): (preimage cells, digest cells), each a list of
    (wire, row), by the placement rule documented at qpgpu_synth_p2_sites in include/qpgpu.h."""
    lay = pack_p2_layout(pack_words) or dict(zip(P2_LAYOUT_FIELDS, (0, 12, 24, 25, 29, 65, 87, 0, 0, 135)))
    ln, blocks, slot = site
    pre = []
    for i in range(ln):
        k, j = divmod(i, 8)
        pre.append((lay["w_input"] + j, 8 * slot + 3) if k == 0 else (4 * j + 2, 8 * (slot + k - 1) + 4))
    last = 8 * (slot + blocks - 1) + 3
    return pre, [(lay["w_output"] + i, last) for i in range(4)]


class DeviceBuffer:
    """Device allocation owned through the C ABI (hipMalloc underneath)."""

    def __init__(self, gpu, nbytes):
        self.gpu, self.nbytes = gpu, nbytes
        p = ctypes.c_void_p()
        gpu._check(gpu.lib.qpgpu_malloc(gpu.ctx, nbytes, ctypes.byref(p)))
        self.ptr = p.value

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        self.gpu._check(self.gpu.lib.qpgpu_memcpy_h2d(self.gpu.ctx, self.ptr, arr.ctypes.data, arr.nbytes))
        return self

    def download(self, count=None, dtype=np.uint64):
        n = self.nbytes // np.dtype(dtype).itemsize if count is None else count
        out = np.empty(n, dtype=dtype)
        self.gpu._check(self.gpu.lib.qpgpu_memcpy_d2h(self.gpu.ctx, out.ctypes.data, self.ptr, out.nbytes))
        return out

    def free(self, scrub=False):
        """scrub=True for buffers that held a witness: zeroed on the device before release."""
        if self.ptr:
            if scrub:
                self.gpu.lib.qpgpu_free_scrubbed(self.gpu.ctx, self.ptr, self.nbytes)
            else:
                self.gpu.lib.qpgpu_free(self.gpu.ctx, self.ptr)
            self.ptr = None


class Circuit:
    """A circuit pack loaded on the GPU (constants/sigmas committed, workspace allocated)."""

    def __init__(self, gpu, pack_words, max_batch=1):
        self.gpu = gpu
        pw = np.ascontiguousarray(pack_words, dtype=np.uint64)
        h = ctypes.c_void_p()
        gpu._check(gpu.lib.qpgpu_circuit_load_batch(gpu.ctx, pw.ctypes.data, pw.size, max_batch, ctypes.byref(h)))
        self.h = h
        self.max_batch = max_batch
        self.num_public_inputs = int(pw[9])

    def scrub(self):
        """Overwrite every witness-derived device region of the workspace now."""
        self.gpu._check(self.gpu.lib.qpgpu_circuit_scrub(self.h))

    def prove_batch_dev(self, d_wires_list, public_inputs_list, outs=None):
        """Lockstep batch: one device witness and one public-input vector per proof; returns the list of proof bytes."""
        nb = len(d_wires_list)
        ptrs = (ctypes.c_void_p * nb)(*[_ptr(w) for w in d_wires_list])
        pis = [np.ascontiguousarray(p, dtype=np.uint64) for p in public_inputs_list]
        pip = (ctypes.c_void_p * nb)(*[p.ctypes.data for p in pis])
        size = self.proof_size()
        if outs is None:
            outs = [np.empty(size, dtype=np.uint8) for _ in range(nb)]
        op = (ctypes.c_void_p * nb)(*[o.ctypes.data for o in outs])
        lens = (ctypes.c_size_t * nb)()
        self.gpu._check(self.gpu.lib.qpgpu_prove_batch_dev(self.h, ptrs, nb, pip, op, size, lens))
        return [outs[i][:lens[i]].tobytes() for i in range(nb)]

    def close(self):
        if self.h:
            self.gpu.lib.qpgpu_circuit_free(self.h)
            self.h = None

    def proof_size(self):
        return self.gpu.lib.qpgpu_proof_size(self.h)

    def set_witness_check(self, on=True):
        """Make prove() return QPGPU_EUNSAT (-4) for a witness that violates a gate or copy constraint."""
        self.gpu._check(self.gpu.lib.qpgpu_circuit_set_witness_check(self.h, 1 if on else 0))

    def set_blinding_seed(self, seed):
        """Zero-knowledge packs: fixes the salts of the next proof (reproducible bytes)."""
        self.gpu._check(self.gpu.lib.qpgpu_circuit_set_blinding_seed(self.h, seed))

    def constants_sigmas_cap(self, cap_height=4):
        out = np.empty((1 << cap_height, 4), dtype=np.uint64)
        self.gpu._check(self.gpu.lib.qpgpu_circuit_constants_sigmas_cap(self.h, out.ctypes.data, out.size))
        return out

    def witness_info(self):
        """(generator instances, dependency levels, caller-supplied cells) of the witness-generation plan."""
        a, b, c3 = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
        self.gpu._check(self.gpu.lib.qpgpu_witness_info(self.h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c3)))
        return a.value, b.value, c3.value

    def gate_rows(self, gate_type):
        """The trace rows of a gate type (QPCP type codes: 4 PoseidonGate, 14 the Poseidon2 gate, ...)."""
        n = ctypes.c_size_t()
        self.gpu._check(self.gpu.lib.qpgpu_circuit_gate_rows(self.h, gate_type, None, 0, ctypes.byref(n)))
        rows = np.empty(n.value, dtype=np.uint32)
        self.gpu._check(self.gpu.lib.qpgpu_circuit_gate_rows(self.h, gate_type, rows.ctypes.data, rows.size, ctypes.byref(n)))
        return rows

    def witness_free_mask(self, num_wires, n):
        """uint8 [num_wires, n]: 1 where the caller supplies the cell (PartialWitness), 0 where generation writes it."""
        m = np.empty((num_wires, n), dtype=np.uint8)
        self.gpu._check(self.gpu.lib.qpgpu_witness_free_mask(self.h, m.ctypes.data, m.size))
        return m

    def generate_witness(self, wires, public_inputs):
        """Fills every generator- or copy-determined cell of a host wire matrix [num_wires, n]; returns the full matrix."""
        w = np.array(wires, dtype=np.uint64, order="C", copy=True)
        p = np.ascontiguousarray(public_inputs, dtype=np.uint64)
        self.gpu._check(self.gpu.lib.qpgpu_generate_witness(self.h, w.ctypes.data, p.ctypes.data))
        return w

    def generate_witness_dev(self, d_wires, public_inputs, batch=1):
        """In place on `batch` device-resident wire matrices laid out back to back; public_inputs [batch, num_pis]."""
        p = np.ascontiguousarray(public_inputs, dtype=np.uint64)
        self.gpu._check(self.gpu.lib.qpgpu_generate_witness_batch_dev(self.h, _ptr(d_wires), batch, p.ctypes.data))

    def generate_witness_partial_dev(self, cells, values, public_inputs, d_wires):
        """plonky2's PartialWitness as (cell = row * num_wires + wire, value) assignments -> full witness in d_wires.
        Raises QpGpuError(-4) when a target is set twice with different values."""
        cl = np.ascontiguousarray(cells, dtype=np.uint64); vl = np.ascontiguousarray(values, dtype=np.uint64)
        assert cl.shape == vl.shape
        p = np.ascontiguousarray(public_inputs, dtype=np.uint64)
        self.gpu._check(self.gpu.lib.qpgpu_generate_witness_partial_dev(self.h, cl.ctypes.data, vl.ctypes.data, cl.size,
                                                                        p.ctypes.data, _ptr(d_wires)))

    def generate_witness_partial_batch_dev(self, cells, values, public_inputs, d_wires):
        """`batch` PartialWitnesses over one cell list: values [batch, count], public_inputs [batch, num_pis], d_wires batch
        matrices back to back. Returns the per-witness status list (0 or -4); never raises for an unsatisfied witness."""
        cl = np.ascontiguousarray(cells, dtype=np.uint64); vl = np.ascontiguousarray(values, dtype=np.uint64).reshape(-1, cl.size)
        p = None if public_inputs is None else np.ascontiguousarray(public_inputs, dtype=np.uint64)     # None: derived (witness_public_inputs_dev reads them)
        batch = vl.shape[0]
        status = (ctypes.c_int * batch)()
        rc = self.gpu.lib.qpgpu_generate_witness_partial_batch_dev(self.h, cl.ctypes.data, cl.size, vl.ctypes.data, None if p is None else p.ctypes.data, batch,
                                                                   _ptr(d_wires), status)
        if rc not in (0, -4):
            self.gpu._check(rc)
        return list(status)

    def generate_witness_partial_batch_blinded_dev(self, cells, values, public_inputs, d_wires, n_blinding, seeds=None):
        """The same with the LAST n_blinding cells drawn on the device (RandomValueGenerator targets of a zero-knowledge circuit):
        values [batch, count - n_blinding]; seeds: batch x 32 bytes for reproducible tests, None = OS entropy per witness."""
        cl = np.ascontiguousarray(cells, dtype=np.uint64)
        vl = np.ascontiguousarray(values, dtype=np.uint64).reshape(-1, cl.size - n_blinding)
        p = None if public_inputs is None else np.ascontiguousarray(public_inputs, dtype=np.uint64)
        batch = vl.shape[0]
        sd = None
        if seeds is not None:
            sd = bytes(seeds)
            assert len(sd) == 32 * batch
        status = (ctypes.c_int * batch)()
        rc = self.gpu.lib.qpgpu_generate_witness_partial_batch_blinded_dev(self.h, cl.ctypes.data, cl.size, n_blinding, vl.ctypes.data, sd,
                                                                           None if p is None else p.ctypes.data, batch, _ptr(d_wires), status)
        if rc not in (0, -4):
            self.gpu._check(rc)
        return list(status)

    def witness_public_inputs_dev(self, d_wires, batch=1):
        """The public inputs of `batch` resident witnesses, [batch, num_public_inputs] (what plonky2's prove() reads out of the
        partition witness)."""
        out = np.empty((batch, self.num_public_inputs), dtype=np.uint64)
        self.gpu._check(self.gpu.lib.qpgpu_witness_public_inputs_dev(self.h, _ptr(d_wires), batch, out.ctypes.data))
        return out

    def witness_partial_prepare(self, cells, max_batch):
        cl = np.ascontiguousarray(cells, dtype=np.uint64)
        self.gpu._check(self.gpu.lib.qpgpu_witness_partial_prepare(self.h, cl.ctypes.data, cl.size, max_batch))

    def prove(self, wires, public_inputs):
        """wires: host array [num_wires, n]; returns proof bytes."""
        w = np.ascontiguousarray(wires, dtype=np.uint64)
        p = np.ascontiguousarray(public_inputs, dtype=np.uint64)
        out = np.empty(self.proof_size(), dtype=np.uint8)
        ln = ctypes.c_size_t()
        self.gpu._check(self.gpu.lib.qpgpu_prove(self.h, w.ctypes.data, p.ctypes.data, out.ctypes.data, out.size, ctypes.byref(ln)))
        return out[:ln.value].tobytes()

    def prove_dev(self, d_wires, public_inputs, out=None):
        """wires resident in HBM (DeviceBuffer / torch tensor / raw pointer)."""
        p = np.ascontiguousarray(public_inputs, dtype=np.uint64)
        if out is None:
            out = np.empty(self.proof_size(), dtype=np.uint8)
        ln = ctypes.c_size_t()
        self.gpu._check(self.gpu.lib.qpgpu_prove_dev(self.h, _ptr(d_wires), p.ctypes.data, out.ctypes.data, out.size, ctypes.byref(ln)))
        return out[:ln.value].tobytes()


ORACLE_VALUES, ORACLE_COEFFS, ORACLE_BLINDING, ORACLE_DEVICE_INPUT = 0, 1, 2, 4


class Verifier:
    """Host-side verifier of one circuit (include/qpgpu_verify.h): plonky2's VerifierCircuitData::verify over the circuit pack.
    circuit: a loaded Circuit whose constants/sigmas cap becomes the verifier data (without one the cap is rebuilt from the
    pack on the host). hasher: 0 Poseidon, 1 Poseidon2 (params None = the built-in qp-poseidon-core set)."""

    def __init__(self, pack_words, circuit=None, cap=None, hasher=0, params=None):
        lib = load_library()
        self.lib = lib
        pw = np.ascontiguousarray(pack_words, dtype=np.uint64)
        if cap is None and circuit is not None:
            cap = circuit.constants_sigmas_cap(int(pw[11]))
        cw = None if cap is None else np.ascontiguousarray(cap, dtype=np.uint64).reshape(-1)
        pr = None if params is None else np.ascontiguousarray(params, dtype=np.uint64)
        h = ctypes.c_void_p(); err = ctypes.create_string_buffer(200)
        rc = lib.qpgpu_verifier_create(pw.ctypes.data, pw.size, None if cw is None else cw.ctypes.data, 0 if cw is None else cw.size, hasher,
                                       None if pr is None else pr.ctypes.data, 0 if pr is None else pr.size, ctypes.byref(h), err)
        if rc:
            raise QpGpuError(rc, err.value.decode())
        self.h = h
        self.reason = ""

    def close(self):
        if self.h:
            self.lib.qpgpu_verifier_free(self.h); self.h = None

    def proof_size(self):
        return int(self.lib.qpgpu_verifier_proof_size(self.h))

    def verify(self, proof):
        """True when accepted; otherwise False and .reason says which check failed."""
        err = ctypes.create_string_buffer(200)
        b = bytes(proof)
        rc = self.lib.qpgpu_verifier_verify(self.h, b, len(b), err)
        self.reason = err.value.decode()
        return rc == 0

    def verify_many(self, proofs, threads=0):
        """[accepted?] per proof, verified on up to `threads` host threads (0 = all cores); .reason names the first rejection."""
        n = len(proofs)
        if n == 0:
            return []
        bufs = [bytes(p) for p in proofs]
        ptrs = (ctypes.c_char_p * n)(*bufs)
        lens = (ctypes.c_size_t * n)(*[len(b) for b in bufs])
        res = (ctypes.c_int * n)()
        err = ctypes.create_string_buffer(200)
        self.lib.qpgpu_verifier_verify_many(self.h, ptrs, lens, n, threads, res, err)
        self.reason = err.value.decode()
        return [r == 0 for r in res]


class ChallengerState(ctypes.Structure):
    """qpgpu_challenger: plonky2's Challenger as plain data (host only)."""
    _fields_ = [("sponge_state", ctypes.c_uint64 * 12), ("input_buffer", ctypes.c_uint64 * 8),
                ("output_buffer", ctypes.c_uint64 * 8), ("input_len", ctypes.c_uint32), ("output_len", ctypes.c_uint32)]


class Challenger:
    """gpu=None: the process-default hasher (first ABI); gpu=QpGpu: that context's hasher."""

    def __init__(self, gpu=None):
        self.lib = load_library()
        self.gpu = gpu
        self.state = ChallengerState()
        self.lib.qpgpu_challenger_init(ctypes.byref(self.state))

    def observe(self, xs):
        x = np.ascontiguousarray(xs, dtype=np.uint64).ravel()
        if self.gpu is None:
            self.lib.qpgpu_challenger_observe(ctypes.byref(self.state), x.ctypes.data, x.size)
        else:
            rc = self.lib.qpgpu_ctx_challenger_observe(self.gpu.ctx, ctypes.byref(self.state), x.ctypes.data, x.size)
            if rc != 0:
                raise QpGpuError(rc, "invalid challenger state")

    def get(self):
        if self.gpu is None:
            return int(self.lib.qpgpu_challenger_get(ctypes.byref(self.state)))
        v = ctypes.c_uint64()
        rc = self.lib.qpgpu_ctx_challenger_get(self.gpu.ctx, ctypes.byref(self.state), ctypes.byref(v))
        if rc != 0:
            raise QpGpuError(rc, "invalid challenger state")
        return int(v.value)

    def get_n(self, n):
        return [self.get() for _ in range(n)]


class _FriRange(ctypes.Structure):
    _fields_ = [("oracle", ctypes.c_uint32), ("first", ctypes.c_uint32), ("count", ctypes.c_uint32)]


class _FriBatch(ctypes.Structure):
    _fields_ = [("point", ctypes.c_uint64 * 2), ("num_ranges", ctypes.c_uint32), ("ranges", _FriRange * 8)]


class _FriParams(ctypes.Structure):
    _fields_ = [("rate_bits", ctypes.c_uint32), ("cap_height", ctypes.c_uint32), ("proof_of_work_bits", ctypes.c_uint32),
                ("num_query_rounds", ctypes.c_uint32), ("num_reduction_rounds", ctypes.c_uint32),
                ("reduction_arity_bits", ctypes.c_uint32 * 16)]


class PolyOracle:
    """A committed polynomial batch (plonky2 PolynomialBatch) resident on the GPU."""

    def __init__(self, gpu, polys, rate_bits=3, cap_height=4, coeffs=False, blinding=False, blinding_seed=0,
                 blinding_stream=0):
        self.gpu = gpu
        if isinstance(polys, np.ndarray):
            a = np.ascontiguousarray(polys, dtype=np.uint64)
            num_polys, n = a.shape
            ptr, flags = a.ctypes.data, 0
        else:                       # (device pointer, num_polys, n)
            dev, num_polys, n = polys
            ptr, flags = _ptr(dev), ORACLE_DEVICE_INPUT
        self.num_polys, self.n = num_polys, n
        self.degree_bits, self.rate_bits, self.cap_height = n.bit_length() - 1, rate_bits, cap_height
        assert 1 << self.degree_bits == n
        flags |= (ORACLE_COEFFS if coeffs else ORACLE_VALUES) | (ORACLE_BLINDING if blinding else 0)
        h = ctypes.c_void_p()
        gpu._check(gpu.lib.qpgpu_oracle_commit(gpu.ctx, ptr, num_polys, self.degree_bits, rate_bits, cap_height, flags,
                                               blinding_seed, blinding_stream, ctypes.byref(h)))
        self.h = h

    def close(self):
        if self.h:
            self.gpu.lib.qpgpu_oracle_free(self.h)
            self.h = None

    def cap(self):
        out = np.empty((1 << self.cap_height, 4), dtype=np.uint64)
        self.gpu._check(self.gpu.lib.qpgpu_oracle_cap(self.h, out.ctypes.data, out.size))
        return out

    def eval(self, point, first=0, count=None):
        """Evaluations at an extension point (a, b): array [count, 2]."""
        count = self.num_polys - first if count is None else count
        pt = np.array(point, dtype=np.uint64)
        out = np.empty((count, 2), dtype=np.uint64)
        self.gpu._check(self.gpu.lib.qpgpu_oracle_eval(self.h, pt.ctypes.data, first, count, out.ctypes.data))
        return out

    def read(self, lde=False, first=0, count=None):
        count = self.num_polys - first if count is None else count
        out = np.empty((count, self.n << (self.rate_bits if lde else 0)), dtype=np.uint64)
        self.gpu._check(self.gpu.lib.qpgpu_oracle_read(self.h, 1 if lde else 0, first, count, out.ctypes.data))
        return out


def fri_prove(gpu, oracles, batches, challenger, reduction_arity_bits, rate_bits=3, cap_height=4, proof_of_work_bits=16,
              num_query_rounds=28):
    """PolynomialBatch::prove_openings. batches: [(point(a, b), [(oracle, first, count), ...]), ...]; challenger is
    advanced in place. Returns the FriProof bytes."""
    hs = (ctypes.c_void_p * len(oracles))(*[o.h for o in oracles])
    bs = (_FriBatch * len(batches))()
    for b, (point, ranges) in zip(bs, batches):
        b.point[0], b.point[1] = int(point[0]), int(point[1])
        b.num_ranges = len(ranges)
        for r, (o, f, cnt) in zip(b.ranges, ranges):
            r.oracle, r.first, r.count = o, f, cnt
    prm = _FriParams(rate_bits, cap_height, proof_of_work_bits, num_query_rounds, len(reduction_arity_bits))
    for i, a in enumerate(reduction_arity_bits):
        prm.reduction_arity_bits[i] = a
    size = gpu.lib.qpgpu_fri_proof_size(hs, len(oracles), ctypes.byref(prm))
    if size == 0:
        raise QpGpuError(-1, "fri_prove: bad oracles or FRI parameters")
    out = np.empty(size, dtype=np.uint8)
    ln = ctypes.c_size_t()
    gpu._check(gpu.lib.qpgpu_fri_prove(gpu.ctx, hs, len(oracles), ctypes.byref(bs), len(batches), ctypes.byref(prm),
                                       ctypes.byref(challenger.state), out.ctypes.data, out.size, ctypes.byref(ln)))
    return out[:ln.value].tobytes()


class ProvingPool:
    """Several proofs of one circuit in flight on one GPU (qpgpu_pool_*): worker threads, streams and circuit copies live
    inside the library."""

    def __init__(self, pack_words, workers=4, device=0, max_batch=1, devices=None, host_witness=False):
        """devices: a list of HIP devices (one entry may repeat) -> qpgpu_pool_create_multi with `workers` workers on each."""
        self.lib = load_library()
        pw = np.ascontiguousarray(pack_words, dtype=np.uint64)
        h = ctypes.c_void_p()
        devs = [device] if devices is None else list(devices)
        arr = (ctypes.c_int * len(devs))(*devs)
        rc = self.lib.qpgpu_pool_create_multi(arr, len(devs), pw.ctypes.data, pw.size, workers, max_batch, 1 if host_witness else 0, ctypes.byref(h))
        if rc != 0:
            raise QpGpuError(rc, "qpgpu_pool_create failed")
        self.h = h
        self._keep = {}
        # the workers write into the callers' output buffers: the pool must have drained before the interpreter frees them,
        # also when an exception unwinds past the owner or the process exits without close()
        import atexit
        atexit.register(self.close)

    def close(self):
        if self.h:
            self.lib.qpgpu_pool_destroy(self.h)      # drains the queue first
            self.h = None
            self._keep.clear()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def proof_size(self):
        return self.lib.qpgpu_pool_proof_size(self.h)

    def set_witness_check(self, on=True):
        rc = self.lib.qpgpu_pool_set_witness_check(self.h, 1 if on else 0)
        if rc != 0:
            raise QpGpuError(rc, self.lib.qpgpu_pool_last_error(self.h).decode())

    def submit(self, d_wires, public_inputs, out=None):
        """Queue one proof of a device-resident witness; returns a ticket for wait()."""
        p = np.ascontiguousarray(public_inputs, dtype=np.uint64)
        if out is None:
            out = np.empty(self.proof_size(), dtype=np.uint8)
        t = ctypes.c_uint64()
        rc = self.lib.qpgpu_pool_submit(self.h, _ptr(d_wires), p.ctypes.data, out.ctypes.data, out.size, ctypes.byref(t))
        if rc != 0:
            raise QpGpuError(rc, self.lib.qpgpu_pool_last_error(self.h).decode())
        self._keep[t.value] = (p, out, d_wires)
        return t.value

    def _submitted(self, rc, t, keep):
        if rc != 0:
            raise QpGpuError(rc, self.lib.qpgpu_pool_last_error(self.h).decode())
        self._keep[t.value] = keep
        return t.value

    def submit_on(self, device_index, d_wires, public_inputs, out=None):
        """A device-resident witness on the pool's device number `device_index` (only that device's workers take it)."""
        p = np.ascontiguousarray(public_inputs, dtype=np.uint64)
        out = np.empty(self.proof_size(), dtype=np.uint8) if out is None else out
        t = ctypes.c_uint64()
        rc = self.lib.qpgpu_pool_submit_on(self.h, device_index, _ptr(d_wires), p.ctypes.data, out.ctypes.data, out.size, ctypes.byref(t))
        return self._submitted(rc, t, (p, out, d_wires))

    def submit_host(self, wires, public_inputs, out=None):
        """A full wire matrix in host memory (any worker of any device uploads and proves it)."""
        w = np.ascontiguousarray(wires, dtype=np.uint64); p = np.ascontiguousarray(public_inputs, dtype=np.uint64)
        out = np.empty(self.proof_size(), dtype=np.uint8) if out is None else out
        t = ctypes.c_uint64()
        rc = self.lib.qpgpu_pool_submit_host(self.h, w.ctypes.data, p.ctypes.data, out.ctypes.data, out.size, ctypes.byref(t))
        return self._submitted(rc, t, (p, out, w))

    def set_partial_cells(self, cells, n_blinding=0):
        """n_blinding: the last n_blinding cells are the blinding wires of a zero-knowledge circuit, drawn on the device per proof."""
        cl = np.ascontiguousarray(cells, dtype=np.uint64)
        rc = self.lib.qpgpu_pool_set_partial_cells_blinded(self.h, cl.ctypes.data, cl.size, n_blinding)
        if rc != 0:
            raise QpGpuError(rc, self.lib.qpgpu_pool_last_error(self.h).decode())
        self._ncells = cl.size - n_blinding

    def submit_partial(self, values, public_inputs, out=None):
        """A PartialWitness over the pool's cell list: stage s1 on the device, then the proof. public_inputs=None: the proof carries
        the public inputs its witness holds (circuits that compute them)."""
        v = np.ascontiguousarray(values, dtype=np.uint64)
        p = None if public_inputs is None else np.ascontiguousarray(public_inputs, dtype=np.uint64)
        assert v.size == self._ncells
        out = np.empty(self.proof_size(), dtype=np.uint8) if out is None else out
        t = ctypes.c_uint64()
        rc = self.lib.qpgpu_pool_submit_partial(self.h, v.ctypes.data, None if p is None else p.ctypes.data, out.ctypes.data, out.size, ctypes.byref(t))
        return self._submitted(rc, t, (p, out, None))

    def wait(self, ticket, copy=True):
        """The proof of a ticket (bytes); copy=False: only its length — the bytes are in the `out` buffer given to submit()."""
        ln = ctypes.c_size_t()
        rc = self.lib.qpgpu_pool_wait(self.h, ticket, ctypes.byref(ln))
        p, out, _ = self._keep.pop(ticket, (None, None, None))
        if rc != 0:
            raise QpGpuError(rc, self.lib.qpgpu_pool_last_error(self.h).decode())
        return out[:ln.value].tobytes() if copy else ln.value


class _Stage3:
    """mixin: stage s3 wrappers"""

    def poseidon_permute(self, states):
        st = np.ascontiguousarray(states, dtype=np.uint64).reshape(-1, 12)
        d = self.to_device(st)
        self._check(self.lib.qpgpu_poseidon_permute_dev(self.ctx, d.ptr, st.shape[0]))
        out = d.download().reshape(-1, 12)
        d.free()
        return out

    def poseidon2_hash_pad10(self, preimages, params=None):
        """The fork's application hash on the device (qpgpu_poseidon2_hash_pad10_dev): preimages [count, len] -> [count, 4].
        params None = qp-poseidon-core's pinned set; otherwise the 146-word block."""
        pre = np.ascontiguousarray(preimages, dtype=np.uint64)
        pre = pre.reshape(1, -1) if pre.ndim == 1 else pre
        count, ln = pre.shape
        d_in = self.to_device(pre) if pre.size else self.alloc(8)
        d_out = self.alloc(max(count, 1) * 32)
        try:
            if params is None:
                self._check(self.lib.qpgpu_poseidon2_hash_pad10_dev(self.ctx, None, 0, d_in.ptr, ln, count, d_out.ptr))
            else:
                blk = np.ascontiguousarray(params, dtype=np.uint64)
                self._check(self.lib.qpgpu_poseidon2_hash_pad10_dev(self.ctx, blk.ctypes.data, blk.size, d_in.ptr, ln, count, d_out.ptr))
            self.sync()
            return d_out.download()[:count * 4].reshape(count, 4).copy()
        finally:
            d_in.free(); d_out.free()

    def merkle_digest_count(self, log_leaves, cap_height):
        return self.lib.qpgpu_merkle_digest_count(log_leaves, cap_height)

    def merkle_build_dev(self, d_cols, col_stride, n_cols, log_leaves, cap_height, d_digests):
        cap = np.empty((1 << cap_height, 4), dtype=np.uint64)
        self._check(self.lib.qpgpu_merkle_build_dev(self.ctx, _ptr(d_cols), col_stride, n_cols, log_leaves,
                                                    cap_height, _ptr(d_digests), cap.ctypes.data))
        return cap

    def merkle_build_rows_dev(self, d_rows, width, log_leaves, cap_height, d_digests):
        cap = np.empty((1 << cap_height, 4), dtype=np.uint64)
        self._check(self.lib.qpgpu_merkle_build_rows_dev(self.ctx, _ptr(d_rows), width, log_leaves, cap_height,
                                                         _ptr(d_digests), cap.ctypes.data))
        return cap


class QpGpu(_Stage3):
    """One context = one GPU + one HIP stream."""

    def __init__(self, device=0, stream=None, hasher=None):
        """hasher: None = the process default; "poseidon"; or the Poseidon2 block (rc_ext, rc_int, diag_m1, m4)."""
        self.lib = load_library()
        ctx = ctypes.c_void_p()
        rc = self.lib.qpgpu_ctx_create(device, ctypes.byref(ctx))
        if rc != 0:
            raise QpGpuError(rc, "qpgpu_ctx_create failed (no gfx950 device visible?)")
        self.ctx = ctx
        if hasher is not None:
            if isinstance(hasher, str):
                self._check(self.lib.qpgpu_ctx_set_hasher(self.ctx, 0, None, 0))
            else:
                flat = _p2_block(*hasher)
                self._check(self.lib.qpgpu_ctx_set_hasher(self.ctx, 1, flat.ctypes.data, flat.size))
        if stream is not None:
            self._check(self.lib.qpgpu_ctx_set_stream(self.ctx, ctypes.c_void_p(stream)))

    def close(self):
        if self.ctx:
            self.lib.qpgpu_ctx_destroy(self.ctx)
            self.ctx = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc):
        if rc != 0:
            raise QpGpuError(rc, self.lib.qpgpu_last_error(self.ctx).decode())

    def last_error(self):
        return self.lib.qpgpu_last_error(self.ctx).decode()

    def sync(self):
        self._check(self.lib.qpgpu_sync(self.ctx))

    def pci_bus_id(self):
        """"dddd:bb:dd.f" of the context's device (the key of its sysfs directory)."""
        buf = ctypes.create_string_buffer(16)
        self._check(self.lib.qpgpu_ctx_pci_bus_id(self.ctx, buf, 16))
        return buf.value.decode()

    def profile(self, on=True):
        self._check(self.lib.qpgpu_profile_enable(self.ctx, 1 if on else 0))

    def profile_read(self, kernel):
        ms, n = ctypes.c_double(), ctypes.c_uint64()
        self._check(self.lib.qpgpu_profile_read(self.ctx, kernel.encode(), ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value

    def alloc(self, nbytes):
        return DeviceBuffer(self, nbytes)

    def to_device(self, arr):
        arr = np.ascontiguousarray(arr)
        return DeviceBuffer(self, arr.nbytes).upload(arr)

    # ---- stage s2 ----
    def ntt_host(self, data, log_n, inverse=False, coset_shift=0, bitrev=False):
        """plonky2 fft / ifft / coset_fft on a host array [batch, 2^log_n]; returns a new array."""
        a = np.array(data, dtype=np.uint64, copy=True).reshape(-1, 1 << log_n)
        flags = (NTT_INVERSE if inverse else 0) | (NTT_OUT_BITREV if bitrev else 0)
        self._check(self.lib.qpgpu_ntt_batch(self.ctx, a.ctypes.data, log_n, a.shape[0], flags, coset_shift))
        return a

    def ntt_dev(self, d_in, d_out, log_n, batch, inverse=False, coset_shift=0, bitrev=False):
        flags = (NTT_INVERSE if inverse else 0) | (NTT_OUT_BITREV if bitrev else 0)
        self._check(self.lib.qpgpu_ntt_batch_dev(self.ctx, _ptr(d_in), _ptr(d_out), log_n, batch, flags, coset_shift))

    def lde_dev(self, d_coeffs, d_out, log_n, rate_bits, batch, coset_shift=MULT_GEN, bitrev=False):
        flags = NTT_OUT_BITREV if bitrev else 0
        self._check(self.lib.qpgpu_lde_batch_dev(self.ctx, _ptr(d_coeffs), _ptr(d_out), log_n, rate_bits, batch, flags, coset_shift))


def _ptr(x):
    if isinstance(x, DeviceBuffer):
        return x.ptr
    if hasattr(x, "data_ptr"):  # torch tensor living in HBM
        return x.data_ptr()
    return int(x)
