"""ctypes view of oracle/liboracle.so — the CPU restatement used as the parity checker.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
P = 0xFFFFFFFF00000001


def _vp(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def usable_cpus(cap=16):
    """Threads worth starting: the affinity mask, the cgroup CPU quota when there is one, and a cap. OpenMP's default
    (every CPU of the host) oversubscribes a container with a CPU quota and makes the spin-waiting runtime crawl."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return max(1, min(n, cap))


class Oracle:
    def __init__(self):
        so = os.path.join(ORACLE_DIR, "liboracle.so")
        srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith((".c", ".h", ".inc"))]
        if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
            subprocess.check_call(["make", "-C", ORACLE_DIR])
        self.lib = L = ctypes.CDLL(so)
        u64 = ctypes.c_uint64
        for name in ("orc_gl_mul", "orc_gl_add", "orc_gl_sub", "orc_gl_pow"):
            getattr(L, name).restype = u64
            getattr(L, name).argtypes = [u64, u64]
        L.orc_gl_inv.restype = u64; L.orc_gl_inv.argtypes = [u64]
        L.orc_gl_root.restype = u64; L.orc_gl_root.argtypes = [ctypes.c_uint]
        L.orc_fft.argtypes = [ctypes.c_void_p, ctypes.c_uint]
        L.orc_ifft.argtypes = [ctypes.c_void_p, ctypes.c_uint]
        L.orc_coset_fft.argtypes = [ctypes.c_void_p, ctypes.c_uint, u64]
        L.orc_coset_ifft.argtypes = [ctypes.c_void_p, ctypes.c_uint, u64]
        L.orc_dft_naive.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint]
        L.orc_fft_batch.argtypes = [ctypes.c_void_p, ctypes.c_uint, ctypes.c_size_t, ctypes.c_int]
        L.orc_lde_batch.argtypes = [ctypes.c_void_p, ctypes.c_uint, ctypes.c_uint, ctypes.c_size_t, u64,
                                    ctypes.c_void_p, ctypes.c_void_p]
        L.orc_poseidon_permute.argtypes = [ctypes.c_void_p]
        L.orc_poseidon_round_constants.argtypes = [ctypes.c_void_p]
        L.orc_hash_n_to_m_no_pad.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
        L.orc_hash_or_noop.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
        L.orc_two_to_one.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        L.orc_merkle_build.restype = ctypes.c_size_t
        L.orc_merkle_build.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_uint,
                                       ctypes.c_void_p, ctypes.c_void_p]
        L.orc_merkle_path.restype = ctypes.c_size_t
        L.orc_merkle_path.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint, ctypes.c_size_t, ctypes.c_void_p]
        L.orc_challenger_size.restype = ctypes.c_size_t
        L.orc_challenger_init.argtypes = [ctypes.c_void_p]
        L.orc_challenger_observe.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
        L.orc_challenger_get.restype = u64; L.orc_challenger_get.argtypes = [ctypes.c_void_p]
        L.orc_challenger_pow_response.restype = u64
        L.orc_challenger_pow_response.argtypes = [ctypes.c_void_p, u64]
        L.orc_circuit_load.restype = ctypes.c_void_p
        L.orc_circuit_load.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
        L.orc_circuit_free.argtypes = [ctypes.c_void_p]
        L.orc_proof_size.restype = ctypes.c_size_t; L.orc_proof_size.argtypes = [ctypes.c_void_p]
        L.orc_prove.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                                ctypes.POINTER(ctypes.c_size_t)]
        L.orc_prove_seeded.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p,
                                       ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
        L.orc_verify.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
        L.orc_trace_len.restype = ctypes.c_size_t; L.orc_trace_len.argtypes = [ctypes.c_char_p]
        L.orc_trace_get.argtypes = [ctypes.c_char_p, ctypes.c_void_p]
        L.orc_poseidon_fast_partial.restype = ctypes.c_size_t; L.orc_poseidon_fast_partial.argtypes = [ctypes.c_void_p]
        L.orc_p2_params_size.restype = ctypes.c_size_t
        L.orc_p2_qp_params.argtypes = [ctypes.c_void_p]
        L.orc_select_hasher_p2.argtypes = [ctypes.c_void_p]
        L.orc_p2_permute.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.orc_p2_hash_pad10.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
        L.orc_bytes_to_u64s.restype = ctypes.c_size_t
        L.orc_bytes_to_u64s.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p]
        L.orc_bytes_to_digest.argtypes = [ctypes.c_char_p, ctypes.c_void_p]
        L.orc_digest_to_bytes.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
        L.orc_generate_witness.restype = ctypes.c_int
        L.orc_generate_witness.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                                           ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)]
        L.orc_witness_plan_create.restype = ctypes.c_void_p
        L.orc_witness_plan_create.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_int)]
        L.orc_witness_plan_free.argtypes = [ctypes.c_void_p]
        L.orc_witness_plan_circuit.restype = ctypes.c_void_p; L.orc_witness_plan_circuit.argtypes = [ctypes.c_void_p]
        L.orc_witness_generate.restype = ctypes.c_int
        L.orc_witness_generate.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p,
                                           ctypes.POINTER(ctypes.c_uint64)]
        L.orc_commit_prove_many.restype = ctypes.c_int
        L.orc_commit_prove_many.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p,
                                            ctypes.c_void_p, ctypes.c_size_t]
        L.orc_set_threads(usable_cpus())

    def select_poseidon2(self, rc_ext, rc_int, diag_m1, m4):
        """Proof-system permutation := Poseidon2 with these parameters (kept alive here); select_poseidon() undoes it."""
        buf = np.zeros(self.lib.orc_p2_params_size() // 8, dtype=np.uint64)
        flat = np.concatenate([np.asarray(rc_ext, dtype=np.uint64).ravel(), np.asarray(rc_int, dtype=np.uint64).ravel(),
                               np.asarray(diag_m1, dtype=np.uint64).ravel(), np.asarray(m4, dtype=np.uint64).ravel()])
        buf[:flat.size] = flat
        self._p2 = buf
        self.lib.orc_select_hasher_p2(_vp(buf))
    def select_poseidon(self):
        self.lib.orc_select_hasher_p2(None); self._p2 = None
    def set_threads(self, n): self.lib.orc_set_threads(int(n))
    def max_threads(self): return int(self.lib.orc_max_threads())

    # ---- stage s1: generate_partial_witness over a pack (oracle/witness.c) ----
    WIT_OK, WIT_CONFLICT, WIT_INCOMPLETE, WIT_UNSUPPORTED, WIT_BAD_PACK, WIT_ZERO_INVERSE = 0, 1, 2, 3, 4, 5
    def generate_witness(self, pack_words, cells, values, public_inputs):
        """(rc, wires[num_wires, n], conflict_cell): rc 0 = every generator ran, 1 = a target was set twice with different values
        (conflict_cell = row * num_wires + wire), 2 = generators left waiting for unset targets."""
        pw = np.ascontiguousarray(pack_words, dtype=np.uint64)
        nw, n = int(pw[2]), 1 << int(pw[1])
        c = np.ascontiguousarray(cells, dtype=np.uint64); v = np.ascontiguousarray(values, dtype=np.uint64)
        p = None if public_inputs is None else np.ascontiguousarray(public_inputs, dtype=np.uint64)     # None: derived (read them from the trace)
        wires = np.zeros((nw, n), dtype=np.uint64); bad = ctypes.c_uint64()
        rc = self.lib.orc_generate_witness(_vp(pw), pw.size, _vp(c), _vp(v), c.size, None if p is None else _vp(p), _vp(wires), ctypes.byref(bad))
        return rc, wires, bad.value

    # ---- field ----
    def mul(self, a, b): return self.lib.orc_gl_mul(a, b)
    def add(self, a, b): return self.lib.orc_gl_add(a, b)
    def sub(self, a, b): return self.lib.orc_gl_sub(a, b)
    def inv(self, a): return self.lib.orc_gl_inv(a)
    def pow(self, a, e): return self.lib.orc_gl_pow(a, e)
    def root(self, log_n): return self.lib.orc_gl_root(log_n)

    # ---- transforms (return new arrays) ----
    def fft(self, a, log_n):
        x = np.array(a, dtype=np.uint64, copy=True); self.lib.orc_fft(_vp(x), log_n); return x
    def ifft(self, a, log_n):
        x = np.array(a, dtype=np.uint64, copy=True); self.lib.orc_ifft(_vp(x), log_n); return x
    def coset_fft(self, a, log_n, shift):
        x = np.array(a, dtype=np.uint64, copy=True); self.lib.orc_coset_fft(_vp(x), log_n, shift); return x
    def coset_ifft(self, a, log_n, shift):
        x = np.array(a, dtype=np.uint64, copy=True); self.lib.orc_coset_ifft(_vp(x), log_n, shift); return x
    def dft_naive(self, a, log_n):
        x = np.ascontiguousarray(a, dtype=np.uint64); out = np.empty_like(x)
        self.lib.orc_dft_naive(_vp(x), _vp(out), log_n); return out
    def fft_batch(self, a, log_n, inverse=False):
        x = np.array(a, dtype=np.uint64, copy=True).reshape(-1, 1 << log_n)
        self.lib.orc_fft_batch(_vp(x), log_n, x.shape[0], 1 if inverse else 0); return x
    def lde_batch(self, values, log_n, rate_bits, shift):
        v = np.ascontiguousarray(values, dtype=np.uint64).reshape(-1, 1 << log_n)
        coeffs = np.empty_like(v); out = np.empty((v.shape[0], 1 << (log_n + rate_bits)), dtype=np.uint64)
        self.lib.orc_lde_batch(_vp(v), log_n, rate_bits, v.shape[0], shift, _vp(coeffs), _vp(out))
        return coeffs, out

    # ---- hashing ----
    def poseidon(self, state):
        s = np.array(state, dtype=np.uint64, copy=True); assert s.size == 12
        self.lib.orc_poseidon_permute(_vp(s)); return s
    def round_constants(self):
        rc = np.empty(360, dtype=np.uint64); self.lib.orc_poseidon_round_constants(_vp(rc)); return rc
    def fast_partial(self):
        out = np.empty(1024, dtype=np.uint64); n = self.lib.orc_poseidon_fast_partial(_vp(out)); return out[:n].copy()
    def hash_n_to_m(self, inp, m):
        x = np.ascontiguousarray(inp, dtype=np.uint64); out = np.empty(m, dtype=np.uint64)
        self.lib.orc_hash_n_to_m_no_pad(_vp(x), x.size, _vp(out), m); return out
    def hash_or_noop(self, inp):
        x = np.ascontiguousarray(inp, dtype=np.uint64); out = np.empty(4, dtype=np.uint64)
        self.lib.orc_hash_or_noop(_vp(x), x.size, _vp(out)); return out
    def two_to_one(self, l, r):
        l = np.ascontiguousarray(l, dtype=np.uint64); r = np.ascontiguousarray(r, dtype=np.uint64)
        out = np.empty(4, dtype=np.uint64); self.lib.orc_two_to_one(_vp(l), _vp(r), _vp(out)); return out
    def merkle(self, leaves, cap_height):
        lv = np.ascontiguousarray(leaves, dtype=np.uint64); n, w = lv.shape
        dig = np.empty((2 * n, 4), dtype=np.uint64); cap = np.empty((1 << cap_height, 4), dtype=np.uint64)
        tot = self.lib.orc_merkle_build(_vp(lv), n, w, cap_height, _vp(dig), _vp(cap))
        return dig[:tot].copy(), cap
    def merkle_path(self, digests, n_leaves, cap_height, index):
        d = np.ascontiguousarray(digests, dtype=np.uint64); out = np.empty((64, 4), dtype=np.uint64)
        ln = self.lib.orc_merkle_path(_vp(d), n_leaves, cap_height, index, _vp(out)); return out[:ln].copy()


class Challenger:
    def __init__(self, orc):
        self.orc = orc
        self.buf = ctypes.create_string_buffer(orc.lib.orc_challenger_size())
        orc.lib.orc_challenger_init(self.buf)
    def observe(self, xs):
        x = np.ascontiguousarray(xs, dtype=np.uint64).ravel()
        self.orc.lib.orc_challenger_observe(self.buf, _vp(x), x.size)
    def get(self): return self.orc.lib.orc_challenger_get(self.buf)
    def get_n(self, n): return [self.get() for _ in range(n)]
    def pow_response(self, nonce): return self.orc.lib.orc_challenger_pow_response(self.buf, nonce)


class OracleCircuit:
    """A loaded circuit pack on the oracle side: prove / verify / stage trace."""
    def __init__(self, orc, pack_words):
        self.orc = orc
        pw = np.ascontiguousarray(pack_words, dtype=np.uint64)
        self.h = orc.lib.orc_circuit_load(_vp(pw), pw.size)
        if not self.h:
            raise ValueError("oracle rejected the circuit pack")
    def close(self):
        if self.h:
            self.orc.lib.orc_circuit_free(self.h); self.h = None
    def proof_size(self): return self.orc.lib.orc_proof_size(self.h)
    def prove(self, wires, pis, seed=0):
        w = np.ascontiguousarray(wires, dtype=np.uint64); p = np.ascontiguousarray(pis, dtype=np.uint64)
        out = np.empty(self.proof_size(), dtype=np.uint8); ln = ctypes.c_size_t()
        rc = self.orc.lib.orc_prove_seeded(self.h, _vp(w), _vp(p), seed, _vp(out), out.size, ctypes.byref(ln))
        assert rc == 0 and ln.value == out.size, (rc, ln.value, out.size)
        return out.tobytes()
    def verify(self, proof_bytes):
        b = np.frombuffer(proof_bytes, dtype=np.uint8)
        return self.orc.lib.orc_verify(self.h, _vp(b), b.size)
    def trace(self, name):
        n = self.orc.lib.orc_trace_len(name.encode())
        out = np.empty(n, dtype=np.uint64)
        if n: self.orc.lib.orc_trace_get(name.encode(), _vp(out))
        return out


class OracleProver:
    """`commit().prove()` on the CPU for one circuit: the partition and the generator list are prepared once (setup), then
    PartialWitnesses are turned into proofs one proof per thread (orc_commit_prove_many) — bench.py's cpu_baseline."""
    def __init__(self, orc, pack_words):
        self.orc = orc
        pw = np.ascontiguousarray(pack_words, dtype=np.uint64)
        rc = ctypes.c_int()
        self.h = orc.lib.orc_witness_plan_create(_vp(pw), pw.size, ctypes.byref(rc))
        if not self.h:
            raise ValueError("oracle rejected the circuit pack for witness generation (%d)" % rc.value)
        self.proof_len = orc.lib.orc_proof_size(orc.lib.orc_witness_plan_circuit(self.h))
        self.shape = (int(pw[2]), 1 << int(pw[1]))
    def close(self):
        if self.h:
            self.orc.lib.orc_witness_plan_free(self.h); self.h = None
    def generate(self, cells, values, public_inputs):
        c = np.ascontiguousarray(cells, dtype=np.uint64); v = np.ascontiguousarray(values, dtype=np.uint64); p = np.ascontiguousarray(public_inputs, dtype=np.uint64)
        wires = np.zeros(self.shape, dtype=np.uint64); bad = ctypes.c_uint64()
        rc = self.orc.lib.orc_witness_generate(self.h, _vp(c), _vp(v), c.size, _vp(p), _vp(wires), ctypes.byref(bad))
        return rc, wires, bad.value
    def commit_prove_many(self, cells, values, public_inputs):
        """values [nproofs, count], public_inputs [nproofs, num_pis] -> list of proof bytes (raises when any proof fails)."""
        c = np.ascontiguousarray(cells, dtype=np.uint64); v = np.ascontiguousarray(values, dtype=np.uint64).reshape(-1, c.size)
        p = np.ascontiguousarray(public_inputs, dtype=np.uint64).reshape(v.shape[0], -1)
        outs = np.zeros((v.shape[0], self.proof_len), dtype=np.uint8)
        bad = self.orc.lib.orc_commit_prove_many(self.h, v.shape[0], _vp(c), c.size, _vp(v), _vp(p), _vp(outs), self.proof_len)
        if bad:
            raise RuntimeError("%d of %d CPU proofs failed" % (bad, v.shape[0]))
        return [o.tobytes() for o in outs]
