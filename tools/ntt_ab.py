#!/usr/bin/env python3
"""Durations of the two launches of the 2^20 x 128 transform (HIP events of the library, forward + inverse alternating), for A/B runs
of NTT variants selected through the environment (QPGPU_NTT_TW = 1 running twiddle product, 3 full twiddle table).
usage: ntt_ab.py <label> [reps]"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_package()
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
gpu = pkg.QpGpu(0)
log_n, B = 20, 128
n = 1 << log_n
x = gpu.alloc(B * n * 8); y = gpu.alloc(B * n * 8); z = gpu.alloc(B * n * 8)
base = np.random.default_rng(5).integers(0, pkg.P, n, dtype=np.uint64)
for c in range(B):
    col = np.ascontiguousarray(((base * np.uint64(2 * c + 1)) % np.uint64(pkg.P)) if c else base)
    gpu._check(gpu.lib.qpgpu_memcpy_h2d(gpu.ctx, x.ptr + c * n * 8, col.ctypes.data, n * 8))
for _ in range(3):
    gpu.ntt_dev(x, y, log_n, B); gpu.ntt_dev(y, z, log_n, B, inverse=True)
gpu.sync()
gpu.profile(True)
for _ in range(reps):
    gpu.ntt_dev(x, y, log_n, B); gpu.ntt_dev(y, z, log_n, B, inverse=True)
gpu.sync()
ms_s, n_s = gpu.profile_read("ntt_pass_strided"); ms_r, n_r = gpu.profile_read("ntt_pass_rows")
gpu.profile(False)
a = np.empty(n, dtype=np.uint64); b = np.empty(n, dtype=np.uint64)
ok = True
for c in (0, 77, 127):
    gpu._check(gpu.lib.qpgpu_memcpy_d2h(gpu.ctx, a.ctypes.data, z.ptr + c * n * 8, n * 8))
    gpu._check(gpu.lib.qpgpu_memcpy_d2h(gpu.ctx, b.ctypes.data, x.ptr + c * n * 8, n * 8))
    ok = ok and bool(np.array_equal(a, b))
gpu._check(gpu.lib.qpgpu_memcpy_d2h(gpu.ctx, a.ctypes.data, y.ptr, n * 8))
print(json.dumps({"label": sys.argv[1], "strided_ms": round(ms_s / n_s, 4), "rows_ms": round(ms_r / n_r, 4), "transform_ms": round(ms_s / n_s + ms_r / n_r, 4),
                  "hbm_frac": round(16.0 * n * B / ((ms_s / n_s + ms_r / n_r) * 1e-3) / 8e12, 4), "round_trip_ok": ok, "fwd_col0_xor": hex(int(np.bitwise_xor.reduce(a)))}))
