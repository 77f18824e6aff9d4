#!/usr/bin/env python3
"""Runs only the 2^20 x 128 NTT + inverse loop (BASELINE configs[1]); used under rocprofv3 (kernel trace, --pmc passes) to read the
NTT kernels without the rest of the proof pipeline. No torch: the process holds the one ROCm runtime libqpgpu.so links
(/opt/rocm), which is also the one rocprofv3 preloads. Usage: ntt_only.py [steps]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_package()
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
gpu = pkg.QpGpu(0)
print("runtime stack:", " ".join(sorted({ln.split()[-1] for ln in open("/proc/self/maps") if any(k in ln for k in ("libamdhip64", "libhsa-runtime64", "librocprofiler-sdk"))})))
log_n, B = 20, 128
n = 1 << log_n
x = gpu.alloc(B * n * 8); y = gpu.alloc(B * n * 8); z = gpu.alloc(B * n * 8)
base = np.random.default_rng(5).integers(0, pkg.P, n, dtype=np.uint64)
for c in range(B):
    col = np.ascontiguousarray(((base * np.uint64(2 * c + 1)) % np.uint64(pkg.P)) if c else base)
    gpu._check(gpu.lib.qpgpu_memcpy_h2d(gpu.ctx, x.ptr + c * n * 8, col.ctypes.data, n * 8))
for _ in range(steps):
    gpu.ntt_dev(x, y, log_n, B)
    gpu.ntt_dev(y, z, log_n, B, inverse=True)
gpu.sync()
a = np.empty(n, dtype=np.uint64); b = np.empty(n, dtype=np.uint64)
for c in (0, 1, 63, 127):
    gpu._check(gpu.lib.qpgpu_memcpy_d2h(gpu.ctx, a.ctypes.data, z.ptr + c * n * 8, n * 8))
    gpu._check(gpu.lib.qpgpu_memcpy_d2h(gpu.ctx, b.ctypes.data, x.ptr + c * n * 8, n * 8))
    assert np.array_equal(a, b), c
gpu.close()
print("ok")
