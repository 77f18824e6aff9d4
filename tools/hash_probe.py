"""Isolated timing of the library's hashing kernels (run under rocprofv3 --kernel-trace --stats): a 2^21-leaf tree over 135
columns (the shape of one lockstep batch's wires commitment) and the bare permutation kernel."""
import sys, numpy as np
sys.path.insert(0, "/root/repo")
import __graft_entry__ as ge
pkg = ge.load_package()
gpu = pkg.QpGpu(0)
LOG, W = 21, 135
n = 1 << LOG
cols = gpu.alloc(n * W * 8)
chunk = np.random.default_rng(1).integers(0, pkg.P, n, dtype=np.uint64)
for c in range(W):
    gpu._check(gpu.lib.qpgpu_memcpy_h2d(gpu.ctx, cols.ptr + c * n * 8, chunk.ctypes.data, n * 8))
dig = gpu.alloc(gpu.merkle_digest_count(LOG, 4) * 32)
for _ in range(3):
    gpu.merkle_build_dev(cols, n, W, LOG, 4, dig)
gpu.sync()
st = gpu.to_device(np.random.default_rng(2).integers(0, pkg.P, (1 << 20, 12), dtype=np.uint64))
for _ in range(5):
    gpu._check(gpu.lib.qpgpu_poseidon_permute_dev(gpu.ctx, st.ptr, 1 << 20))
gpu.sync()
print("ok")
