"""Wall time of the wires-shaped commitment of one lockstep batch (2^21 leaves x 135 columns, cap height 4) and of its leaf level
alone, host timers around a synchronised loop. usage: leaf_time.py <label> [reps]"""
import json, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import __graft_entry__ as ge
pkg = ge.load_package()
gpu = pkg.QpGpu(0)
LOG, W = 21, 135
n = 1 << LOG
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
cols = gpu.alloc(n * W * 8)
rng = np.random.default_rng(1)
for c in range(W):
    chunk = rng.integers(0, pkg.P, n, dtype=np.uint64)
    gpu._check(gpu.lib.qpgpu_memcpy_h2d(gpu.ctx, cols.ptr + c * n * 8, chunk.ctypes.data, n * 8))
dig = gpu.alloc(gpu.merkle_digest_count(LOG, 4) * 32)
cap = gpu.merkle_build_dev(cols, n, W, LOG, 4, dig)
gpu.sync()
t0 = time.perf_counter()
for _ in range(reps):
    gpu.merkle_build_dev(cols, n, W, LOG, 4, dig)
gpu.sync()
ms = (time.perf_counter() - t0) / reps * 1e3
perms = n * 17 + n - 16
print(json.dumps({"label": sys.argv[1], "tree_ms": round(ms, 3), "G_perm_per_s": round(perms / ms / 1e6, 3), "cap0": hex(int(cap[0][0]))}))
