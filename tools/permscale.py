import sys, time, ctypes, numpy as np
sys.path.insert(0, "/root/repo")
import __graft_entry__ as ge
pkg = ge.load_package()
gpu = pkg.QpGpu(0)
for n in (16384, 65536, 131072, 262144, 524288, 1048576):
    st = np.random.default_rng(1).integers(0, pkg.P, (n, 12), dtype=np.uint64)
    d = gpu.to_device(st)
    for _ in range(2):
        gpu._check(gpu.lib.qpgpu_poseidon_permute_dev(gpu.ctx, d.ptr, n))
    gpu.sync()
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        gpu._check(gpu.lib.qpgpu_poseidon_permute_dev(gpu.ctx, d.ptr, n))
    gpu.sync()
    dt = (time.perf_counter() - t0) / reps
    print(f"n={n:8d} waves/SIMD={n/65536:5.2f}  {dt*1e6:8.1f} us  {n/dt/1e9:6.3f} Gperm/s")
    d.free()
