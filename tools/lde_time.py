#!/usr/bin/env python3
"""Coset LDE 2^13 -> 2^16 over 4320 columns (32 proofs x 135 wires: one lockstep batch's wires oracle) and the inverse
transform that precedes it, timed per launch kind. usage: lde_time.py [log_n] [columns]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
pkg = ge.load_package()
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 13
cols = int(sys.argv[2]) if len(sys.argv) > 2 else 4320
dev = torch.device("cuda", 0)
gpu = pkg.QpGpu(0, stream=torch.cuda.current_stream(dev).cuda_stream)
n = 1 << log_n
g = torch.Generator(device=dev); g.manual_seed(1)
x = torch.randint(0, 1 << 62, (cols, n), dtype=torch.int64, device=dev, generator=g)
c = torch.empty_like(x)
y = torch.empty((cols, n << 3), dtype=torch.int64, device=dev)
def run(reps):
    for _ in range(reps):
        gpu.ntt_dev(x, c, log_n, cols, inverse=True)
        gpu.lde_dev(c, y, log_n, 3, cols, bitrev=True)
run(2); torch.cuda.synchronize(dev)
gpu.profile(True)
t0 = time.perf_counter(); run(10); torch.cuda.synchronize(dev); dt = (time.perf_counter() - t0) / 10
names = ["ntt_pass_strided", "ntt_pass_rows", "ntt_pass_single"]
prof = {k: gpu.profile_read(k) for k in names}
gpu.profile(False)
lv = cols * (n * log_n + (n << 3) * (log_n + 3))
print(f"ifft 2^{log_n} + LDE x8 over {cols} columns: {dt*1e3:.3f} ms per pair; {lv/dt/1e12:.2f} T element-levels/s "
      f"(the 2^20 x 128 transform runs at {(1<<27)*20/1.39e-3/1e12:.2f}); per proof of 135+20+16 columns: {dt*1e3*171/cols:.4f} ms")
print({k: (round(v[0] / max(v[1], 1), 4), v[1]) for k, v in prof.items()})
gpu.close()
