#!/usr/bin/env python3
"""Runs only the 2^20 x 128 NTT + inverse loop (BASELINE configs[1]); used under rocprofv3 --pmc to read the
HBM traffic of the NTT kernels without the rest of the proof pipeline. Usage: ntt_only.py [steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
pkg = ge.load_package()
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda", 0)
gpu = pkg.QpGpu(0, stream=torch.cuda.current_stream(dev).cuda_stream)
log_n, B = 20, 128
n = 1 << log_n
g = torch.Generator(device=dev); g.manual_seed(5)
x = (torch.randint(0, 0xFFFFFFFF, (B, n), dtype=torch.int64, device=dev, generator=g) << 32) | torch.randint(0, 1 << 32, (B, n), dtype=torch.int64, device=dev, generator=g)
y = torch.empty_like(x); z = torch.empty_like(x)
for _ in range(steps):
    gpu.ntt_dev(x, y, log_n, B)
    gpu.ntt_dev(y, z, log_n, B, inverse=True)
torch.cuda.synchronize(dev)
assert torch.equal(z, x)
gpu.close()
print("ok")
