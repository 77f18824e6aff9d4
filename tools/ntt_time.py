#!/usr/bin/env python3
"""Times the 2^20 x 128 NTT + inverse (BASELINE configs[1]) per launch kind with the library's HIP-event profile and
prints one JSON line. Tuning knobs are read from the environment by the library (QPGPU_NTT_LOGT, QPGPU_NTT_TW).
Usage: ntt_time.py [label] [steps]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
pkg = ge.load_package()
label = sys.argv[1] if len(sys.argv) > 1 else "base"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda", 0)
gpu = pkg.QpGpu(0, stream=torch.cuda.current_stream(dev).cuda_stream)
log_n, B = 20, 128
n = 1 << log_n
g = torch.Generator(device=dev); g.manual_seed(5)
x = (torch.randint(0, 0xFFFFFFFF, (B, n), dtype=torch.int64, device=dev, generator=g) << 32) | torch.randint(0, 1 << 32, (B, n), dtype=torch.int64, device=dev, generator=g)
y = torch.empty_like(x); z = torch.empty_like(x)
for _ in range(3):
    gpu.ntt_dev(x, y, log_n, B); gpu.ntt_dev(y, z, log_n, B, inverse=True)
torch.cuda.synchronize(dev)
gpu.profile(True)
for _ in range(steps):
    gpu.ntt_dev(x, y, log_n, B); gpu.ntt_dev(y, z, log_n, B, inverse=True)
ms_s, n_s = gpu.profile_read("ntt_pass_strided"); ms_r, n_r = gpu.profile_read("ntt_pass_rows")
gpu.profile(False)
per = ms_s / n_s + ms_r / n_r
print(json.dumps({"label": label, "env": {k: v for k, v in os.environ.items() if k.startswith("QPGPU_NTT")},
                  "strided_ms": round(ms_s / n_s, 4), "rows_ms": round(ms_r / n_r, 4), "transform_ms": round(per, 4),
                  "frac_of_8TBps": round(16.0 * n * B / (per * 1e-3) / 8e12, 4), "roundtrip_ok": bool(torch.equal(z, x))}))
gpu.close()
