#!/usr/bin/env python3
"""bench.py — headline measurement for the MI355X wormhole-prover backend.

Workload (BASELINE.json configs[1]): batched 2^20-point Goldilocks NTT + inverse on one MI355X,
column-major batch resident in HBM, bit-exact vs plonky2::field::fft conventions (the parity tests
prove that; here one column is re-checked against the CPU oracle after the timed region).

A "step" = forward NTT then inverse NTT over the whole batch (B columns x 2^20 points).
Algorithmic bytes: 16*N*B per direction => 32*N*B per step.

N ranks: independent batches, one per GPU, no data-path collective (the path shards by column);
rank 0 prints one JSON line. value = whole-job algorithmic GB/s.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

LOG_N = 20
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable copy)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=128, help="columns per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import __graft_entry__ as ge
    pkg = ge.load_package()

    n = 1 << LOG_N
    B = args.batch
    stream = torch.cuda.current_stream(dev)
    gpu = pkg.QpGpu(local_rank, stream=stream.cuda_stream)

    # synthetic input already resident in HBM: random canonical field elements
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank)
    hi = torch.randint(0, 0xFFFFFFFF, (B, n), dtype=torch.int64, device=dev, generator=g)  # < 2^32 - 1
    lo = torch.randint(0, 1 << 32, (B, n), dtype=torch.int64, device=dev, generator=g)
    x = (hi << 32) | lo          # hi < 0xFFFFFFFF => value < p; int64 holds the u64 bit pattern
    del hi, lo
    y = torch.empty_like(x)
    z = torch.empty_like(x)

    def step():
        gpu.ntt_dev(x, y, LOG_N, B)                 # forward, natural -> natural
        gpu.ntt_dev(y, z, LOG_N, B, inverse=True)   # inverse (1/n included)

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    ok = bool(torch.equal(z, x))  # round trip must be the identity, on every rank
    alg_bytes_step = 32.0 * n * B
    value = alg_bytes_step * args.steps * world / dt / 1e9

    roofline = None
    cpu_baseline = None
    if rank == 0:
        # roofline leg: per-kernel HIP-event timing on the same stream (separate from the timed region)
        gpu.profile(True)
        for _ in range(max(3, min(args.steps, 10))):
            step()
        ms_s, n_s = gpu.profile_read("ntt_pass_strided")
        ms_r, n_r = gpu.profile_read("ntt_pass_rows")
        gpu.profile(False)
        # one transform = one strided launch + one rows launch; each launch is credited half of the
        # transform's 16*N*B algorithmic bytes (DESIGN.md "roofline accounting")
        per_transform_ms = ms_s / max(n_s, 1) + ms_r / max(n_r, 1)
        achieved = 16.0 * n * B / (per_transform_ms * 1e-3) / 1e9
        roofline = {
            "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
            "kernel": "ntt_pass_kernel<5,5> (strided launch + rows launch = one 2^20 transform)",
            "avg_ms": {"ntt_pass_strided": round(ms_s / max(n_s, 1), 4), "ntt_pass_rows": round(ms_r / max(n_r, 1), 4)},
            "algorithmic_bytes_per_transform": 16 * n * B,
        }
        # correctness spot check against the CPU oracle (checker only; not in any timed region)
        import oracle_binding
        orc = oracle_binding.Oracle()
        col = x[0].cpu().numpy().view(np.uint64)
        ok = ok and bool(np.array_equal(y[0].cpu().numpy().view(np.uint64), orc.fft(col, LOG_N)))
        if not args.no_cpu_baseline:
            threads = max(1, min(len(os.sched_getaffinity(0)), 16))
            orc.set_threads(threads)
            sample_cols = 2 * threads
            s = x[:sample_cols].cpu().numpy().view(np.uint64).copy()
            reps = 0
            t1 = time.perf_counter()
            while True:
                f = orc.fft_batch(s, LOG_N)
                orc.fft_batch(f, LOG_N, inverse=True)
                reps += 1
                if time.perf_counter() - t1 > 10.0 or reps >= 50:
                    break
            cdt = time.perf_counter() - t1
            cpu_baseline = {
                "value": round(32.0 * n * sample_cols * reps / cdt / 1e9, 3), "unit": "GB/s",
                "cores": threads, "kind": "port",
                "sample": f"{reps} x (fwd+inv) over {sample_cols} columns of 2^20, oracle/fft.c, one column per OpenMP thread",
            }
    gpu.close()
    if not ok:
        raise SystemExit("bench.py: NTT round trip / oracle check FAILED")
    if rank == 0:
        print(json.dumps({
            "metric": "NTT HBM GB/s vs peak (2^20-point Goldilocks NTT + inverse)", "value": round(value, 1), "unit": "GB/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: 2^20-point Goldilocks NTT + inverse, column-major batch in HBM",
                       "log_n": LOG_N, "columns_per_gpu": B, "step": "fft then ifft over the batch",
                       "algorithmic_bytes_per_step": int(alg_bytes_step)},
            "roofline": roofline, "cpu_baseline": cpu_baseline,
        }))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
