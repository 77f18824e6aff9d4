#!/usr/bin/env python3
"""bench.py — headline measurement for the MI355X wormhole-prover backend.

BASELINE.json metric: "Wormhole proofs/sec + ms/proof at 1/2/4/8 GPUs; NTT HBM GB/s vs peak".

Headline workload (BASELINE configs[2], "Full Wormhole proof (LDE + Poseidon Merkle commit + FRI) on 1 MI355X"), timed the way
the reference's bench is (`prover.commit(&inputs).unwrap().prove()`, wormhole/prover/benches/prover.rs:38): from CircuitInputs in
host memory to proof bytes in host memory, on the Wormhole leaf circuit restated natively (qpgpu_leaf_circuit_build:
WormholeCircuit::new statement by statement on the library's restatement of plonky2's builder; 135 wires, 80 routed,
standard_recursion_config FRI: rate 1/8, cap height 4, 28 queries, 16 PoW bits, arity 16; NoopGate padding to 2^13 rows). Per proof:
commit (fill_witness + target map, on the submitting host thread), witness generation (stage s1, device, one pass per lockstep
batch), stages s2..s12 (device). A step = S proofs per GPU, S = --streams x --batch (6 workers x 32 proofs in lockstep by default:
every worker has its HIP stream, batched workspace and host transcript thread inside the library's proving pool).
value = proofs/s over all ranks. `prove_only_resident_witness` is the narrower region rounds 1-3 reported.

N ranks (BASELINE configs[3]): independent proofs per rank; every step's proof bytes are gathered over RCCL (to rank 0, the rank
that consumes them) inside the timed region, overlapped with the next step's proving. No other data-path collective.

Also reported (BASELINE configs[1], "NTT HBM GB/s vs peak"): the 2^20-point Goldilocks NTT + inverse over 128
columns; `roofline` is for that kernel pair, measured with HIP events on the launch stream by the library.
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

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md; ~6.3 TB/s achievable copy)
STAGES = ["prove_commit_wires", "prove_partial_products", "prove_commit_zs", "prove_quotient", "prove_commit_quotient",
          "prove_openings", "prove_fri_batch", "prove_fri_commit", "prove_pow", "prove_queries"]


class EngineClock:
    """Engine clock and board power of the card a context runs on, while a leg runs: sysfs hwmon (freq1_input in Hz,
    power1_average / power1_input in microwatts) under /sys/bus/pci/devices/<qpgpu_ctx_pci_bus_id>, read by a thread every 20 ms.
    The issue bounds this file quotes are priced at the 2.4 GHz peak engine clock; under sustained load the card delivers less
    (profiles/r04_clocks.txt), so each leg also says at which clock it ran. Never fails a run: summary() is None without sysfs."""

    def __init__(self, gpu, sysfs_root="/sys/bus/pci/devices"):
        import glob as _g
        self.freq = self.power = None
        self.samples = []
        self._stop = False
        self._th = None
        self._t0 = 0.0
        try:
            base = os.path.join(sysfs_root, gpu.pci_bus_id(), "hwmon")
            for hw in sorted(_g.glob(os.path.join(base, "hwmon*"))):
                if os.path.exists(os.path.join(hw, "freq1_input")):
                    self.freq = os.path.join(hw, "freq1_input")
                    for name in ("power1_average", "power1_input"):
                        if os.path.exists(os.path.join(hw, name)):
                            self.power = os.path.join(hw, name)
                            break
                    break
        except Exception:
            pass

    @staticmethod
    def _rd(path):
        try:
            with open(path) as f:
                return int(f.read().strip())
        except (OSError, ValueError):
            return None

    def _run(self):
        while not self._stop:
            f = self._rd(self.freq)
            w = self._rd(self.power) if self.power else None
            if f:
                self.samples.append((time.perf_counter(), f / 1e6, w / 1e6 if w else None))
            time.sleep(0.02)

    def __enter__(self):
        self.samples = []
        self._stop = False
        if self.freq:
            import threading
            self._th = threading.Thread(target=self._run, daemon=True)
            self._t0 = time.perf_counter()
            self._th.start()
        return self

    def __exit__(self, *a):
        self._stop = True
        if self._th:
            self._th.join()
            self._th = None

    def summary(self, skip_s=0.1):
        """Samples later than skip_s after the start (the ramp out of idle is not the leg's clock)."""
        xs = [x for x in self.samples if x[0] - self._t0 >= skip_s] if self.samples else []
        if not xs:
            return None
        mhz = sorted(x[1] for x in xs)
        out = {"mean": round(sum(mhz) / len(mhz), 1), "min": round(mhz[0], 1), "max": round(mhz[-1], 1), "samples": len(mhz),
               "peak_used_for_bounds": 2400.0, "source": "sysfs hwmon freq1_input, 20 ms"}
        pw = [x[2] for x in xs if x[2]]
        if pw:
            out["board_power_w_mean"] = round(sum(pw) / len(pw), 1)
        return out


def ntt_leg(pkg, gpu, log_n, batch, steps):
    """BASELINE configs[1]: fwd + inverse NTT over `batch` columns of 2^log_n. Returns (GB/s, roofline dict, ok, column in, column out).
    Device memory through the library's own allocator (no torch in a one-rank run: one ROCm runtime stack in the process)."""
    n = 1 << log_n
    x = gpu.alloc(batch * n * 8); y = gpu.alloc(batch * n * 8); z = gpu.alloc(batch * n * 8)
    rng = np.random.default_rng(99)
    base = rng.integers(0, pkg.P, n, dtype=np.uint64)
    col0 = None
    for c in range(batch):                       # 128 different columns: a seeded column, multiplied column by column on the host
        col = (base * np.uint64(2 * c + 1)) % np.uint64(pkg.P) if c else base      # (wrapping product mod 2^64, reduced: any canonical values do)
        col = np.ascontiguousarray(col % np.uint64(pkg.P))
        if c == 0:
            col0 = col.copy()
        gpu._check(gpu.lib.qpgpu_memcpy_h2d(gpu.ctx, x.ptr + c * n * 8, col.ctypes.data, n * 8))

    def step():
        gpu.ntt_dev(x, y, log_n, batch)
        gpu.ntt_dev(y, z, log_n, batch, inverse=True)

    for _ in range(2):
        step()
    gpu.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    gpu.sync()
    dt = time.perf_counter() - t0
    gbs = 32.0 * n * batch * steps / dt / 1e9
    gpu.profile(True)
    for _ in range(5):
        step()
    ms_s, n_s = gpu.profile_read("ntt_pass_strided")
    ms_r, n_r = gpu.profile_read("ntt_pass_rows")
    gpu.profile(False)
    per_transform_ms = ms_s / max(n_s, 1) + ms_r / max(n_r, 1)
    achieved = 16.0 * n * batch / (per_transform_ms * 1e-3) / 1e9
    # the same loop held for about half a second with the card's clock read beside it
    clk = EngineClock(gpu)
    sustained = None
    if clk.freq:
        reps = max(steps, int(0.5 / max(dt / steps, 1e-4)))
        with clk:
            t1 = time.perf_counter()
            for _ in range(reps):
                step()
            gpu.sync()
            sdt = time.perf_counter() - t1
        sustained = {"seconds": round(sdt, 3), "fwd_inv_GBps": round(32.0 * n * batch * reps / sdt / 1e9, 1), "engine_clock_mhz": clk.summary()}
    roof = {
        # what bounds the kernel pair is VALU issue (valu_roofline below: modelled issue time / measured time), not HBM; achieved /
        # peak / frac stay the algorithmic-bytes figures BASELINE.json's metric asks for ("NTT HBM GB/s vs peak")
        "bound": "valu", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 4), "hbm_frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
        "kernel": "ntt_pass_split_kernel<5,5,*> (one strided launch + one rows launch = one 2^20 transform; each launch is credited half of the transform's 16*N*B algorithmic bytes)",
        "avg_ms": {"ntt_pass_strided": round(ms_s / max(n_s, 1), 4), "ntt_pass_rows": round(ms_r / max(n_r, 1), 4)},
        "algorithmic_bytes_per_transform": 16 * n * batch, "workload": f"2^{log_n} points x {batch} columns",
    }
    if sustained:
        roof["sustained"] = sustained
    # HBM traffic from the PMC counters cannot be collected inside this process; it is read from the committed
    # rocprofv3 --pmc summary of the same kernels on the same workload (profiles/*ntt_pmc_summary.json)
    # A summary counts only while it was measured on THIS library's NTT kernels: tools/r03_collect.py stores the hash of the
    # kernel sources (tools/kernel_id.py) in it; on a mismatch the fields stay null and `counters_stale` says why.
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from kernel_id import kernel_source_id
    kid = kernel_source_id("ntt")
    roof["kernel_source_id"] = kid
    stale = []
    try:
        import glob
        latest = sorted(glob.glob(os.path.join(ROOT, "profiles", "*ntt_pmc_summary.json")))[-1]
        pm = json.load(open(latest))
        if pm.get("kernel_source_id") != kid:
            stale.append(os.path.relpath(latest, ROOT))
        elif pm.get("algorithmic_bytes_per_transform") == 16 * n * batch:
            roof["traffic"] = pm["hbm_bytes_per_transform"]
            roof["traffic_source"] = os.path.relpath(latest, ROOT)
    except Exception:
        pass
    # VALU issue counters of the same two kernels (rocprofv3 --pmc SQ_INSTS_VALU / GRBM_GUI_ACTIVE passes, summarised by
    # tools/r02_collect.py): lane-level VALU instructions per field element of one forward transform, and the share of the
    # chip's VALU issue slots (one wave instruction per SIMD per 4 cycles) those instructions occupied
    roof["valu_insts_per_element"] = None
    roof["valu_issue_frac"] = None
    try:
        latest = sorted(glob.glob(os.path.join(ROOT, "profiles", "*ntt_valu_summary.json")))[-1]
        vs = json.load(open(latest))
        if vs.get("kernel_source_id") != kid:
            stale.append(os.path.relpath(latest, ROOT))
        elif vs.get("elements_per_transform") == n * batch:
            roof["valu_insts_per_element"] = vs["valu_insts_per_element"]
            roof["valu_issue_frac"] = vs["valu_issue_frac"]            # the 4-cycle-slot convention; kept for comparison with round 2
            # the VALU roofline with measured prices (tools/r03_collect.py): issue time of the kernels' instruction mix at the
            # per-class costs measured on this chip, over the launches' measured cycles
            roof["valu_roofline"] = {"frac": vs.get("valu_roofline_frac"),
                                     "per_kernel": {k: e.get("valu_roofline") for k, e in vs.get("kernels", {}).items() if e.get("pass", "").startswith("forward")},
                                     "lds_bank_conflict_share": {k: e.get("lds_bank_conflict_share") for k, e in vs.get("kernels", {}).items() if e.get("pass", "").startswith("forward")},
                                     "wait_any_share_of_wave_cycles": {k: e.get("wait_any_share_of_wave_cycles") for k, e in vs.get("kernels", {}).items() if e.get("pass", "").startswith("forward")}}
            roof["valu_source"] = os.path.relpath(latest, ROOT)
    except Exception:
        pass
    if stale:
        roof["counters_stale"] = "not reported: measured on other NTT kernel sources than this library's (" + ", ".join(stale) + ")"
    # round trip on every column, forward output of column 0 handed back for the oracle's check
    ok = True
    back = np.empty(n, dtype=np.uint64); orig = np.empty(n, dtype=np.uint64)
    for c in range(batch):
        gpu._check(gpu.lib.qpgpu_memcpy_d2h(gpu.ctx, back.ctypes.data, z.ptr + c * n * 8, n * 8))
        gpu._check(gpu.lib.qpgpu_memcpy_d2h(gpu.ctx, orig.ctypes.data, x.ptr + c * n * 8, n * 8))
        ok = ok and bool(np.array_equal(back, orig))
    out0 = np.empty(n, dtype=np.uint64)
    gpu._check(gpu.lib.qpgpu_memcpy_d2h(gpu.ctx, out0.ctypes.data, y.ptr, n * 8))
    for b_ in (x, y, z):
        b_.free()
    return gbs, roof, ok, col0, out0


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N rank processes (fresh interpreters, created before this
    process makes any GPU or torch.cuda call), one per GPU, with the torchrun environment; rank 0 prints the JSON line.
    Returns the exit code."""
    import socket
    import subprocess
    import torch     # device_count() does not initialise the GPU
    backend = os.environ.get("QPGPU_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if backend == "nccl" and ndev < n:
        print(f"bench.py: --gpus {n} needs {n} visible GPUs, found {ndev} (QPGPU_BENCH_BACKEND=gloo rehearses several "
              "ranks on fewer GPUs)", file=sys.stderr)
        return 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        while any(p.poll() is None for p in procs):
            time.sleep(0.2)
            failed = [p.returncode for p in procs if p.poll() is not None and p.returncode]
            if failed:                # a failed rank strands the others in a collective: stop exactly those we started
                rc = failed[0]
                for q in procs:
                    if q.poll() is None:
                        q.terminate()
                break
        for p in procs:
            p.wait()
            rc = rc or p.returncode
    except KeyboardInterrupt:
        for q in procs:
            if q.poll() is None:
                q.terminate()
        rc = 130
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60, help="timed steps (default long enough that the pool's ramp and drain are < 2 %% of the window)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--degree-bits", type=int, default=13)
    ap.add_argument("--batch-degree-bits", type=int, default=16, help="rows of the private / public batch circuits of the tree leg")
    ap.add_argument("--streams", type=int, default=6, help="proving workers per GPU (one HIP stream + host transcript thread each)")
    ap.add_argument("--batch", type=int, default=32, help="proofs a worker proves in lockstep (qpgpu_prove_batch_dev)")
    ap.add_argument("--hash-hints", type=int, default=0, help="1: commit also hands in the leaf circuit's hash-chain states computed on the host (qpgpu_leaf_hash_hints): "
                                                               "stage s1 then runs the 61 hash rows side by side and checks them (14 dependency levels instead of 120)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ntt", action="store_true")
    ap.add_argument("--headline-only", action="store_true", help="skip the per-stage, witness-generation and end-to-end legs (profiling the timed region)")
    ap.add_argument("--no-tree", action="store_true", help="skip the 64-leaf aggregation-tree leg (BASELINE configs[4])")
    ap.add_argument("--all-gather", action="store_true", help="every rank receives every rank's proof bytes (all_gather) instead of the default gather to rank 0, "
                                                               "the rank that consumes them (SURVEY.md 8e)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args.gpus))       # this process never touches a GPU: the ranks are fresh interpreters

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; they must agree")
    # PyTorch is plumbing for the multi-rank launch only (torch.distributed over RCCL). A one-rank run imports no torch: the process
    # then holds ONE ROCm runtime stack (libqpgpu.so's /opt/rocm libamdhip64 + libhsa-runtime64) instead of torch's bundled HIP next
    # to whatever HSA a profiler preloads. QPGPU_BENCH_TORCH=1 forces the torch path on one rank.
    use_torch = world > 1 or os.environ.get("QPGPU_BENCH_TORCH") == "1"
    torch = dist = dev = coll_dev = None
    backend = os.environ.get("QPGPU_BENCH_BACKEND", "nccl")
    if use_torch:
        import torch
        import torch.distributed as dist
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
        if backend == "nccl" and torch.cuda.device_count() < (int(os.environ.get("LOCAL_WORLD_SIZE", world))):
            raise SystemExit(f"bench.py: {world} ranks but only {torch.cuda.device_count()} GPUs visible (one rank per GPU)")
        # rehearsal on a one-GPU box: QPGPU_BENCH_BACKEND=gloo puts every rank on the visible GPUs round-robin and does
        # the gather on host tensors; the driver's multi-GPU runs use the default (nccl = RCCL over xGMI)
        local_rank = local_rank % torch.cuda.device_count()
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        coll_dev = dev if backend == "nccl" else torch.device("cpu")
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if backend == "nccl":
                dist.init_process_group(backend="nccl", device_id=dev)
            else:
                dist.init_process_group(backend=backend)

    import __graft_entry__ as ge
    pkg = ge.load_package()
    from concurrent.futures import ThreadPoolExecutor
    WORKERS, LOCKSTEP = max(1, args.streams), max(1, args.batch)
    S = WORKERS * LOCKSTEP                                          # proofs in flight per GPU = proofs per step per GPU
    try:
        gpu = pkg.QpGpu(local_rank, stream=torch.cuda.current_stream(dev).cuda_stream) if use_torch else pkg.QpGpu(local_rank)
    except pkg.QpGpuError as e:
        raise SystemExit(f"bench.py needs an MI355X: the product path has no CPU fallback ({e})")
    gpus = [gpu]
    # which runtime libraries this process ended up with (one line on stderr; the evidence scripts keep it)
    try:
        libs = sorted({ln.split()[-1] for ln in open("/proc/self/maps") if any(k in ln for k in ("libamdhip64", "libhsa-runtime64", "librocprofiler-sdk"))})
        print("bench.py runtime stack: " + " ".join(libs), file=sys.stderr)
    except OSError:
        pass

    # ---- the leaf circuit (setup, untimed: the reference builds the circuit in the bench's setup closure too,
    # wormhole/prover/benches/prover.rs:35-37): WormholeCircuit::new restated natively (qpgpu_leaf_circuit_build), padded with NoopGate
    # rows to 2^degree_bits rows (the reference states >= 2^12 for its circuits, common/src/circuit.rs:463-467; the restated gates
    # alone fill 2^8) ----
    d = args.degree_bits
    L = pkg.leaf
    import leaf_cases
    leaf = L.LeafCircuit(min_degree_bits=d)
    pack = leaf.pack
    circs = [pkg.Circuit(g, pack) for g in gpus]                    # per-stream workspace, no allocation while proving
    circ = circs[0]
    proof_len = circ.proof_size()
    nw_, n_ = 135, 1 << leaf.info["degree_bits"]
    mat_bytes = nw_ * n_ * 8
    outs = [np.empty(proof_len, dtype=np.uint8) for _ in range(S)]
    pool = ThreadPoolExecutor(max_workers=1)
    pool_gen = ThreadPoolExecutor(max_workers=1)
    gathered = None
    step_gather = None

    def barrier():
        if not use_torch:
            gpu.sync()          # (the pool's proofs have been waited for by then; this covers the context's own stream)
            return
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    import queue

    # S different CircuitInputs per GPU, resident in host memory as the reference's are (setup, untimed). Input 0 is the reference
    # bench's own (build_dummy_circuit_inputs, wormhole/aggregator/src/dummy_proof.rs:125-170 as used by prover.rs:31-42); the others
    # are spends that are NOT dummies: their own secrets, Merkle paths of 1..16 levels, headers and nullifiers, so that every
    # conditional binding of the circuit is live
    inputs_all = [leaf_cases.dummy_inputs(L)]
    for i in range(1, S):
        x = leaf_cases.real_inputs(L, depth=1 + (i % 16), seed=100000 * rank + i, secret_index=i % 2)
        x.exit_account_1[0] = i & 0xFF; x.exit_account_1[1] = (i >> 8) & 0xFF
        inputs_all.append(x)
    # WORKERS lockstep batches of LOCKSTEP proofs in flight: streams, circuit workspaces and transcript threads live inside
    # the library (qpgpu_pool_create_multi); the cell list of WormholeProver::commit is resolved once per worker
    prover_pool = pkg.ProvingPool(pack, workers=WORKERS, devices=[local_rank], max_batch=LOCKSTEP)
    HINTS = bool(args.hash_hints)
    cells0, values0, pis0 = leaf.commit(inputs_all[0])
    prover_pool.set_partial_cells(leaf.commit(inputs_all[0], hash_hints=True)[0] if HINTS else cells0)
    commit_buf = (np.empty(L.LT_COUNT + L.HASH_HINTS, dtype=np.uint64), np.empty(L.LT_COUNT + L.HASH_HINTS, dtype=np.uint64))
    hints_fn = L._lib().qpgpu_leaf_hash_hints
    pis_all = [np.empty(21, dtype=np.uint64) for _ in range(S)]
    commit_fn = L._lib().qpgpu_leaf_commit
    import ctypes as _ct
    commit_n, hint_n, commit_err = _ct.c_size_t(), _ct.c_size_t(), _ct.create_string_buffer(160)
    tm_ptr, cb_ptr, vb_ptr = leaf.target_map.ctypes.data, commit_buf[0].ctypes.data, commit_buf[1].ctypes.data

    def commit_and_submit(i, out):
        """`prover.commit(&inputs)?.prove()` for input i: fill_witness on this (host) thread, then stage s1 + s2..s12 in the pool
        (the values are copied at submit)."""
        if commit_fn(_ct.byref(inputs_all[i]), tm_ptr, cb_ptr, vb_ptr, L.LT_COUNT, _ct.byref(commit_n), pis_all[i].ctypes.data, commit_err) != 0:
            raise SystemExit("bench.py: commit failed: " + commit_err.value.decode())
        if HINTS:     # the hash call sites' sponge states behind the 299 assignments (qpgpu_leaf.h "hash hints"): same witness, 14 levels instead of 120
            if hints_fn(_ct.byref(inputs_all[i]), vb_ptr + 8 * commit_n.value, L.HASH_HINTS, _ct.byref(hint_n), commit_err) != 0:
                raise SystemExit("bench.py: hash hints failed: " + commit_err.value.decode())
            return prover_pool.submit_partial(commit_buf[1][:commit_n.value + hint_n.value], pis_all[i], out)
        return prover_pool.submit_partial(commit_buf[1][:commit_n.value], pis_all[i], out)

    def run_steps(k, resident=None):
        """k steps = k*S proofs through the library's proving pool. All jobs are queued ahead and the workers free-run; the main
        thread closes step j when its S proofs are written and, with several ranks, gathers that step's proof bytes over RCCL
        while the workers are already proving step j+1. resident: prove from full witnesses already in HBM instead (sub-leg)."""
        AHEAD = max(1, min(8, 3000 // S))   # steps queued ahead of the one being collected (the pool keeps at most 4096 unwaited jobs)
        ring = min(k, AHEAD + 1)            # step j writes block j % ring, i.e. reuses step j - AHEAD - 1's
        # the workers write each step's proofs straight into one pinned block per ring slot; with several ranks that block is
        # what the step's collective sends (sharding.ProofBlockGather: no per-proof copies on the host)
        nonlocal gathered, step_gather
        if step_gather is None or len(step_gather.send) < ring:
            if use_torch:
                step_gather = pkg.sharding.ProofBlockGather(S, proof_len, dist if world > 1 else None, coll_dev, blocks=AHEAD + 1,
                                                            root=None if args.all_gather or world == 1 else 0)
            else:
                step_gather = pkg.sharding.LocalProofBlocks(S, proof_len, blocks=AHEAD + 1)
        sg = step_gather

        def submit_step(j):
            if resident is not None:
                return [prover_pool.submit(resident.ptr + i * mat_bytes, pis_all[i], sg.slot(j % ring, i)) for i in range(S)]
            return [commit_and_submit(i, sg.slot(j % ring, i)) for i in range(S)]
        tickets = {j: submit_step(j) for j in range(min(k, AHEAD))}
        last = None
        step_done.clear()
        for j in range(k):
            if j + AHEAD < k:
                tickets[j + AHEAD] = submit_step(j + AHEAD)
            for t in tickets.pop(j):
                if prover_pool.wait(t, copy=False) != proof_len:
                    raise SystemExit("bench.py: a proof of unexpected length")
            if world > 1:
                _tg = time.perf_counter()
                gathered = sg.gather(j % ring)        # [world][S][proof_len] on the host (rank 0; every rank with --all-gather)
                gather_ms.append((time.perf_counter() - _tg) * 1e3)
            last = sg.slot(j % ring, 0).tobytes()
            step_done.append(time.perf_counter())
        return last

    step_done = []
    gather_ms = []
    proof = run_steps(max(args.warmup, 1))
    barrier()
    hclk = EngineClock(gpu)
    with hclk:
        t0 = time.perf_counter()
        proof = run_steps(args.steps)
        barrier()
        dt = time.perf_counter() - t0
    headline_clock = hclk.summary(skip_s=0.3)
    # spread: the timed region cut into three consecutive windows of steps (this rank's clock)
    windows = []
    if args.steps >= 3:
        cuts = [0, args.steps // 3, 2 * args.steps // 3, args.steps]
        marks = [t0] + step_done
        windows = [round((cuts[i + 1] - cuts[i]) * S * world / (marks[cuts[i + 1]] - marks[cuts[i]]), 1) for i in range(3)]
    step_ms = [round((b - a) * 1e3, 2) for a, b in zip([t0] + step_done[:-1], step_done)][:16]
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        if rank == 0:   # every rank's S proofs arrived; rank 0's own are unchanged
            assert tuple(gathered.shape) == (world, S, proof_len) and gathered[0, 0].numpy().tobytes() == proof
        gm = torch.tensor([float(np.mean(gather_ms)) if gather_ms else 0.0], dtype=torch.float64, device=coll_dev)
        gml = [torch.zeros_like(gm) for _ in range(world)]
        dist.all_gather(gml, gm)
        gather_ms_per_rank = [round(float(x.item()), 2) for x in gml]
    else:
        gather_ms_per_rank = []
    value = args.steps * S * world / dt

    extra = {}
    ok = True
    # the same S proofs from FULL witnesses already resident in HBM (stage s1 outside the timed region): what rounds 1-3 reported
    # as the headline; kept as a sub-key. The witnesses are generated here by the batched PartialWitness entry.
    w_all = gpu.alloc(S * mat_bytes)
    gen_c = pkg.Circuit(gpu, pack, max_batch=min(S, 32))
    vals_all = np.stack([leaf.commit(x)[1] for x in inputs_all])
    pis_mat = np.stack([leaf.commit(x)[2] for x in inputs_all])
    for k0 in range(0, S, 32):
        k1 = min(S, k0 + 32)
        st_ = gen_c.generate_witness_partial_batch_dev(cells0, vals_all[k0:k1], pis_mat[k0:k1], w_all.ptr + k0 * mat_bytes)
        if any(st_):
            raise SystemExit("bench.py: witness generation failed: " + gpu.last_error())
    gpu.sync()
    gen_c.close()
    wires0 = np.empty((nw_, n_), dtype=np.uint64)                # witness 0 as proved, for the oracle's byte-parity check below
    gpu._check(gpu.lib.qpgpu_memcpy_d2h(gpu.ctx, wires0.ctypes.data, w_all.ptr, mat_bytes))
    wires, pis = wires0, pis0
    w_t = gpu.to_device(wires0)
    rs = max(6, min(args.steps, 20))
    run_steps(2, resident=w_all)
    barrier()
    tr = time.perf_counter()
    proof_res = run_steps(rs, resident=w_all)
    barrier()
    dtr = time.perf_counter() - tr
    if world > 1:
        t = torch.tensor([dtr], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dtr = float(t.item())
    ok = ok and proof_res == proof
    extra["prove_only_resident_witness"] = {"proofs_per_s": round(rs * S * world / dtr, 1), "steps": rs,
                                            "note": "stages s2..s12 from full witnesses resident in HBM (the timed region of rounds 1-3's headline); "
                                                    "`value` above also contains commit (a1, host) and witness generation (s1, device)"}

    # ---- BASELINE configs[4]: the recursive aggregator tree, shape-equivalent ----
    # 64 leaf proofs -> 8 private batches of 8 (2^16 rows, zero-knowledge) -> 1 public batch of 8 (2^16 rows), reference
    # call stack SURVEY 3.4. The recursive circuits themselves need the Rust builder, so every level proves a synthetic
    # circuit of the level's size and configuration; what is real: the sharding, the per-level gather of proof bytes and
    # the proving work per level. Witness generation of the recursive verifiers (host work in the reference) is not included.
    tree = None
    tree_check = None
    if not args.no_tree:
        agg = pkg.aggregation
        # recursive-verifier gate mix (Poseidon, extension arithmetic, Reducing*, RandomAccess, Exponentiation, PoseidonMds,
        # CosetInterpolation); private batch: standard_recursion_zk_config with 60 routed wires
        # (reference common/src/circuit.rs:396-402), public batch: standard_recursion_config
        rec = dict(poseidon=True, base_sum=True, ext_arith=True, recursion=True)
        NB_PIS = 21 * 8 + 8
        tleaf = pkg.synth_circuit(d, num_wires=135, num_routed=80, num_public_inputs=21, seed=1000, poseidon=True, base_sum=True, poseidon2=True)
        tpriv = pkg.synth_circuit(args.batch_degree_bits, num_wires=135, num_routed=60, num_public_inputs=NB_PIS, seed=78, **rec)
        tpriv[0][14] = 1
        tpub = None
        if rank == 0:
            tpub = pkg.synth_circuit(args.batch_degree_bits, num_wires=135, num_routed=80, num_public_inputs=agg.public_batch_pi_len(8, 8), seed=77, **rec)
        atree = agg.AggregationTree(pkg, gpus[0], rank, world, tleaf, tpriv, tpub, leaf_batch=64)
        dd = dist if world > 1 else None
        tree_exchange = "all" if args.all_gather else "root"
        atree.run(dd, coll_dev, exchange=tree_exchange)
        barrier()
        t2 = time.perf_counter()
        t_leaves, t_batches, t_root = atree.run(dd, coll_dev, exchange=tree_exchange)
        barrier()
        tdt = time.perf_counter() - t2
        if world > 1 and tree_exchange == "root":   # untimed: the checker below wants every rank's leaf proofs on rank 0
            mine = [p_ for p_ in t_leaves if p_ is not None]
            got_ = pkg.sharding.gather_proof_bytes(mine, dd, coll_dev, root=0)
            if got_ is not None:
                t_leaves = [p_ for r_ in got_ for p_ in r_]
        if world > 1:
            tt = torch.tensor([tdt], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            tdt = float(tt.item())
        assert len(t_leaves) == 64 and len(t_batches) == 8
        tree = {"leaves": 64, "private_batches": 8, "public_batches": 1, "seconds": round(tdt, 4),
                "trees_per_s": round(1.0 / tdt, 3), "levels_rank0": dict(atree.times),
                "witness_dependency_levels": {k: int(p_.circ.witness_info()[1]) for k, p_ in (("leaf", atree.leaf), ("private", atree.private), ("public", atree.public)) if p_ is not None},
                "shape": f"leaf 2^{d} rows (80 routed); private batch 2^{args.batch_degree_bits} rows zero-knowledge, 60 routed wires; "
                         f"public batch 2^{args.batch_degree_bits} rows, 80 routed; 135 wires; batches carry the 14-gate recursive-verifier mix",
                "note": "shape-equivalent synthetic circuits per level; each level parses the previous level's gathered proof bytes, "
                        "runs the reference's admission checks, pads / shuffles, derives the public inputs the wrapper circuit would emit, regenerates its witness on the device (stage s1) and proves — a rank's "
                        "leaves (up to 64) and its private batches (up to 8) each as one lockstep batch; the inner proofs are not verified in-circuit; stage s1 of the 2^16-row stand-ins is bound by their dependency depth (witness_dependency_levels x ~40 us per "
                        "level: its slowest generator, e.g. a lane-cooperative PoseidonGate row; one pass for all proofs of a lockstep batch) and is most of the private and public levels' time; reference (paper/main.tex:449-492): 64*0.020 + 8*5.39 + 3.84 = 48 s "
                        "sequential on an M2 Max"}
        if rank == 0:
            # checker, untimed: the oracle verifies the root and one proof per level, and the root's public inputs hold the 8
            # batch proofs' public inputs in rank order (they in turn the leaves')
            import oracle_binding as ob
            orc_t = ob.Oracle()
            ok_t = True
            for pk, pf in ((tleaf[0], t_leaves[63]), (tpriv[0], t_batches[7]), (tpub[0], t_root)):
                oc_t = ob.OracleCircuit(orc_t, pk)
                ok_t = ok_t and oc_t.verify(pf) == 0
                oc_t.close()
            # the root's public inputs (12 + 14*8*8 felts, public_batch/circuit/constants.rs) forward every batch proof's exit
            # slots and nullifiers in rank order; a batch's exit slots carry its 8 leaves' amounts merged per account
            rp = agg.proof_public_inputs(t_root, agg.public_batch_pi_len(8, 8))
            ok_t = ok_t and int(rp[11]) == 128 and tuple(rp[6:10].tolist()) == agg.TEST_BLOCK_HASH
            for b in range(8):
                bp = agg.proof_public_inputs(t_batches[b], NB_PIS)
                ok_t = ok_t and bool(np.array_equal(rp[12 + 80 * b:12 + 80 * (b + 1)], bp[8:88]))
                ok_t = ok_t and bool(np.array_equal(rp[12 + 640 + 32 * b:12 + 640 + 32 * (b + 1)], bp[88:120]))
                want = {}
                for j in range(8):
                    lp = agg.proof_public_inputs(t_leaves[8 * b + j], 21)
                    for acct, amt in ((tuple(lp[8:12].tolist()), int(lp[1])), (tuple(lp[12:16].tolist()), int(lp[2]))):
                        want[acct] = want.get(acct, 0) + amt
                got = {tuple(bp[9 + 5 * k:13 + 5 * k].tolist()): int(bp[8 + 5 * k]) for k in range(16) if int(bp[8 + 5 * k])}
                ok_t = ok_t and got == {a: v for a, v in want.items() if v}
            tree["checked"] = ("oracle verifier accepts leaf 63, private batch 7 and the root; the root forwards the 8 batch proofs' exit slots and "
                               "nullifiers in rank order and every batch's exit slots are its leaves' amounts merged per account") if ok_t else "FAILED"
            tree_check = ok_t
        atree.close()
    # The same 64-leaf shape with circuits that CHECK something: leaves of the restated Wormhole leaf circuit from CircuitInputs,
    # 8 first-level circuits over 8 leaf proofs each carrying the PRIVATE-BATCH logic, 1 second-level circuit over the 8 first-level
    # proofs carrying the PUBLIC-BATCH logic; every wrapper also verifies the Merkle half of its inner proofs in-circuit and replays
    # their transcripts (qpgpu_wrapper_circuit_build; what that leaves out is listed in csrc/wrapper_circuit.cpp).
    attest = None
    if not args.no_tree:
        try:
            at = pkg.recursion.AttestingTree(pkg, gpu, per_batch=8, batches=8, rank=rank, world=world, aggregator_address=bytes([3] * 32), zero_knowledge=True)
            # the same 64 inputs on every rank: 48 real spends of ONE block (their leaves in one 4-ary tree of depth 3), every spend
            # with its own exit accounts, 6 per batch; the reference's dummy in the two other slots of a batch, at moving positions
            rng_a = np.random.default_rng(4)
            def acct():
                b_ = rng_a.integers(0, 256, 32, dtype=np.uint8); b_[7::8] &= 0x7F
                return b_.tobytes()
            spends = leaf_cases.shared_tree_inputs(L, 48, depth=3, seed=9, exits=[(acct(), acct()) for _ in range(48)])
            at_inputs = []
            for b_ in range(8):
                slots_ = spends[6 * b_:6 * b_ + 6]
                slots_.insert(b_ % 7, inputs_all[0]); slots_.insert((3 * b_ + 1) % 8, inputs_all[0])
                at_inputs += slots_
            dd_ = dist if world > 1 else None
            at.run(at_inputs, dd_, coll_dev)
            barrier()
            ta = time.perf_counter()
            a_leaves, a_l1, a_root = at.run(at_inputs, dd_, coll_dev)
            barrier()
            adt = time.perf_counter() - ta
            if world > 1:
                tt_ = torch.tensor([adt], dtype=torch.float64, device=coll_dev)
                dist.all_reduce(tt_, op=dist.ReduceOp.MAX)
                adt = float(tt_.item())
            ok_a = all(at.leaf_ver.verify(p_) for p_ in a_leaves[-1:])
            parsed = None
            if rank == 0:
                ok_a = ok_a and bool(at.w2_ver.verify(a_root)) and bool(at.w1_ver.verify(a_l1[7]))
                n_root = pkg.aggregation.public_batch_pi_len(8, 8)
                root_pis = np.frombuffer(a_root[-8 * n_root:], dtype=np.uint64)
                want_root = at.expected_root_public_inputs(np.stack([at.leaf.commit(x_)[2] for x_ in at_inputs]))
                ok_a = ok_a and bool(np.array_equal(root_pis, want_root))
                hdr_, slots_, nulls_ = pkg.aggregation.parse_public_batch_public_inputs(root_pis, 8, 8)
                parsed = {"total_exit_slots": hdr_["total_exit_slots"], "block_number": hdr_["block_number"], "nonzero_exit_slots": sum(1 for s_ in slots_ if s_[0]),
                          "summed_output_amount": sum(s_[0] for s_ in slots_), "nullifiers": len(nulls_)}
                ok_a = ok_a and parsed["nonzero_exit_slots"] == 96 and parsed["summed_output_amount"] == 48 * 297 and hdr_["aggregator_address"] == bytes([3] * 32)
            attest = {"leaves": 64, "real_spends": 48, "first_level": 8, "second_level": 1, "first_level_zero_knowledge": True, "first_level_blinding_rows": at.w1.info["rows_blinding"], "seconds": round(adt, 4), "levels_rank0": dict(at.times), "ranks": world,
                      "degree_bits": {"leaf": at.leaf.info["degree_bits"], "first_level": at.w1.info["degree_bits"], "second_level": at.w2.info["degree_bits"]},
                      "poseidon_gate_rows": {"first_level": at.w1.info["rows_poseidon"], "second_level": at.w2.info["rows_poseidon"]},
                      "root_public_inputs": parsed,
                      "per_proof_ms": {"private_batch_N8_zero_knowledge_lockstep8": round(1e3 * at.times["first_level_s"] / max(1, len(at.my_batches)), 2),
                                       "public_batch_M8": round(1e3 * at.times["second_level_s"], 2)},
                      "reference_published": {"private_batch_N8_s": 5.39, "public_batch_M8_s": 3.84, "tree_64_leaves_sequential_s": 48, "hw": "Apple M2 Max 12c",
                                              "source": "paper/main.tex:466-470,488-492; BASELINE.md section 1 (the fork's own circuits, whose sizes are 2^16 for both layers)"},
                      "checked": "the library's verifier accepts the root, a first-level proof and a leaf; the root's public inputs are the PublicBatchPublicInputs the host "
                                 "restatement of the two layers' logic computes from the 64 leaves' public inputs, and parse (96 paid exit slots, 64 nullifiers)" if ok_a else "FAILED",
                      "arithmetic_extension_rows": {"first_level": at.w1.info["rows_before_padding"] - at.w1.info["rows_poseidon"] - at.w1.info["rows_random_access"] - at.w1.info["rows_base_sum"] - at.w1.info["rows_arithmetic"] - at.w1.info["rows_constant"]},
                      "note": "every wrapper is a complete recursive verifier of its inner proofs (QPGPU_WRAPPER_VERIFY): for each inner proof it hashes the public inputs, replays "
                              "the Fiat-Shamir transcript in-circuit (challenges, proof-of-work check, query indices), checks in every one of the 28 query rounds that the four opened "
                              "rows and every FRI step's coset hash up their Merkle paths to the committed caps, evaluates the openings against the vanishing polynomial at zeta "
                              "(every gate of the inner circuit, permutation argument; the same generic expressions as the host verifier, csrc/verify_math.hpp) and the FRI "
                              "consistency arithmetic (reduced openings, coset interpolation per step, final polynomial); first level + the private-batch circuit's own "
                              "constraints (dummy flags, block / asset / fee consistency, exit-account grouping, distinct real nullifiers, dummy nullifiers = H(H(preimage)), "
                              "sorting network) and built zero-knowledge as the reference's private layer is (60 routed wires, CircuitBuilder::blind's rows with fresh random "
                              "wires per proof, salted Merkle leaves), second level + the public-batch circuit's, restated on the native builder (circuit_logic.rs of each "
                              "layer). The gate set, row order and blinding counts are the native builder's / upstream plonky2's, not the fork's (verifier data differs). "
                              "Times include commit on the host (fill_witness; fill_private_batch_witness per inner proof)."}
            ok = ok and ok_a
            at.close()
        except pkg.QpGpuError as e:
            attest = {"error": str(e)}
    if rank == 0:
        extra["attesting_tree"] = attest
        extra["aggregation_tree"] = tree
        ok = ok and tree_check is not False
        def extra_legs():
            nonlocal ok
            # per-stage breakdown (HIP events recorded by the library on the launch stream; separate leg)
            gpu.profile(True)
            for _ in range(5):
                circ.prove_dev(w_t, pis, outs[0])       # one proof alone on the GPU: per-stage latency
            stages = {}
            for s in STAGES:
                ms, cnt = gpu.profile_read(s)
                stages[s] = round(ms / max(cnt, 1), 4)
            leaf_ms, leaf_n = gpu.profile_read("merkle_leaf_hash")
            gpu.profile(False)
            # the host-buffer entry (witness handed over in pageable host memory: H2D over PCIe inside the call, device
            # copies scrubbed afterwards); reported for DESIGN.md, never the headline
            circ.prove(wires, pis)
            th = time.perf_counter()
            for _ in range(3):
                circ.prove(wires, pis)
            extra["host_witness_ms_per_proof"] = round((time.perf_counter() - th) / 3 * 1e3, 3)
            # what the optional witness check costs (one pass over the trace rows + one sync): single proof, on vs off
            def _t(n=5):
                t_ = time.perf_counter()
                for _ in range(n):
                    circ.prove_dev(w_t, pis, outs[0])
                return (time.perf_counter() - t_) / n * 1e3
            _t(2); off_ms = _t()
            circ.set_witness_check(True); _t(2); on_ms = _t(); circ.set_witness_check(False)
            extra["witness_check"] = {"single_proof_ms_off": round(off_ms, 3), "single_proof_ms_on": round(on_ms, 3),
                                      "note": "qpgpu_circuit_set_witness_check: filtered gate constraints on every trace row + permutation "
                                              "product closure before the quotient stage; returns QPGPU_EUNSAT naming the row"}
            extra["single_proof_latency_ms"] = round(sum(stages.values()), 4)
            # the same region as the headline for ONE proof alone on the device: commit (host) + stage s1 + stages s2..s12 + bytes back
            def _one(i_):
                c_, v_, p_ = leaf.commit(inputs_all[i_])
                circ.generate_witness_partial_dev(c_, v_, p_, w_t)
                return circ.prove_dev(w_t, p_, outs[0])
            _one(1); _one(2)
            t1_ = time.perf_counter()
            for i_ in range(10):
                _one(1 + i_ % (S - 1))
            extra["single_proof_commit_prove_ms"] = round((time.perf_counter() - t1_) / 10 * 1e3, 3)
            # the same with the front-end's hash hints (include/qpgpu_leaf.h): the hash chains' states computed on the host ride along as
            # extra assignments, stage s1 runs the 61 hash rows side by side and checks them — 14 dependency levels instead of 120
            def _one_hinted(i_):
                c_, v_, p_ = leaf.commit(inputs_all[i_], hash_hints=True)
                circ.generate_witness_partial_dev(c_, v_, p_, w_t)
                return circ.prove_dev(w_t, p_, outs[0])
            same_ = _one_hinted(1) == _one(1); _one_hinted(2)
            t1_ = time.perf_counter()
            for i_ in range(10):
                _one_hinted(1 + i_ % (S - 1))
            extra["single_proof_commit_prove_ms_hash_hints"] = {"ms": round((time.perf_counter() - t1_) / 10 * 1e3, 3), "same_proof_bytes": bool(same_),
                                                                "dependency_levels": circ.witness_info()[1]}
            ok = ok and same_
            _one(0)                                      # (w_t holds input 0's witness again)
            extra["proof_stage_ms"] = stages
            extra["merkle_leaf_hash_avg_ms"] = round(leaf_ms / max(leaf_n, 1), 4)
            # the kernel that dominates a proof is integer-ALU-bound, not HBM-bound: its rate next to the bytes it moves
            # measured on the leaf-hash kernel itself (HIP events of the library's "merkle_leaf_hash" stage): a 2^20-leaf tree over
            # 135 columns, the shape of a lockstep batch's wires commitment (17 permutations per leaf)
            HL = 20
            d_cols = gpu.alloc((135 << HL) * 8)
            col_ = np.random.default_rng(3).integers(0, pkg.P, 1 << HL, dtype=np.uint64)
            for c_ in range(135):
                gpu._check(gpu.lib.qpgpu_memcpy_h2d(gpu.ctx, d_cols.ptr + (c_ << HL) * 8, col_.ctypes.data, col_.nbytes))
            d_dig = gpu.alloc(gpu.merkle_digest_count(HL, 4) * 32)
            gpu.merkle_build_dev(d_cols, 1 << HL, 135, HL, 4, d_dig); gpu.sync()
            gpu.profile(True)
            for _ in range(3):
                gpu.merkle_build_dev(d_cols, 1 << HL, 135, HL, 4, d_dig)
            gpu.sync()
            hash_ms, hash_n = gpu.profile_read("merkle_leaf_hash")
            gpu.profile(False)
            perm_rate = 17 * (1 << HL) * hash_n / (hash_ms * 1e-3)
            # the same commitment held for about half a second with the card's clock read beside it (HIP events as above)
            pclk = EngineClock(gpu)
            hash_sustained = None
            if pclk.freq:
                gpu.profile(True)
                with pclk:
                    t1_ = time.perf_counter()
                    while time.perf_counter() - t1_ < 0.5:
                        for _ in range(8):
                            gpu.merkle_build_dev(d_cols, 1 << HL, 135, HL, 4, d_dig)
                        gpu.sync()
                sh_ms, sh_n = gpu.profile_read("merkle_leaf_hash")
                gpu.profile(False)
                hash_sustained = {"seconds": round(time.perf_counter() - t1_, 3), "permutations_per_s": round(17 * (1 << HL) * sh_n / (sh_ms * 1e-3) / 1e9, 3),
                                  "engine_clock_mhz": pclk.summary()}
            d_cols.free(); d_dig.free()
            lde_leaves = 1 << (d + 3)
            mx_on = os.environ.get("QPGPU_MX", "1") != "0"
            extra["poseidon_hashing"] = {"bound": "valu (full rounds) beside mfma (the 22 partial rounds as one int8 GEMM)" if mx_on else "valu",
                                         "permutations_per_s": round(perm_rate / 1e9, 3), "unit": "G/s",
                                         "kernel": ("mx::leaf_hash_kernel" if mx_on else "tp::leaf_hash_kernel<PoseidonV1>") + ", 2^20 leaves x 135 columns",
                                         "wires_leaf_hash_algorithmic_GBps": round(8.0 * 135 * lde_leaves / 1e9 / (17 * lde_leaves / perm_rate), 1),
                                         "note": "leaf hashing reads 8*W bytes per leaf and runs ceil(W/8) permutations: at the permutation "
                                                 "rate above the wires oracle streams this many GB/s, far below HBM"}
            if hash_sustained:
                extra["poseidon_hashing"]["sustained"] = hash_sustained
            try:   # hardware counters of the same kernel, committed (tools/gpurun_scripts/mx_pmc.sh -> tools/collect_mx_counters.py)
                import glob as _glob
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                from kernel_id import kernel_source_id
                latest_mx = sorted(_glob.glob(os.path.join(ROOT, "profiles", "*mx_leaf_hash_counters.json")))[-1]
                with open(latest_mx) as f:
                    lk = json.load(f)
                if mx_on and lk.get("kernel_source_id") == kernel_source_id("hash_mx"):
                    ph = extra["poseidon_hashing"]
                    ph.update({k: lk[k] for k in ("valu_insts_per_permutation", "mfma_insts_per_permutation", "mfma_busy_frac") if k in lk})
                    ph["counters_source"] = os.path.relpath(latest_mx, ROOT)
                    # the VALU issue bound of this instruction count: a wave instruction serves 64 permutations, a SIMD issues one VALU
                    # instruction per 3.5 cycles at this mix (measured: 3.51, `cycles_per_valu_inst_per_simd` in the counters file),
                    # 1024 SIMDs at the 2.4 GHz peak engine clock
                    bound = 1024 * 64 / (lk["valu_insts_per_permutation"] * 3.5) * 2.4
                    ph["valu_issue_bound_G_per_s"] = round(bound, 3)
                    ph["frac"] = round(perm_rate / 1e9 / bound, 3)
                    if hash_sustained and hash_sustained["engine_clock_mhz"]:
                        ck = hash_sustained["engine_clock_mhz"]["mean"]
                        hash_sustained["valu_issue_bound_at_observed_clock_G_per_s"] = round(bound * ck / 2400.0, 3)
                        hash_sustained["frac_at_observed_clock"] = round(hash_sustained["permutations_per_s"] / (bound * ck / 2400.0), 3)
                    ph["frac_note"] = ("measured permutations/s over the VALU issue bound of the kernel's own instruction count (valu_insts_per_permutation x 3.5 cycles per wave "
                                       "instruction per SIMD, 1024 SIMDs, 2.4 GHz); the rest is the clock under load, 40 bytes per lane of spilled registers at the 128-register cap "
                                       "(four waves per SIMD; the register file is unified on this chip: no spare AGPRs) and the sponge's column loads")
                else:
                    extra["poseidon_hashing"]["counters_stale"] = "not reported: measured on other hashing kernel sources than this library's (" + os.path.relpath(latest_mx, ROOT) + ")"
            except (OSError, KeyError, ValueError, ImportError, IndexError):
                pass

            # stage s1 alone: PartialWitness (299 assignments + 21 public inputs) -> full witness, one at a time and a lockstep batch
            gens, levels, free = circ.witness_info()
            cb_ = pkg.Circuit(gpu, pack, max_batch=LOCKSTEP)
            dB = gpu.alloc(LOCKSTEP * mat_bytes)
            vB = np.stack([vals_all[i % S] for i in range(LOCKSTEP)]); pB = np.stack([pis_mat[i % S] for i in range(LOCKSTEP)])
            cb_.witness_partial_prepare(cells0, LOCKSTEP)
            cb_.generate_witness_partial_batch_dev(cells0, vB[:1], pB[:1], dB); gpu.sync()
            s1_ok = bool((dB.download(count=nw_ * n_) == wires0.ravel()).all())
            tw = time.perf_counter()
            for _ in range(5):
                cb_.generate_witness_partial_batch_dev(cells0, vB[:1], pB[:1], dB)
            gpu.sync()
            s1_single = (time.perf_counter() - tw) / 5
            cb_.generate_witness_partial_batch_dev(cells0, vB, pB, dB); gpu.sync()
            tw = time.perf_counter()
            for _ in range(3):
                cb_.generate_witness_partial_batch_dev(cells0, vB, pB, dB)
            gpu.sync()
            s1_batch = (time.perf_counter() - tw) / 3
            # the same with the front-end's hash hints behind the 299 assignments (host: + the hash chains' states; device: the plan built for
            # that list, the hash rows side by side and checked)
            comH = [leaf.commit(inputs_all[i % S], hash_hints=True) for i in range(LOCKSTEP)]
            vH = np.stack([c_[1] for c_ in comH])
            cb_.witness_partial_prepare(comH[0][0], LOCKSTEP)
            cb_.generate_witness_partial_batch_dev(comH[0][0], vH[:1], pB[:1], dB); gpu.sync()
            h_ok = bool((dB.download(count=nw_ * n_) == wires0.ravel()).all())
            h_levels = cb_.witness_info()[1]
            tw = time.perf_counter()
            for _ in range(5):
                cb_.generate_witness_partial_batch_dev(comH[0][0], vH[:1], pB[:1], dB)
            gpu.sync()
            h_single = (time.perf_counter() - tw) / 5
            cb_.generate_witness_partial_batch_dev(comH[0][0], vH, pB, dB); gpu.sync()
            tw = time.perf_counter()
            for _ in range(3):
                cb_.generate_witness_partial_batch_dev(comH[0][0], vH, pB, dB)
            gpu.sync()
            h_batch = (time.perf_counter() - tw) / 3
            dB.free(scrub=True); cb_.close()
            tc_ = time.perf_counter()
            for _ in range(2000):
                leaf.commit(inputs_all[1])
            commit_us = (time.perf_counter() - tc_) / 2000 * 1e6
            tc_ = time.perf_counter()
            for _ in range(500):
                leaf.commit(inputs_all[1], hash_hints=True)
            commit_h_us = (time.perf_counter() - tc_) / 500 * 1e6
            extra["witness_generation"] = {"generator_instances": gens, "dependency_levels": levels, "caller_supplied_cells": free,
                                           "commit_host_us": round(commit_us, 2), "single_ms": round(s1_single * 1e3, 3), "batch": LOCKSTEP,
                                           "batched_ms_per_witness": round(s1_batch / LOCKSTEP * 1e3, 4), "equals_full_witness": s1_ok,
                                           "note": "commit = qpgpu_leaf_commit through ctypes (fill_witness + target map, host); s1 = "
                                                   "qpgpu_generate_witness_partial_batch_dev on the restated leaf circuit: one kernel launch per dependency level",
                                           "with_hash_hints": {"assignments": int(comH[0][0].size), "dependency_levels": h_levels, "commit_host_us": round(commit_h_us, 2),
                                                               "single_ms": round(h_single * 1e3, 3), "batched_ms_per_witness": round(h_batch / LOCKSTEP * 1e3, 4),
                                                               "equals_full_witness": h_ok}}
            ok = ok and s1_ok and h_ok

            def leaf_throughput(circuit_kw, reps=6, hasher_note=None):
                """commit + prove at full throughput on another build of the leaf circuit (same inputs, workers, lockstep)."""
                lc_ = L.LeafCircuit(**circuit_kw)
                pl = pkg.ProvingPool(lc_.pack, workers=WORKERS, devices=[local_rank], max_batch=LOCKSTEP)
                cm = [lc_.commit(x) for x in inputs_all]
                pl.set_partial_cells(cm[0][0])
                o_ = [np.empty(pl.proof_size(), dtype=np.uint8) for _ in range(S)]
                dt_ = None
                for rep_ in range(2):
                    t_ = time.perf_counter()
                    tk = [pl.submit_partial(cm[i % S][1], cm[i % S][2], o_[i % S]) for i in range(reps * S)]
                    for t2 in tk:
                        pl.wait(t2, copy=False)
                    dt_ = time.perf_counter() - t_
                pl.close()
                return lc_, cm, o_, {"proofs_per_s": round(reps * S / dt_, 1), "ms_per_proof": round(dt_ / (reps * S) * 1e3, 4), "degree_bits": lc_.info["degree_bits"]}
            # the reference states ">= 2^12" for its circuits (common/src/circuit.rs:463-467): the same measurement at 2^12 rows, and at the
            # size the restated gates alone need (2^8: what this builder's leaf circuit costs without padding)
            for key, kw in (("degree_bits_12", dict(min_degree_bits=12)), ("unpadded_circuit", dict(min_degree_bits=0))):
                try:
                    extra[key] = leaf_throughput(kw)[3]
                except pkg.QpGpuError as e:
                    extra[key] = {"error": str(e)}
            extra["unpadded_circuit"]["note"] = ("the restated circuit's own size: %d gate rows -> 2^%d; the headline pads it to 2^%d because the reference states its "
                                                 "circuits are >= 2^12 rows (the fork's builder and Poseidon2 gate cannot be read offline)" % (leaf.info["rows_before_padding"], extra["unpadded_circuit"].get("degree_bits", 0), d))
            # The same at full throughput under the OTHER candidate proof-system hasher: the reference does not tell which
            # permutation backs PoseidonGoldilocksConfig in the fork, so the hasher is a plug; this is the headline's
            # counterpart should it be Poseidon2 (qp-poseidon-core's parameters, pinned by the reference's seven vectors): the
            # public-input hash is then built from Poseidon2 gate rows too. One proof is held against the oracle.
            try:
                qp_ = pkg.poseidon2_qp_params()
                pkg.set_hasher_poseidon2(*qp_)
                try:
                    lq, cmq, oq, resq = leaf_throughput(dict(min_degree_bits=d, inner_hasher=1))
                    import oracle_binding as _ob
                    orq = _ob.Oracle(); orq.select_poseidon2(*qp_)
                    rcq, wq, _ = orq.generate_witness(lq.pack, cmq[0][0], cmq[0][1], cmq[0][2])
                    ocq = _ob.OracleCircuit(orq, lq.pack)
                    okq = bool(rcq == 0 and ocq.prove(wq, cmq[0][2]) == oq[0].tobytes())
                    ocq.close(); orq.select_poseidon()
                    extra["poseidon2_hasher"] = dict(resq, bytes_equal_oracle=okq,
                                                     note="same circuit, inputs, workers and lockstep batches as the headline; Merkle trees / challenger / proof of work / public-input "
                                                          "hash under Poseidon2 with qp-poseidon-core's parameters (large launches on the matrix-pipe build as well)")
                    ok = ok and okq
                finally:
                    pkg.set_hasher_poseidon()
            except pkg.QpGpuError as e:
                extra["poseidon2_hasher"] = {"error": str(e)}

        if not args.headline_only:
            extra_legs()
            # the same stages for one lockstep batch (HIP events around each stage of the whole batch)
            cb = pkg.Circuit(gpu, pack, max_batch=LOCKSTEP)
            ptrs = [w_all.ptr + (i % S) * mat_bytes for i in range(LOCKSTEP)]
            cb.prove_batch_dev(ptrs, pis_all[:LOCKSTEP] if S >= LOCKSTEP else [pis_all[i % S] for i in range(LOCKSTEP)])
            gpu.profile(True)
            for _ in range(3):
                cb.prove_batch_dev(ptrs, [pis_all[i % S] for i in range(LOCKSTEP)])
            # ... preceded by stage s1 for the same batch (PartialWitnesses -> full witnesses)
            dW = gpu.alloc(LOCKSTEP * mat_bytes)
            vW = np.stack([vals_all[i % S] for i in range(LOCKSTEP)]); pW = np.stack([pis_mat[i % S] for i in range(LOCKSTEP)])
            for _ in range(3):
                cb.generate_witness_partial_batch_dev(cells0, vW, pW, dW)
            bst = {}
            for s_ in ["witness_generate"] + STAGES:
                ms, cnt = gpu.profile_read(s_)
                bst[s_] = round(ms / max(cnt, 1), 4)
            gpu.profile(False)
            dW.free(scrub=True)
            cb.close()
            extra["lockstep_batch_stage_ms"] = dict(bst, batch=LOCKSTEP, total=round(sum(bst.values()), 4),
                                                    per_proof=round(sum(bst.values()) / LOCKSTEP, 4))
        # checker (not timed): the oracle verifies the GPU proof and reproduces its bytes
        import oracle_binding
        orc = oracle_binding.Oracle()
        oc = oracle_binding.OracleCircuit(orc, pack)
        ok = ok and oc.verify(proof) == 0
        # the library's own host verifier (include/qpgpu_verify.h) on the same proof, and its throughput over one step's proofs
        pv = pkg.Verifier(pack, circuit=circ)
        tv = time.perf_counter()
        ok = ok and pv.verify(proof)
        one_ms = (time.perf_counter() - tv) * 1e3
        flipped = bytearray(proof); flipped[len(proof) // 2] ^= 1
        ok = ok and not pv.verify(bytes(flipped))
        tv = time.perf_counter()
        ok = ok and all(pv.verify_many([proof] * 64))
        extra["host_verifier"] = {"single_ms": round(one_ms, 3), "proofs_per_s_all_cores": round(64 / (time.perf_counter() - tv), 1),
                                  "note": "qpgpu_verifier_verify: transcript replay, vanishing polynomial at zeta, proof of work, 28 query rounds; verifier data = "
                                          "the circuit handle's constants/sigmas cap; also run by the aggregation tree at every commit and on the root"}
        pv.close()
        cpu_baseline = None
        if not args.no_cpu_baseline and world == 1:   # reported at N=1 only
            # the same timed region on the host cores: commit + generate_partial_witness + prove per proof, ONE PROOF PER THREAD
            # (independent proofs need no barrier per loop; the inner OpenMP regions run on their proof's own thread), on every
            # core this process may use; setup (circuit commitment, partition, generator list) outside, as in the reference's bench
            threads = oracle_binding.usable_cpus(1024)
            orc.set_threads(threads)
            oprov = oracle_binding.OracleProver(orc, pack)
            nb_ = threads
            vals_c = np.stack([vals_all[i % S] for i in range(nb_)]); pis_c = np.stack([pis_mat[i % S] for i in range(nb_)])
            reps = 0
            t1 = time.perf_counter()
            while True:
                cpu_proofs = oprov.commit_prove_many(cells0, vals_c, pis_c)
                reps += 1
                if time.perf_counter() - t1 > 12.0 or reps >= 8:
                    break
            cdt = time.perf_counter() - t1
            ok = ok and cpu_proofs[0] == proof
            # one proof alone with the loops parallelised over the same threads: the latency form
            tl_ = time.perf_counter()
            cpu_one = oc.prove(wires0, pis0)
            lat = time.perf_counter() - tl_
            ok = ok and cpu_one == proof
            oprov.close()
            cpu_baseline = {
                "value": round(reps * nb_ / cdt, 4), "unit": "proofs/s", "cores": threads, "kind": "port",
                "sample": f"{reps} x {nb_} proofs (commit + witness generation + prove, the headline's inputs) with oracle/witness.c + oracle/prove.c, one proof per "
                          f"thread on {threads} threads (every core this process may use; cpus_visible {os.cpu_count()}); the reference's Rayon prover cannot be built here (no Rust toolchain)",
                "single_proof_latency_s_all_threads": round(lat, 3),
            }
        if cpu_baseline is not None:
            # what the reference itself publishes (another machine, the real leaf circuit, witness generation included): the
            # port above is scalar C (no vectorised Poseidon, no packed field arithmetic) and far slower than plonky2's Rayon/AVX prover
            cpu_baseline["reference_published"] = {"leaf_ms": 20, "hw": "Apple M2 Max 12c", "source": "paper/main.tex:449,455"}
            cpu_baseline["port_vs_published"] = ("the port needs %.1f core-seconds per proof against 0.24 for the published reference figure (20 ms on 12 cores, a circuit of unknown size >= 2^12 rows): "
                                                 "scalar C whose Poseidon permutation takes ~4.7 us; read GPU/CPU ratios against the published figure" % (threads / cpu_baseline["value"]))
        extra["cpu_baseline"] = cpu_baseline
        oc.close()
        if not args.no_ntt:
            gbs, roof, ntt_ok, col_in, col_out = ntt_leg(pkg, gpu, 20, 128, 10)
            ok = ok and ntt_ok and bool(np.array_equal(col_out, orc.fft(col_in, 20)))
            extra["roofline"] = roof
            extra["ntt_2p20_fwd_inv_GBps"] = round(gbs, 1)
    pool.shutdown(); pool_gen.shutdown()
    prover_pool.close()
    for c_ in circs:
        c_.close()
    for g_ in gpus:
        g_.close()
    if not ok:
        raise SystemExit("bench.py: correctness check FAILED (oracle verify / byte parity / NTT round trip)")
    if rank == 0:
        line = {
            "metric": "Wormhole proofs/sec", "value": round(value, 3), "unit": "proofs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "ms_per_proof": round(dt / (args.steps * S) * 1e3, 4),
            "engine_clock_mhz": headline_clock,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "window_proofs_per_s": windows, "step_ms_rank0": step_ms, "gather_ms_rank0": [round(x, 2) for x in gather_ms[-16:]],
            "gather_ms_mean_per_rank": gather_ms_per_rank,
            "proof_exchange": "none (one rank)" if world == 1 else ("all_gather of every step's proof bytes" if args.all_gather else "gather of every step's proof bytes to rank 0, the consuming rank"),
            "host_threads_per_rank": {"library_proving_workers": WORKERS, "python_threads": __import__("threading").active_count(), "cpus_visible": os.cpu_count()},
            "config": {"workload": "BASELINE configs[2]/[3]: `prover.commit(&inputs)?.prove()` (the reference bench's timed region, wormhole/prover/benches/prover.rs:38) "
                                   "on the Wormhole leaf circuit restated natively (csrc/leaf_circuit.cpp), from CircuitInputs in host memory to proof bytes in host memory: "
                                   "commit (fill_witness, host thread) -> witness generation (s1, device) -> LDE + Poseidon Merkle commit + quotient + FRI (s2..s12, device); "
                                   "proofs_per_step_per_gpu different inputs per GPU per step (input 0 = the reference bench's build_dummy_circuit_inputs, the others real spends with "
                                   "Merkle paths of 1..16 levels)",
                       "circuit": {"degree_bits": d, "gate_rows_before_padding": leaf.info["rows_before_padding"], "rows": {k[5:]: v for k, v in leaf.info.items() if k.startswith("rows_")},
                                   "caveats": "the circuit proves the reference's statement over the reference's gate set, built by this repository's restatement of plonky2's builder: "
                                              "row order, the Poseidon2 gate's wire layout (pack table, LAYOUT UNPINNED) and the circuit digest cannot be matched to the fork offline; padded with "
                                              "NoopGate rows from 2^%d to 2^%d because the reference states its circuits are >= 2^12 rows (common/src/circuit.rs:464-467) and profiles its leaf at degree 13 (wormhole/circuit/src/profile.rs:121)" % (max(5, (leaf.info["rows_before_padding"] - 1).bit_length()), d)},
                       "degree_bits": d, "gates": "PublicInput, Constant, BaseSum<2>(63 limbs), Arithmetic(20 ops), Poseidon(123 constraints; public-input hash), Poseidon2 gate (123 constraints; the leaf's 61 application-hash permutations), Noop; 2 selector groups", "num_wires": 135, "num_routed_wires": 80, "rate_bits": 3, "cap_height": 4,
                       "num_query_rounds": 28, "proof_of_work_bits": 16, "fri_arity_bits": 4, "proof_bytes": proof_len, "proofs_in_flight_per_gpu": S, "proofs_per_step_per_gpu": S, "workers": WORKERS, "lockstep_batch": LOCKSTEP,
                       "multi_gpu": ("independent proofs per rank + one %s of each step's proof bytes (%s)" % ("all_gather" if args.all_gather else "gather to rank 0", "RCCL" if backend == "nccl" else backend + " rehearsal, ranks share the visible GPUs")) if world > 1 else "single GPU"},
        }
        line.update(extra)
        if "roofline" not in line:
            line["roofline"] = None
        if "cpu_baseline" not in line:
            line["cpu_baseline"] = None
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
