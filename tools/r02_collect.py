#!/usr/bin/env python3
"""Turns the output of tools/r02_final.sh (gpurun_out/r02_final) into the tracked files under profiles/:
  r02_final_bench.json                  the default `python bench.py` line
  r02_final_bench_2rank_gloo.json       `bench.py --gpus 2` self-launched (two ranks on the box's one GPU, gloo rehearsal)
  r02_final_bench_kernel_stats.csv / r02_final_bench_ntt_2p20_launches.json    rocprofv3 --kernel-trace --stats of the default
                                        command and the 2^20 x 128 NTT launches inside it
  r02_final_ntt_only_kernel_stats.csv / r02_final_ntt_only_ntt_2p20_launches.json   the same for tools/ntt_only.py
  r02_final_headline_kernel_stats.csv / r02_final_single_worker_kernel_stats.csv   the headline leg alone with six workers
                                        overlapping, and with one worker (every kernel alone on the GPU: isolated stage costs)
  r02_final_ntt_pmc_summary.json        HBM bytes per transform (FETCH_SIZE / WRITE_SIZE passes)
  r02_final_ntt_valu_summary.json       VALU instructions per element and issue-slot share (SQ passes)
  r02_final_ntt_isa_hist.json           static instruction mix of the compiled NTT kernels (tools/isa_hist.py)
usage: python tools/r02_collect.py [src_dir]"""
import glob, json, os, shutil, sqlite3, subprocess, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r02_final")
DST = os.path.join(ROOT, "profiles")
PY = sys.executable
N_ELEMS = (1 << 20) * 128
SIMDS = 256 * 4
XCDS = 8


def last_json_line(path):
    for line in reversed(open(path).read().strip().split("\n")):
        if line.startswith("{"):
            return json.loads(line)
    raise SystemExit(f"no JSON line in {path}")


for name, out in (("bench.json", "r02_final_bench.json"), ("bench_2rank_gloo.json", "r02_final_bench_2rank_gloo.json")):
    p = os.path.join(SRC, name)
    if os.path.exists(p) and os.path.getsize(p):
        json.dump(last_json_line(p), open(os.path.join(DST, out), "w"), indent=1)
        print("wrote", out)

# the box summarised its kernel traces (tools/r02_final.sh -> tools/profile_summary.py): sum_bench = `python3 bench.py` (the default
# command: kernel stats of the whole run and the 2^20 x 128 NTT launches inside it, what roofline.avg_ms must agree with),
# sum_ntt_only = tools/ntt_only.py, sum_headline = the headline leg alone (kernel shares of a proof)
for prefix, out in (("sum_bench", "r02_final_bench"), ("sum_ntt_only", "r02_final_ntt_only"), ("sum_headline", "r02_final_headline"), ("sum_single_worker", "r02_final_single_worker")):
    for suffix in ("_kernel_stats.csv", "_ntt_2p20_launches.json"):
        src = os.path.join(SRC, prefix + suffix)
        if os.path.exists(src) and os.path.getsize(src) > 2 and not (prefix in ("sum_headline", "sum_single_worker") and suffix.endswith(".json")):
            shutil.copy(src, os.path.join(DST, out + suffix))
            print("copied", out + suffix)

if os.path.isdir(os.path.join(SRC, "pmc_fetch")) and os.path.isdir(os.path.join(SRC, "pmc_write")):
    subprocess.check_call([PY, os.path.join(ROOT, "tools", "pmc_summary.py"), os.path.join(SRC, "pmc_fetch"), os.path.join(SRC, "pmc_write"),
                           os.path.join(DST, "r02_final_ntt_pmc_summary.json")])

# SQ counters: rocpd databases, one per pass
per = defaultdict(lambda: defaultdict(list))
meta = {}
for db_path in sorted(glob.glob(os.path.join(SRC, "pmc_sq*", "**", "*results.db"), recursive=True)):
    db = sqlite3.connect(db_path)
    for kname, cname, val, dur, vgpr, lds in db.execute(
            "select kernel_name, counter_name, value, duration, vgpr_count, lds_block_size from counters_collection"):
        if "_kernel<5, 5" not in kname:
            continue
        key = kname.split("(anonymous namespace)::")[-1].split("(")[0]
        per[key][cname].append(val)
        per[key]["_dur_" + os.path.basename(os.path.dirname(db_path))].append(dur)
        meta[key] = {"vgpr": vgpr, "lds_bytes": lds}
if per:
    kernels = {}
    for key, cs in sorted(per.items()):
        e = dict(meta[key])
        for c, v in cs.items():
            if c.startswith("_dur_"):
                e["duration_us_under_" + c[5:]] = round(sum(v) / len(v) / 1e3, 1)
            else:
                e[c] = round(sum(v) / len(v), 1)
        inv, rows = [s.strip() == "true" for s in key.split("<")[1].rstrip(">").split(",")[2:4]]
        e["pass"] = ("inverse " if inv else "forward ") + ("rows" if rows else "strided")
        wave_insts = e.get("SQ_INSTS_VALU")
        if wave_insts:
            e["valu_insts_per_element"] = round(wave_insts * 64 / N_ELEMS, 1)
            # GRBM_GUI_ACTIVE comes from the second pass; the kernel's duration there at the clock it implies
            cyc = e.get("GRBM_GUI_ACTIVE")
            if cyc:
                e["gpu_cycles"] = round(cyc / XCDS)     # the counter is summed over the 8 XCDs
                e["valu_issue_frac"] = round(wave_insts * 4 / (SIMDS * cyc / XCDS), 3)
        kernels[key] = e
    fwd = [e for e in kernels.values() if e["pass"].startswith("forward")]
    out = {"source": "rocprofv3 --kernel-trace --pmc <SQ set 1 | SQ set 2 + GRBM_GUI_ACTIVE> -- python3 tools/ntt_only.py 3 (tools/r02_final.sh), MI355X",
           "workload": "2^20 points x 128 columns", "elements_per_transform": N_ELEMS,
           "definitions": {"valu_insts_per_element": "SQ_INSTS_VALU (wave instructions) x 64 lanes / (2^20 x 128 elements), summed over the strided and the rows launch of one forward transform",
                           "valu_issue_frac": "SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 cycles of the launch; the counter is summed over the 8 XCDs): the share of the chip's VALU issue slots (one wave64 instruction per SIMD per 4 cycles) the launch used; per transform = the launches weighted by their cycles"},
           "kernels": kernels}
    if len(fwd) == 2 and all("valu_insts_per_element" in e for e in fwd):
        out["valu_insts_per_element"] = round(sum(e["valu_insts_per_element"] for e in fwd), 1)
        if all("GRBM_GUI_ACTIVE" in e for e in fwd):
            out["valu_issue_frac"] = round(sum(e["SQ_INSTS_VALU"] for e in fwd) * 4 / (SIMDS * sum(e["GRBM_GUI_ACTIVE"] for e in fwd) / XCDS), 3)
    json.dump(out, open(os.path.join(DST, "r02_final_ntt_valu_summary.json"), "w"), indent=1)
    print("wrote r02_final_ntt_valu_summary.json", out.get("valu_insts_per_element"), out.get("valu_issue_frac"))

hist = subprocess.run([PY, os.path.join(ROOT, "tools", "isa_hist.py"), os.path.join(ROOT, "qp-zk-circuits_amd", "csrc", "ntt_inst_3.hip"), "--json"],
                      capture_output=True, text=True)
if hist.returncode == 0 and hist.stdout.strip():
    open(os.path.join(DST, "r02_final_ntt_isa_hist.json"), "w").write(hist.stdout)
    print("wrote r02_final_ntt_isa_hist.json")
else:
    print("isa_hist failed:", hist.stderr[-300:])
