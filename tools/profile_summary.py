"""Summarise a rocprofv3 --kernel-trace --stats output directory: copies the kernel stats table and extracts the
per-launch durations of the 2^20 x 128 NTT launches (Grid_Size_Y == 128) the bench's roofline object refers to.
usage: python tools/profile_summary.py <rocprof_dir> <out_prefix> [profiled command]"""
import csv, glob, json, os, shutil, sys

src, prefix = sys.argv[1], sys.argv[2]
stats = sorted(glob.glob(os.path.join(src, "**", "*kernel_stats.csv"), recursive=True))
trace = sorted(glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True))
if stats:
    shutil.copy(stats[-1], prefix + "_kernel_stats.csv")
launches = {}
for path in trace:
    for r in csv.DictReader(open(path)):
        name = r.get("Kernel_Name", "")
        if "_kernel<5, 5" not in name or "_lde_" in name:      # the LDE variants belong to the proof legs, not to the 2^20 transform
            continue
        gy = int(r.get("Grid_Size_Y", r.get("Grid_Size_y", "0")) or 0)
        wy = int(r.get("Workgroup_Size_Y", "1") or 1)
        if gy // max(wy, 1) != 128 and gy != 128:
            continue
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        gx = int(r.get("Grid_Size_X", r.get("Grid_Size_x", "0")) or 0)
        launches.setdefault(name.split("(anonymous namespace)::")[-1], []).append((gx, dur))
# other legs launch the same kernels on 128 columns of shorter polynomials (a 2^19-point LDE): keep the 2^20-point launches,
# the ones with the widest grid
for k, v in launches.items():
    top = max(g for g, _ in v)
    launches[k] = [d for g, d in v if g == top]
cmd = sys.argv[3] if len(sys.argv) > 3 else "python3 bench.py (default command)"
out = {"source": f"rocprofv3 --kernel-trace --stats -- {cmd}, MI355X; 2^20 x 128 launches selected by grid Y == 128",
       "launches": [{"kernel": k, "calls": len(v), "avg_ms": round(sum(v) / len(v), 4), "min_ms": round(min(v), 4)} for k, v in sorted(launches.items())]}
json.dump(out, open(prefix + "_ntt_2p20_launches.json", "w"), indent=1)
print(json.dumps(out["launches"]))
