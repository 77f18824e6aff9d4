"""HBM traffic of the 2^20 x 128 NTT launches from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE in separate runs, as
MI355X_MICROARCH.md prescribes). FETCH_SIZE is doubled for the fully coalesced rows pass (gfx950 tallies 128-byte
requests at 64 B). The strided pass is calibrated on its known byte count, as the guide asks for access patterns it does not
list: a pass reads every input element exactly once (2^20 x 128 x 8 B = 1 GiB), so a reported value well below that can only
be the same half-tally (16-lane tiles read whole 128-byte segments) and is doubled; a value at or above it (8-lane tiles,
64-byte segments, round 1) is taken as reported.
usage: python tools/pmc_summary.py <fetch_dir> <write_dir> <out.json>"""
import csv, glob, json, os, sys

def load(d, counter):
    out = []
    for path in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(path)):
            if r.get("Counter_Name") != counter or "_kernel<5, 5" not in r.get("Kernel_Name", ""):
                continue
            out.append((int(r.get("Dispatch_Id", 0)), r["Kernel_Name"].split("(anonymous namespace)::")[-1], float(r["Counter_Value"])))
    return sorted(out)

fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
assert len(fetch) == len(write) and fetch, (len(fetch), len(write))
launches, total = [], 0.0
for (_, name, f), (_, name2, w) in zip(fetch, write):
    assert name == name2
    rows = ", true>(" in name.replace("true, true", "x, true").replace("false, true", "x, true")
    must_read = 8.0 * (1 << 20) * 128          # bytes every pass has to read at least
    corr = 2.0 if rows or f * 1024 < 0.75 * must_read else 1.0
    hbm = (f * corr + w) * 1024
    launches.append({"kind": "rows" if rows else "strided", "kernel": name[:48], "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w,
                     "fetch_correction": corr, "hbm_bytes": int(hbm)})
    total += hbm
n_transforms = len(launches) // 2
algo = 16 * (1 << 20) * 128
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 tools/ntt_only.py, MI355X",
       "workload": "2^20 points x 128 columns, forward + inverse", "units": "KB as reported by rocprofv3", "launches": launches,
       "hbm_bytes_per_transform": int(total / n_transforms), "algorithmic_bytes_per_transform": algo,
       "traffic_over_algorithmic": round(total / n_transforms / algo, 3)}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(out["hbm_bytes_per_transform"], out["traffic_over_algorithmic"], len(launches))
