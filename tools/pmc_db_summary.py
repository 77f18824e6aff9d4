#!/usr/bin/env python3
"""Per-kernel counter table from rocprofv3 --pmc result databases (rocpd sqlite, `counters_collection` view).
usage: pmc_db_summary.py <kernel-name substring> <results.db> [<results.db> ...]   -> JSON on stdout"""
import json, sqlite3, sys
from collections import defaultdict
sub = sys.argv[1]
out = {}
for path in sys.argv[2:]:
    db = sqlite3.connect(path)
    rows = db.execute("select dispatch_id, kernel_name, counter_name, value, duration, grid_size, workgroup_size, vgpr_count, lds_block_size from counters_collection").fetchall()
    per = defaultdict(lambda: defaultdict(list))
    meta = {}
    for did, kname, cname, val, dur, grid, wg, vgpr, lds in rows:
        if sub not in kname:
            continue
        key = kname.split("(anonymous namespace)::")[-1].split("(")[0]
        per[key][cname].append(val)
        meta[key] = {"grid": grid, "workgroup": wg, "vgpr": vgpr, "lds": lds}
        per[key]["_duration_ns"].append(dur)
    for k, cs in per.items():
        e = out.setdefault(k, dict(meta[k]))
        for c, v in cs.items():
            n = len(cs["_duration_ns"]) // max(1, len([x for x in cs if not x.startswith("_")]))
            e[c if c != "_duration_ns" else "duration_ns_under_pmc"] = round(sum(v) / len(v), 1)
            e["launches"] = len(v)
print(json.dumps(out, indent=1))
