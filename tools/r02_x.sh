#!/bin/bash
set -o pipefail
O=gpurun_out/r02_x
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $R/$O/prof -o w -- python3 $R/tools/witness_profile.py > $R/$O/log.txt 2>&1; echo "rc=$?" | tee -a $R/$O/summary.txt
cd $R
python - <<'PY' | tee -a $O/summary.txt
import csv, glob, collections
rows=[]
for p in glob.glob("gpurun_out/r02_x/prof/**/*kernel_trace.csv", recursive=True):
    rows+=list(csv.DictReader(open(p)))
w=[r for r in rows if "witness_" in r["Kernel_Name"]]
w.sort(key=lambda r:int(r["Start_Timestamp"]))
half=w[len(w)//2:]
by=collections.defaultdict(list)
prev_end=None; gaps=[]
for r in half:
    name=r["Kernel_Name"].split("witness_")[1].split("(")[0]
    dur=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
    by[name].append((dur,int(r["Grid_Size_X"])))
    if prev_end is not None: gaps.append((int(r["Start_Timestamp"])-prev_end)/1e3)
    prev_end=int(r["End_Timestamp"])
for k,v in by.items():
    d=sorted(x for x,_ in v)
    print(k, "calls",len(v),"sum_ms",round(sum(d)/1e3,2),"median_us",d[len(d)//2],"p90",d[int(len(d)*0.9)],"max",d[-1])
print("gaps: median_us", sorted(gaps)[len(gaps)//2], "sum_ms", round(sum(gaps)/1e3,2))
# duration vs grid size for combined
c=by.get("combined_kernel",[])
import statistics
buckets=collections.defaultdict(list)
for dur,g in c: buckets[min(g//256,8)].append(dur)
for b in sorted(buckets): print("grid blocks ~",b, "n",len(buckets[b]), "median", statistics.median(buckets[b]))
PY
