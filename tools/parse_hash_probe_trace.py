"""Per-launch durations of the hashing kernels in a `rocprofv3 --kernel-trace --output-format csv -d gpurun_out/hp -- python3
tools/hash_probe.py` trace: the last tree build, level by level (kernel, microseconds, grid size)."""
import csv, glob, sys
f = glob.glob('gpurun_out/hp/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
out = []
for r in rows:
    nm = r['Kernel_Name']
    if 'node' in nm or 'leaf_hash' in nm or 'tree_top' in nm:
        out.append((nm[:60], (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3, r.get('Grid_Size') or r.get('Grid_Size_X')))
for o in out[-24:]:
    print(o)
