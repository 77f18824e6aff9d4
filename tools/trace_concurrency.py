"""From a rocprofv3 kernel trace: how busy the GPU is over the proving loop — union coverage of kernel intervals, average
number of kernels in flight, and the idle gaps. usage: python tools/trace_concurrency.py <kernel_trace.csv>"""
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(anonymous namespace)::")[-1][:40]))
rows.sort()
# the densest window: take the middle 60 % of the trace by time
t0, t1 = rows[0][0], max(r[1] for r in rows)
lo, hi = t0 + (t1 - t0) * 0.2, t0 + (t1 - t0) * 0.8
ev = []
for s, e, _ in rows:
    s2, e2 = max(s, lo), min(e, hi)
    if e2 > s2:
        ev.append((s2, 1)); ev.append((e2, -1))
ev.sort()
cur, last, busy, area = 0, lo, 0.0, 0.0
hist = {}
for t, d in ev:
    if cur > 0:
        busy += t - last
    area += cur * (t - last)
    hist[cur] = hist.get(cur, 0) + (t - last)
    last = t; cur += d
span = hi - lo
print(f"window {span/1e6:.1f} ms: busy {busy/span:.3f}, mean kernels in flight {area/span:.2f}")
for k in sorted(hist):
    print(f"  {k} in flight: {hist[k]/span:.3f}")
