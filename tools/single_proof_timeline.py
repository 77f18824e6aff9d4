#!/usr/bin/env python3
"""One leaf proof at a time on an otherwise idle device (commit on the host, stage s1, stages s2..s12, proof bytes back), 20 in a row:
the wall time per proof, and — when run under `rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/single_proof_timeline.py`
and followed by `python3 tools/single_proof_timeline.py --parse DIR` — how much of that the device spent inside kernels, how many
launches a proof takes and where the idle gaps are."""
import csv
import glob
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def run():
    import __graft_entry__ as g
    pkg = g.load_package()
    import leaf_cases as lc
    L = pkg.leaf
    gpu = pkg.QpGpu(0)
    leaf = L.LeafCircuit(min_degree_bits=13)                  # the bench's shape (the reference's leaf has 2^13 rows)
    p = L.LeafProver(pkg, gpu, leaf, hash_hints="--hints" in sys.argv)
    xs = [lc.real_inputs(L, depth=3 + i % 5, seed=i) for i in range(8)]
    for x in xs[:3]:
        p.prove(x)
    gpu.sync()
    n = 20
    t = time.perf_counter()
    for i in range(n):
        p.prove(xs[i % 8])
    dt = (time.perf_counter() - t) / n
    print("single proof, commit%s + s1 + prove: %.3f ms" % (" (with hash hints)" if "--hints" in sys.argv else "", dt * 1e3))
    p.close()


def parse(d):
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))), key=lambda r: r[0])
    # the last 20 proofs: split at the copy pass that ends stage s1 (one per proof)
    starts = [i for i, r in enumerate(rows) if "witness_fill_kernel" in r[2]]
    starts = starts[-20:]
    tot_busy = tot_wall = tot_launch = 0
    gaps = {}
    for a, b in zip(starts[:-1], starts[1:]):
        seg = rows[a:b]
        wall = seg[-1][1] - seg[0][0]
        busy = 0; end = seg[0][0]
        for s, e, nm in seg:
            if e > end:
                busy += e - max(s, end); end = e
        tot_busy += busy; tot_wall += rows[b][0] - seg[0][0]; tot_launch += len(seg)
        for (s0, e0, n0), (s1, e1, n1) in zip(seg[:-1], seg[1:]):
            g = s1 - e0
            if g > 15000:
                k = n0.split("(")[0][-40:] + " -> " + n1.split("(")[0][-40:]
                c, tt = gaps.get(k, (0, 0)); gaps[k] = (c + 1, tt + g)
    m = len(starts) - 1
    print("per proof: %.3f ms between proof starts, %.3f ms inside kernels (%.0f %%), %d launches" % (tot_wall / m / 1e6, tot_busy / m / 1e6, 100.0 * tot_busy / tot_wall, tot_launch // m))
    for k, (c, tt) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:14]:
        print("  gap %7.1f us x %4.1f per proof   %s" % (tt / c / 1e3, c / m, k))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--parse":
        parse(sys.argv[2])
    else:
        run()
