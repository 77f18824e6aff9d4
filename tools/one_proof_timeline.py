"""Timeline of one proof from a rocprofv3 kernel trace: per kernel start offset, duration and the idle gap before it.
Run:  rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python3 tools/one_proof_timeline.py run
then: python3 tools/one_proof_timeline.py show DIR/t_kernel_trace.csv"""
import sys
if sys.argv[1] == "run":
    sys.path.insert(0, ".")
    import __graft_entry__ as g
    pkg = g.load_package()
    gpu = pkg.QpGpu(0)
    pack, wires, pis = pkg.synth_circuit(13, seed=1, poseidon=True, base_sum=True)
    circ = pkg.Circuit(gpu, pack)
    d = gpu.to_device(wires)
    for _ in range(3):
        circ.prove_dev(d, pis)
    import time; time.sleep(0.05)
    circ.prove_dev(d, pis)            # the traced one: last in the file
    circ.close()
else:
    import csv
    rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(anonymous namespace)::")[-1].split("(")[0][:36])
                   for r in csv.DictReader(open(sys.argv[2]))))
    # last proof = kernels after the largest idle gap
    gaps = [(rows[i][0] - rows[i - 1][1], i) for i in range(1, len(rows))]
    start = max(gaps)[1]
    rows = rows[start:]
    t0 = rows[0][0]
    busy = sum(e - s for s, e, _ in rows)
    span = rows[-1][1] - t0
    print(f"kernels {len(rows)}, span {span/1e6:.3f} ms, kernel time {busy/1e6:.3f} ms, idle {1-busy/span:.2%}")
    agg = {}
    prev = t0
    gap_by = {}
    for s, e, n in rows:
        a = agg.setdefault(n, [0, 0, 0]); a[0] += 1; a[1] += e - s; a[2] += max(0, s - prev)
        prev = e
    for n, (c, t, g) in sorted(agg.items(), key=lambda kv: -kv[1][1] - kv[1][2]):
        print(f"{n:38s} x{c:3d}  kernel {t/1e3:8.1f} us   gap-before {g/1e3:8.1f} us")
