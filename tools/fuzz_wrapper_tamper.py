"""Fuzz of the in-circuit verifier on the device against the host verifier: random single-bit flips anywhere in a leaf proof.
For every tampered proof the host verifier's verdict (qpgpu_verifier_verify) and the wrapper circuit's (complete in-circuit
verification, QPGPU_WRAPPER_VERIFY: does stage s1 on the device find a witness?) must be the same; a flip the proof-target filling
refuses outright (a non-canonical element) counts as rejected by both. usage: python tools/fuzz_wrapper_tamper.py [count] [seed]"""
import json, sys, time
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import __graft_entry__ as ge
import leaf_cases as lc
pkg = ge.load_package()
gpu = pkg.QpGpu(0)
L, R = pkg.leaf, pkg.recursion
count = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
leaf = L.LeafCircuit()
lp = L.LeafProver(pkg, gpu, leaf)
good = [lp.prove(lc.real_inputs(L, depth=3, seed=5))[0], lp.prove(lc.dummy_inputs(L))[0]]
ver = pkg.Verifier(leaf.pack, circuit=lp.circ)
B = 8
w = R.WrapperCircuit(leaf.pack, ver, 2, verify=True)
wc = pkg.Circuit(gpu, w.pack, max_batch=B)
d = gpu.alloc(B * (135 << w.info["degree_bits"]) * 8)
stats = {"flips": count, "rejected_by_both": 0, "accepted_by_both": 0, "refused_at_fill": 0, "mismatches": 0, "by_region": {}}
n = len(good[0])
h = pkg.pack_header(leaf.pack)
t0 = time.time()
pending = []


def flush():
    if not pending:
        return
    coms = [p[2] for p in pending]
    st = R.generate_wrapper_witnesses(wc, w, coms, d)
    for (off, host_ok, _), s in zip(pending, st):
        dev_ok = s == 0
        if dev_ok != host_ok:
            stats["mismatches"] += 1
            print("MISMATCH offset", off, "host", host_ok, "wrapper", dev_ok)
        stats["accepted_by_both" if host_ok and dev_ok else "rejected_by_both"] += 1
    pending.clear()


for k in range(count):
    off, bit = int(rng.integers(0, n)), int(rng.integers(0, 8))
    bad = bytearray(good[0]); bad[off] ^= 1 << bit
    bad = bytes(bad)
    region = "public inputs" if off >= n - 8 * 21 else "caps + openings" if off < 3 * 16 * 32 + 16 * (h["num_selectors"] + h["num_constants"] + 80 + 135 + 4 + 2 * h["num_partial_products"] + 16) else "FRI"
    stats["by_region"][region] = stats["by_region"].get(region, 0) + 1
    host_ok = bool(ver.verify(bad))
    try:
        c = w.commit([bad, good[1]])
    except ValueError:
        stats["refused_at_fill"] += 1
        if host_ok:
            stats["mismatches"] += 1; print("MISMATCH offset", off, ": refused at fill, accepted by the host verifier")
        continue
    pending.append((off, host_ok, c))
    if len(pending) == B:
        flush()
flush()
stats["seconds"] = round(time.time() - t0, 1)
print(json.dumps(stats))
sys.exit(1 if stats["mismatches"] else 0)
