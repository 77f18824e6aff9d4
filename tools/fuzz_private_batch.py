"""Fuzz of the private-batch circuit's own logic on the device against its host restatement: random slot contents drawn from a small
vocabulary (two blocks and the dummy's zero block, two assets, two fee rates, three exit accounts, a few nullifiers, amounts up to
2^31) so that grouping, duplicates, dummies with attacker-chosen exits, asset / fee / block mismatches, replayed leaves and exit-sum
overflows all occur. The leaves are proofs of the restated build_fake_leaf_circuit (any 21 public inputs), made on the device; the
private-batch circuit (complete in-circuit verification + build_private_batch_constraints, N = 4) generates its witness on the device
WITHOUT public inputs. Per batch: qpgpu_private_batch_outputs accepts <=> stage s1 finds a witness, and then the public inputs read
out of the witness equal the host's. usage: python tools/fuzz_private_batch.py [batches] [seed]"""
import json, sys, time
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import __graft_entry__ as ge
pkg = ge.load_package()
gpu = pkg.QpGpu(0)
L, R, A = pkg.leaf, pkg.recursion, pkg.aggregation
count = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
N, B = 4, 8
fake = L.LeafCircuit(fragment=L.FRAGMENT_FAKE_LEAF)
fc = pkg.Circuit(gpu, fake.pack)
fv = pkg.Verifier(fake.pack, circuit=fc)
nwf = 135 << fake.info["degree_bits"]
fd = gpu.alloc(nwf * 8)
w = R.WrapperCircuit(fake.pack, fv, N, logic="private_batch", verify=True)
wc = pkg.Circuit(gpu, w.pack, max_batch=B)
words = 135 << w.info["degree_bits"]
d = gpu.alloc(B * words * 8)
none = np.zeros(0, dtype=np.uint64)
BLOCKS = [(0, 0, 0, 0), (0xB10C0001, 2, 3, 4), (0xB10C0002, 2, 3, 4)]
EXITS = [(0, 0, 0, 0), (0x1111, 1, 2, 3), (0x2222, 1, 2, 3), (0x3333, 1, 2, 0xFFFFFFFF00000000)]
cache = {}


def leaf_proof(row):
    key = row.tobytes()
    if key not in cache:
        fc.generate_witness_partial_dev(none, none, row, fd)
        cache[key] = fc.prove_dev(fd, row)
    return cache[key]


def random_row():
    p = np.zeros(21, dtype=np.uint64)
    blk = BLOCKS[0 if rng.integers(0, 4) == 0 else (2 if rng.integers(0, 24) == 0 else 1)]
    dummy = blk == BLOCKS[0]
    p[0] = 1 if rng.integers(0, 40) == 0 else 0
    big = rng.integers(0, 12) == 0
    p[1] = int(rng.integers(0, 1 << 31)) if big else int(rng.integers(0, 1000))
    p[2] = int(rng.integers(0, 1 << 31)) if big else int(rng.integers(0, 1000))
    if dummy and rng.integers(0, 3):
        p[1] = p[2] = 0
    p[3] = 11 if rng.integers(0, 30) == 0 else 10
    p[4:8] = (int(rng.integers(1, 40)), int(rng.choice([0, 0xFFFFFFFF00000000])), 8, 9)
    if dummy and rng.integers(0, 2):
        p[4:8] = 0
    p[8:12] = EXITS[int(rng.integers(0, 4))]; p[12:16] = EXITS[int(rng.integers(0, 4))]
    p[16:20] = blk
    p[20] = 0 if dummy else 42
    return p


stats = {"batches": count, "satisfiable": 0, "unsatisfiable": 0, "mismatches": 0, "reasons": {}}
t0 = time.time()
pending = []


def flush():
    if not pending:
        return
    st = R.generate_wrapper_witnesses(wc, w, [p[2] for p in pending], d)
    got = wc.witness_public_inputs_dev(d, len(pending))
    for k, ((rows, want, _), s) in enumerate(zip(pending, st)):
        if (s == 0) != (want is not None):
            stats["mismatches"] += 1
            print("MISMATCH: host", "accepts" if want is not None else "refuses", "device status", s, rows.tolist())
        elif s == 0:
            stats["satisfiable"] += 1
            if got[k].tolist() != want.tolist():
                stats["mismatches"] += 1
                print("PUBLIC INPUTS DIFFER", rows.tolist())
        else:
            stats["unsatisfiable"] += 1
    pending.clear()


for b in range(count):
    rows = np.stack([random_row() for _ in range(N)])
    pre = rng.integers(0, 1 << 63, (N, 4), dtype=np.uint64)
    try:
        want = A.private_batch_outputs(rows, pre)
    except pkg.QpGpuError as e:
        want = None
        key = str(e).split(":")[-1].strip()[:40]
        stats["reasons"][key] = stats["reasons"].get(key, 0) + 1
    c = w.commit([leaf_proof(r) for r in rows], preimages=pre, derive_public_inputs=True)
    pending.append((rows, want, c))
    if len(pending) == B:
        flush()
flush()
stats["distinct_leaf_proofs"] = len(cache)
stats["seconds"] = round(time.time() - t0, 1)
print(json.dumps(stats))
sys.exit(1 if stats["mismatches"] else 0)
