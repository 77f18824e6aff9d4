"""Fuzz of the public-batch circuit's own logic on the device against its host restatement: M = 3 inner "private batch" proofs of
N = 2 leaves each, their public-input rows (exit-slot count, asset, fee, block hash, block number, 2N (sum, exit account) slots, N
nullifiers, padding) drawn from a small vocabulary so that dummy inners (zero block hash) with attacker-chosen contents, asset / fee
/ block mismatches, a reference taken from the first REAL inner, all-dummy batches and non-zero padding all occur. The inner proofs are
proofs of a stand-in circuit with 21 N + 8 unconstrained public inputs (qpgpu_builder_gadget_circuit(3000 + K)), made on the device;
the public-batch circuit (complete in-circuit verification of the three + build_public_batch_constraints) generates its witness on
the device WITHOUT public inputs. Per batch: qpgpu_public_batch_outputs accepts <=> stage s1 finds a witness, and then the public
inputs read out of the witness equal the host's. usage: python tools/fuzz_public_batch.py [batches] [seed]"""
import ctypes, json, sys, time
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import __graft_entry__ as ge
pkg = ge.load_package()
gpu = pkg.QpGpu(0)
R, A = pkg.recursion, pkg.aggregation
count = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
N, M, B = 2, 3, 8
K = A.private_batch_pi_len(N)

lib = pkg.load_library(); c = ctypes
lib.qpgpu_builder_gadget_circuit.restype = c.c_int
lib.qpgpu_builder_gadget_circuit.argtypes = [c.c_uint, c.c_void_p, c.c_size_t, c.POINTER(c.c_size_t), c.c_void_p, c.c_size_t, c.POINTER(c.c_size_t), c.POINTER(c.c_size_t), c.c_char_p]
n, ni, no = c.c_size_t(), c.c_size_t(), c.c_size_t(); err = c.create_string_buffer(400)
assert lib.qpgpu_builder_gadget_circuit(3000 + K, None, 0, c.byref(n), None, 0, c.byref(ni), c.byref(no), err) == 0, err.value
fake_pack = np.empty(n.value, dtype=np.uint64); cells = np.empty(ni.value + no.value, dtype=np.uint64)
assert lib.qpgpu_builder_gadget_circuit(3000 + K, fake_pack.ctypes.data, fake_pack.size, c.byref(n), cells.ctypes.data, cells.size, c.byref(ni), c.byref(no), err) == 0
fc = pkg.Circuit(gpu, fake_pack)
fv = pkg.Verifier(fake_pack, circuit=fc)
fd = gpu.alloc((135 << int(fake_pack[1])) * 8)
w = R.WrapperCircuit(fake_pack, fv, M, logic="public_batch", verify=True)
wc = pkg.Circuit(gpu, w.pack, max_batch=B)
words = 135 << w.info["degree_bits"]
d = gpu.alloc(B * words * 8)
none = np.zeros(0, dtype=np.uint64)
BLOCKS = [(0, 0, 0, 0), (0xB10C0001, 2, 3, 4), (0xB10C0002, 2, 3, 4), (0, 0, 0, 1)]
EXITS = [(0, 0, 0, 0), (0x1111, 1, 2, 3), (0x2222, 1, 2, 3), (0x3333, 1, 2, 0xFFFFFFFF00000000)]
cache = {}


def inner_proof(row):
    key = row.tobytes()
    if key not in cache:
        fc.generate_witness_partial_dev(none, none, row, fd)
        cache[key] = fc.prove_dev(fd, row)
    return cache[key]


def random_row():
    p = np.zeros(K, dtype=np.uint64)
    r = rng.integers(0, 24)
    blk = BLOCKS[0] if r < 7 else (BLOCKS[2] if r == 7 else (BLOCKS[3] if r == 8 else BLOCKS[1]))
    dummy = blk == BLOCKS[0]
    p[0] = 2 * N if rng.integers(0, 10) else int(rng.integers(0, 9))        # [num_exit_slots, asset_id, volume_fee_bps, block_hash(4), block_number, ...]
    p[1] = 1 if rng.integers(0, 30) == 0 else 0
    p[2] = 11 if rng.integers(0, 30) == 0 else 10
    p[3:7] = blk
    p[7] = 0 if dummy and rng.integers(0, 2) else 42
    if not (dummy and rng.integers(0, 2)):                                  # a dummy inner is all zero half of the time, attacker-filled otherwise
        for s in range(2 * N):
            p[8 + 5 * s] = int(rng.integers(0, 1 << 31)) if rng.integers(0, 8) == 0 else int(rng.integers(0, 1000))
            p[9 + 5 * s:13 + 5 * s] = EXITS[int(rng.integers(0, 4))]
        for k in range(N):
            p[8 + 10 * N + 4 * k:12 + 10 * N + 4 * k] = (int(rng.integers(1, 40)), int(rng.choice([0, 0xFFFFFFFF00000000])), 8, 9)
    if rng.integers(0, 16) == 0:
        p[K - 1] = 7                                                         # the padding is not the circuit's business
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
    rows = np.stack([random_row() for _ in range(M)])
    ab = rng.integers(0, 256, 32, dtype=np.uint8); ab[7::8] &= 0x7F           # four canonical limbs
    addr = ab.tobytes()
    try:
        want = A.public_batch_outputs(rows, N, addr)
    except pkg.QpGpuError as e:
        want = None
        key = str(e).split(":")[-1].strip()[:40]
        stats["reasons"][key] = stats["reasons"].get(key, 0) + 1
    cm = w.commit([inner_proof(r) for r in rows], aggregator_address=addr, derive_public_inputs=True)
    pending.append((rows, want, cm))
    if len(pending) == B:
        flush()
flush()
stats["distinct_inner_proofs"] = len(cache)
stats["seconds"] = round(time.time() - t0, 1)
print(json.dumps(stats))
sys.exit(1 if stats["mismatches"] else 0)
