"""Fuzz of the leaf path on the device against the oracle and the host constraint check: random CircuitInputs — valid spends with
random secrets, amounts under the fee rule, Merkle paths of 0..16 levels at random positions, dummies — and mutations of them
(one byte of a secret / nullifier / block hash / sibling / account, an amount, the depth, a position). For every input:
  * qpgpu_leaf_check_constraints' verdict (the host restatement of what the circuit constrains),
  * stage s1 on the device in lockstep batches (qpgpu_generate_witness_partial_batch_dev): status per witness,
  * oracle/witness.c on the same assignments,
must agree (satisfiable / "set twice with different values"), and for satisfiable inputs the device's wire matrix must equal the
oracle's cell for cell; every 16th satisfiable input is also proven and verified. usage: python tests/soak/fuzz_leaf_inputs.py [count] [seed] [hints]"""
import ctypes, json, sys, time
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import __graft_entry__ as ge
import leaf_cases as lc
import oracle_binding as ob
pkg = ge.load_package()
orc = ob.Oracle()
gpu = pkg.QpGpu(0)
L = pkg.leaf
count = int(sys.argv[1]) if len(sys.argv) > 1 else 512
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
HINTS = len(sys.argv) > 3 and sys.argv[3] == "hints"
leaf = L.LeafCircuit()
B = 32
circ = pkg.Circuit(gpu, leaf.pack, max_batch=B)
ver = pkg.Verifier(leaf.pack, circuit=circ)
nw, n = 135, 1 << leaf.info["degree_bits"]
d = gpu.alloc(B * nw * n * 8)
check = L._lib().qpgpu_leaf_check_constraints
check.argtypes = [ctypes.c_void_p, ctypes.c_char_p]; check.restype = ctypes.c_int


def canon32():
    b = rng.integers(0, 256, 32, dtype=np.uint8); b[7::8] &= 0x7F
    return b.tobytes()


def random_spend():
    x = L.LeafInputs()
    secret = canon32()
    x.asset_id = int(rng.integers(0, 3)); x.volume_fee_bps = int(rng.integers(0, 2000))
    x.transfer_count = int(rng.integers(0, 1 << 40)); x.input_amount = int(rng.integers(1, 1 << 31))
    budget = x.input_amount * (10000 - x.volume_fee_bps) // 10000
    o1 = int(rng.integers(0, budget + 1)); o2 = int(rng.integers(0, budget - o1 + 1))
    x.output_amount_1, x.output_amount_2 = o1, o2
    unsp = L.unspendable_account(secret)
    x.set32("secret", secret).set32("unspendable_account", unsp).set32("nullifier", L.nullifier(secret, x.transfer_count))
    x.set32("exit_account_1", canon32()).set32("exit_account_2", canon32())
    depth = int(rng.integers(0, 17))
    sibs = [[canon32() for _ in range(3)] for _ in range(depth)]
    sorted_sibs, positions, root = L.zk_proof_from_unsorted(L.zk_leaf_hash(unsp, x.transfer_count, x.asset_id, x.input_amount), sibs)
    x.zk_merkle_depth = depth
    if depth:
        ctypes.memmove(x.zk_merkle_siblings, sorted_sibs, len(sorted_sibs))
    for l, p in enumerate(positions):
        x.zk_merkle_positions[l] = p
    x.set32("zk_tree_root", root).set32("parent_hash", canon32()).set32("state_root", canon32()).set32("extrinsics_root", canon32())
    x.block_number = int(rng.integers(0, 1 << 32))
    dg = rng.integers(0, 256, 110, dtype=np.uint8).tobytes()
    ctypes.memmove(x.digest, dg, 110)
    x.set32("block_hash", L.block_hash(bytes(x.parent_hash), x.block_number, bytes(x.state_root), bytes(x.extrinsics_root), root, dg))
    if o1 == 0 and o2 == 0 and rng.integers(0, 2):       # a spend with zero outputs and a zero block hash IS a dummy
        x.set32("block_hash", bytes(32))
    return x


def mutate(x):
    y = x.copy()
    k = int(rng.integers(0, 9))
    if k == 0: y.secret[int(rng.integers(0, 32)) // 8 * 8] ^= 1 << int(rng.integers(0, 8))
    elif k == 1: y.nullifier[int(rng.integers(0, 4)) * 8] ^= 1
    elif k == 2: y.block_hash[int(rng.integers(0, 4)) * 8 + 1] ^= 4
    elif k == 3 and y.zk_merkle_depth: y.zk_merkle_siblings[(int(rng.integers(0, y.zk_merkle_depth)) * 3 + int(rng.integers(0, 3))) * 32 + 8] ^= 2
    elif k == 4: y.unspendable_account[16] ^= 1
    elif k == 5: y.output_amount_1 = (y.output_amount_1 + y.input_amount) & 0xFFFFFFFF
    elif k == 6 and y.zk_merkle_depth: y.zk_merkle_positions[0] = (y.zk_merkle_positions[0] + 1) % 4
    elif k == 7: y.input_amount = (y.input_amount + 1) & 0xFFFFFFFF
    else: y.transfer_count ^= 1
    return y


t0 = time.time()
xs = []
while len(xs) < count:
    x = random_spend() if rng.integers(0, 8) else lc.dummy_inputs(L)
    xs.append(x)
    if rng.integers(0, 2) and len(xs) < count:
        xs.append(mutate(x))
stats = {"hash_hints": HINTS, "inputs": count, "satisfiable": 0, "unsatisfiable": 0, "witnesses_compared": 0, "proofs_verified": 0, "mismatches": 0}
err = ctypes.create_string_buffer(200)
for k0 in range(0, count, B):
    chunk = xs[k0:k0 + B]
    com = [leaf.commit(x) for x in chunk]
    cells = com[0][0]
    if HINTS:    # the device gets the front-end's hash hints as well (the oracle below keeps the 299 assignments: same verdict, same witness)
        comh = [leaf.commit(x, hash_hints=True) for x in chunk]
        st = circ.generate_witness_partial_batch_dev(comh[0][0], np.stack([c[1] for c in comh]), np.stack([c[2] for c in comh]), d)
    else:
        st = circ.generate_witness_partial_batch_dev(cells, np.stack([c[1] for c in com]), np.stack([c[2] for c in com]), d)
    wires = d.download(len(chunk) * nw * n).reshape(len(chunk), nw, n)
    for i, x in enumerate(chunk):
        host_ok = check(ctypes.byref(x), err) == 0
        rc, want, _ = orc.generate_witness(leaf.pack, *com[i])
        dev_ok = st[i] == 0
        if not (host_ok == dev_ok == (rc == orc.WIT_OK)):
            stats["mismatches"] += 1
            print("MISMATCH input", k0 + i, "host", host_ok, err.value.decode(), "device", st[i], "oracle", rc)
            continue
        if dev_ok:
            stats["satisfiable"] += 1
            if not np.array_equal(wires[i], want):
                stats["mismatches"] += 1; print("WITNESS DIFFERS input", k0 + i)
            stats["witnesses_compared"] += 1
            if stats["satisfiable"] % 16 == 0:
                proof = circ.prove_dev(d.ptr + 8 * i * nw * n, com[i][2])
                if not ver.verify(proof):
                    stats["mismatches"] += 1; print("PROOF REJECTED input", k0 + i)
                stats["proofs_verified"] += 1
        else:
            stats["unsatisfiable"] += 1
stats["seconds"] = round(time.time() - t0, 1)
print(json.dumps(stats))
sys.exit(1 if stats["mismatches"] else 0)
