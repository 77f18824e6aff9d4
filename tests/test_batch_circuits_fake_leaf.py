"""The private-batch and public-batch circuits on ARBITRARY leaf public inputs, the reference's way: build_fake_leaf_circuit
(wormhole/tests/test-helpers/src/fake_leaf.rs: 21 free public inputs + three range checks) restated on the native builder
(QPGPU_LEAF_FRAGMENT_FAKE_LEAF), a complete-verifier private-batch circuit over it, and the scenarios of the reference's circuit
tests (private_batch/circuit/circuit_logic.rs:853-2000: recursive_aggregation_tree, _different_blocks_fails,
_mismatched_asset_id_fails, _with_dummy_proofs, _masks_dummy_exit_accounts_to_zero, _real_proof_in_every_slot_succeeds,
_all_dummy_proofs, _mismatched_volume_fee_bps_fails, _exit_sum_overflow_fails, _dummy_nullifiers_are_replaced,
nullifier_region_is_canonically_sorted, private_batch_rejects_malicious_circuit_proofs). CPU only: the oracle proves the fake
leaves and generates the wrapper's witness WITHOUT being handed public inputs (plonky2's order: the generators produce them), so a
conflict is a violated constraint of the circuit, not a disagreement with a supplied value; the public inputs read back from the
trace must equal the host restatement's (qpgpu_private_batch_outputs) where that accepts the slots, and the host restatement must
refuse where the circuit has no witness."""
import numpy as np
import pytest

import oracle_binding as ob

P = 0xFFFFFFFF00000001
N = 4
BLOCK_A, BLOCK_B = (0xB10C0001, 2, 3, 4), (0xB10C0002, 2, 3, 4)
EXITS = [(0x11110001, 1, 0x11110002, 2), (0x22220001, 1, 0x22220002, 2), (0x33330001, 1, 0x33330002, 2), (0x44440001, 1, 0x44440002, 2)]


def leaf_pis(nullifier, block=BLOCK_A, asset=0, fee=10, out=(100, 0), exits=(EXITS[0], (0, 0, 0, 0)), number=42):
    p = np.zeros(21, dtype=np.uint64)
    p[0], p[1], p[2], p[3] = asset, out[0], out[1], fee
    p[4:8] = nullifier if isinstance(nullifier, (tuple, list)) else (nullifier, 7, 8, 9)
    p[8:12], p[12:16] = exits[0], exits[1]
    p[16:20] = block
    p[20] = number
    return p


def dummy_pis(**kw):
    return leaf_pis(kw.pop("nullifier", 0) if "nullifier" in kw else (0, 0, 0, 0), block=(0, 0, 0, 0), out=kw.pop("out", (0, 0)), exits=kw.pop("exits", ((0, 0, 0, 0), (0, 0, 0, 0))), number=0, **kw)


@pytest.fixture(scope="module")
def env(pkg, orc):
    L, R = pkg.leaf, pkg.recursion
    fake = L.LeafCircuit(fragment=L.FRAGMENT_FAKE_LEAF)
    assert fake.info["degree_bits"] == 5
    op = ob.OracleCircuit(orc, fake.pack)
    ver = pkg.Verifier(fake.pack)
    w = R.WrapperCircuit(fake.pack, ver, N, logic="private_batch", verify=True)
    pi_cells = pkg.pack_public_input_cells(w.pack)
    none = np.zeros(0, dtype=np.uint64)
    cache = {}

    def prove(pis):
        key = pis.tobytes()
        if key not in cache:
            rc, wires, _ = orc.generate_witness(fake.pack, none, none, pis)
            assert rc == orc.WIT_OK
            cache[key] = op.prove(wires, pis)
        return cache[key]

    def run(rows, pre):
        """-> (rc, public inputs the CIRCUIT computed or None)"""
        proofs = [prove(r) for r in rows]
        cells, vals, _ = w.commit(proofs, preimages=pre, public_inputs=np.zeros(21 * N + 8, dtype=np.uint64))
        rc, wires, _ = orc.generate_witness(w.pack, cells, vals, None)
        if rc != orc.WIT_OK:
            return rc, None
        return rc, wires[(pi_cells % 135).astype(np.int64), (pi_cells // 135).astype(np.int64)]

    yield pkg, orc, w, run, prove, fake
    op.close(); ver.close()


PRE = np.arange(4 * N, dtype=np.uint64).reshape(N, 4) + 1


def host(pkg, rows, pre=PRE):
    return pkg.aggregation.private_batch_outputs(np.stack(rows), pre)


def test_recursive_aggregation_tree(env):
    pkg, orc, w, run, prove, fake = env
    A = pkg.aggregation
    # four real proofs of one block; leaves 0 and 2 pay the same account, leaf 1 pays one account twice
    rows = [leaf_pis(101, out=(100, 5), exits=(EXITS[0], EXITS[1])), leaf_pis(102, out=(7, 8), exits=(EXITS[2], EXITS[2])),
            leaf_pis(103, out=(30, 0), exits=(EXITS[0], (0, 0, 0, 0))), leaf_pis(104, out=(1, 2), exits=(EXITS[3], EXITS[1]))]
    rc, got = run(rows, PRE)
    assert rc == orc.WIT_OK and got.tolist() == host(pkg, rows).tolist()
    hdr, slots, nulls = A.parse_private_batch_public_inputs(got)
    acct = lambda e: b"".join(int(x).to_bytes(8, "little") for x in e)
    assert hdr["num_exit_slots"] == 8 and hdr["block_number"] == 42 and hdr["volume_fee_bps"] == 10
    assert [s for s in slots if s[0]] == [(130, acct(EXITS[0])), (7, acct(EXITS[1])), (15, acct(EXITS[2])), (1, acct(EXITS[3]))]
    # (an exit account of all zeroes with amount 0 is indistinguishable from a dummy / duplicate slot: that is the point)
    assert sorted(n[:8] for n in nulls) == sorted(int(v).to_bytes(8, "little") for v in (101, 102, 103, 104))


def test_real_proof_in_every_slot_and_any_order(env):
    pkg, orc, w, run, prove, fake = env
    rows = [leaf_pis(201 + i, out=(10 * (i + 1), i), exits=(EXITS[i], EXITS[(i + 1) % 4])) for i in range(4)]
    for order in ([0, 1, 2, 3], [3, 1, 0, 2]):
        r = [rows[i] for i in order]
        rc, got = run(r, PRE)
        assert rc == orc.WIT_OK and got.tolist() == host(pkg, r).tolist()


@pytest.mark.parametrize("name,make", [
    ("different_blocks", lambda: [leaf_pis(1), leaf_pis(2, block=BLOCK_B), leaf_pis(3), leaf_pis(4)]),
    ("mismatched_asset_id", lambda: [leaf_pis(1, asset=5), leaf_pis(2, asset=5), leaf_pis(3, asset=6), leaf_pis(4, asset=5)]),
    ("mismatched_asset_id_in_a_dummy", lambda: [leaf_pis(1), dummy_pis(asset=5), leaf_pis(3), leaf_pis(4)]),
    ("mismatched_volume_fee_bps", lambda: [leaf_pis(1), leaf_pis(2, fee=11), leaf_pis(3), leaf_pis(4)]),
    ("replayed_leaf", lambda: [leaf_pis(1), leaf_pis(2), leaf_pis(1), leaf_pis(4)]),
    ("exit_sum_overflow", lambda: [leaf_pis(10 + i, out=(0xFFFFFFFF // 2, 0)) for i in range(4)]),
])
def test_batches_the_circuit_has_no_witness_for(env, name, make):
    pkg, orc, w, run, prove, fake = env
    rows = make()
    rc, _ = run(rows, PRE)
    assert rc == orc.WIT_CONFLICT, name
    with pytest.raises(pkg.QpGpuError) as e:
        host(pkg, rows)
    assert e.value.code == -4, name


def test_dummy_slots(env):
    pkg, orc, w, run, prove, fake = env
    A = pkg.aggregation
    acct = lambda e: b"".join(int(x).to_bytes(8, "little") for x in e)
    # dummies first: the references come from the first REAL slot; a dummy's fee may differ; a dummy carrying attacker-chosen exit
    # bytes and amounts is masked to the zero account (the poisoned padding template of the audit finding)
    rows = [dummy_pis(fee=99), leaf_pis(31, out=(40, 2), exits=(EXITS[1], EXITS[2])), dummy_pis(exits=(EXITS[3], EXITS[0]), out=(0, 0)), leaf_pis(32, out=(1, 0), exits=(EXITS[1], (0, 0, 0, 0)))]
    rc, got = run(rows, PRE)
    assert rc == orc.WIT_OK and got.tolist() == host(pkg, rows).tolist()
    hdr, slots, nulls = A.parse_private_batch_public_inputs(got)
    assert hdr["volume_fee_bps"] == 10 and hdr["block_hash"][:8] == int(BLOCK_A[0]).to_bytes(8, "little")
    assert [s for s in slots if s[0]] == [(41, acct(EXITS[1])), (2, acct(EXITS[2]))] and all(a == bytes(32) for s, a in slots if not s)
    # dummy nullifiers are replaced by H(H(preimage)) of the slot's preimage, the real ones forwarded; the region is sorted
    want = {A.dummy_nullifier(PRE[0]), A.dummy_nullifier(PRE[2]), acct((31, 7, 8, 9)), acct((32, 7, 8, 9))}
    assert set(nulls) == want
    key = lambda d: [int.from_bytes(d[8 * i:8 * i + 8], "little") for i in range(4)]
    assert [key(d) for d in nulls] == sorted(key(d) for d in nulls)
    # two dummies with EQUAL nullifier words are exempt from the distinctness constraint
    rows2 = [dummy_pis(), leaf_pis(33), dummy_pis(), leaf_pis(34)]
    assert run(rows2, PRE)[0] == orc.WIT_OK
    # all slots dummy: the circuit accepts (zero references; the chain rejects block hash zero), commit's preflight is what refuses it
    rows3 = [dummy_pis()] * N
    rc, got3 = run(rows3, PRE)
    assert rc == orc.WIT_OK and got3.tolist() == host(pkg, rows3).tolist() and got3[1:8].tolist() == [0] * 7


def test_nullifier_region_is_canonically_sorted(env):
    """Limbs next to the field order and equal leading limbs: the comparators work on canonical 32-bit halves (a prover cannot
    present 0 as p), most significant limb first."""
    pkg, orc, w, run, prove, fake = env
    nl = [(P - 1, 0, 0, 0), (P - 1, 0, 0, 1), (0, P - 1, P - 1, P - 1), (0xFFFFFFFF, 0xFFFFFFFF00000000, 1, 0)]
    rows = [leaf_pis(list(n), out=(i + 1, 0), exits=(EXITS[i], (0, 0, 0, 0))) for i, n in enumerate(nl)]
    rc, got = run(rows, PRE)
    assert rc == orc.WIT_OK and got.tolist() == host(pkg, rows).tolist()
    region = got[8 + 2 * N * 5:8 + 2 * N * 5 + 4 * N].reshape(N, 4).tolist()
    assert region == sorted([list(n) for n in nl])


def test_private_batch_rejects_malicious_circuit_proofs(env):
    """A proof of ANOTHER circuit with the same proof shape (21 free public inputs, no range checks) under the legitimate
    circuit's verifier data baked into the wrapper: no witness (private_batch_rejects_malicious_circuit_proofs)."""
    pkg, orc, w, run, prove, fake = env
    L = pkg.leaf
    mal_pack = fake.pack.copy()
    # the malicious circuit: the fake leaf with one constant of its constants/sigmas table changed (a different circuit of the same shape)
    h = pkg.pack_header(mal_pack)
    pos = 18 + h["num_arity_rounds"] + 8 * h["num_gates"] + h["num_routed_wires"] + 4
    mal_pack[pos + (h["num_selectors"] << h["degree_bits"]) + 31] ^= 1          # a constant of the last (padding) row: constrains nothing
    oc = ob.OracleCircuit(orc, mal_pack)
    pis = leaf_pis(77, out=(5, 6))
    none = np.zeros(0, dtype=np.uint64)
    rc, wires, _ = orc.generate_witness(mal_pack, none, none, pis)
    assert rc == orc.WIT_OK
    bad = oc.prove(wires, pis)
    assert oc.verify(bad) == 0                                                  # a valid proof — of the other circuit
    oc.close()
    good = [prove(leaf_pis(71 + i)) for i in range(3)]
    cells, vals, _ = w.commit(good + [bad], preimages=PRE, public_inputs=np.zeros(21 * N + 8, dtype=np.uint64))
    assert orc.generate_witness(w.pack, cells, vals, None)[0] == orc.WIT_CONFLICT
    cells, vals, _ = w.commit(good + [prove(pis)], preimages=PRE, public_inputs=np.zeros(21 * N + 8, dtype=np.uint64))
    assert orc.generate_witness(w.pack, cells, vals, None)[0] == orc.WIT_OK


def test_fake_leaf_range_checks(env):
    """the fake leaf's own constraints (range_check(pis[1..3], 32)): an amount of 2^32 has no witness"""
    pkg, orc, w, run, prove, fake = env
    none = np.zeros(0, dtype=np.uint64)
    for at in (1, 2, 3):
        p = leaf_pis(5); p[at] = 1 << 32
        assert orc.generate_witness(fake.pack, none, none, p)[0] == orc.WIT_CONFLICT
