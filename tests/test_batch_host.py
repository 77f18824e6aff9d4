"""Host side of the two aggregation levels (include/qpgpu_batch.h): public-input parsers, commit preflights, padding /
shuffle / dummy preimages, and the native mirror of the wrapper circuits' outputs.

The cases follow the reference's own tests:
  wormhole/inputs/src/lib.rs:702-776                                   (parsers)
  wormhole/aggregator/src/private_batch/prover/lib.rs:556-815          (dummy template sentinel, batch compatibility)
  wormhole/aggregator/src/private_batch/circuit/circuit_logic.rs:853-1866 (what the private-batch circuit outputs)
  wormhole/aggregator/src/public_batch/prover/lib.rs:763-840, circuit/circuit_logic.rs:167-330
No GPU needed."""
import ctypes
import numpy as np
import pytest

import __graft_entry__ as ge

pkg = ge.load_package()
L = pkg.load_library()
P = 0xFFFFFFFF00000001
ERR = 400
LEAF = 21


def _u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


def call(fn, *args):
    """-> (rc, message)"""
    err = ctypes.create_string_buffer(ERR)
    rc = fn(*args, err)
    return rc, err.value.decode()


for name, argt in {
    "qpgpu_validate_proof_count": [ctypes.c_uint64, ctypes.c_char_p, ctypes.c_char_p],
    "qpgpu_leaf_public_inputs_parse": [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_char_p],
    "qpgpu_private_batch_public_inputs_parse": [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_char_p],
    "qpgpu_public_batch_public_inputs_parse": [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_char_p],
    "qpgpu_private_batch_preflight": [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_char_p],
    "qpgpu_dummy_leaf_template_check": [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p],
    "qpgpu_private_batch_arrange": [ctypes.c_size_t, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_char_p],
    "qpgpu_private_batch_outputs": [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_char_p],
    "qpgpu_public_batch_preflight": [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_char_p],
    "qpgpu_dummy_private_batch_template_check": [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p],
    "qpgpu_public_batch_outputs": [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_char_p],
}.items():
    getattr(L, name).argtypes = argt
    getattr(L, name).restype = ctypes.c_int
L.qpgpu_private_batch_pi_len.argtypes = [ctypes.c_size_t]; L.qpgpu_private_batch_pi_len.restype = ctypes.c_size_t
L.qpgpu_public_batch_pi_len.argtypes = [ctypes.c_size_t, ctypes.c_size_t]; L.qpgpu_public_batch_pi_len.restype = ctypes.c_size_t
L.qpgpu_poseidon2_hash_pad10.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]


class LeafPis(ctypes.Structure):
    _fields_ = [("asset_id", ctypes.c_uint32), ("output_amount_1", ctypes.c_uint32), ("output_amount_2", ctypes.c_uint32),
                ("volume_fee_bps", ctypes.c_uint32), ("nullifier", ctypes.c_uint8 * 32), ("exit_account_1", ctypes.c_uint8 * 32),
                ("exit_account_2", ctypes.c_uint8 * 32), ("block_hash", ctypes.c_uint8 * 32), ("block_number", ctypes.c_uint32)]


class PrivHdr(ctypes.Structure):
    _fields_ = [("num_exit_slots", ctypes.c_uint32), ("asset_id", ctypes.c_uint32), ("volume_fee_bps", ctypes.c_uint32),
                ("block_hash", ctypes.c_uint8 * 32), ("block_number", ctypes.c_uint32), ("n_leaf", ctypes.c_uint32)]


class PubHdr(ctypes.Structure):
    _fields_ = [("aggregator_address", ctypes.c_uint8 * 32), ("asset_id", ctypes.c_uint32), ("volume_fee_bps", ctypes.c_uint32),
                ("block_hash", ctypes.c_uint8 * 32), ("block_number", ctypes.c_uint32), ("total_exit_slots", ctypes.c_uint32)]


class Slot(ctypes.Structure):
    _fields_ = [("summed_output_amount", ctypes.c_uint32), ("exit_account", ctypes.c_uint8 * 32)]


def leaf_pis(asset, fee, block, nullifier=0, amounts=(0, 0), exits=((0,) * 4, (0,) * 4), number=0):
    """The reference tests' leaf_pis(asset, fee, block) / with_nullifier helpers: block and nullifier go into limb 0."""
    p = np.zeros(LEAF, dtype=np.uint64)
    p[0], p[1], p[2], p[3] = asset, amounts[0], amounts[1], fee
    p[4] = nullifier
    p[8:12] = exits[0]; p[12:16] = exits[1]
    p[16] = block
    p[20] = number
    return p


def parse_private(pis):
    pis = _u64(pis)
    hdr = PrivHdr(); slots = (Slot * 128)(); nulls = (ctypes.c_uint8 * (64 * 32))()
    rc, msg = call(L.qpgpu_private_batch_public_inputs_parse, pis.ctypes.data, pis.size, ctypes.byref(hdr), slots, nulls)
    return rc, msg, hdr, slots, nulls


# ------------------------------------------------------------------------------------------------------------ parsers

def test_validate_proof_count_enforces_canonical_range():
    assert call(L.qpgpu_validate_proof_count, 0, b"x")[0] != 0
    assert call(L.qpgpu_validate_proof_count, 65, b"x")[0] != 0
    assert call(L.qpgpu_validate_proof_count, 1, b"x")[0] == 0
    assert call(L.qpgpu_validate_proof_count, 64, b"x")[0] == 0
    assert "exceeds maximum allowed (64)" in call(L.qpgpu_validate_proof_count, 65, b"n")[1]


def test_aggregated_public_inputs_reject_malformed_padded_length():
    rc, msg, *_ = parse_private(np.zeros(9))
    assert rc != 0 and "malformed length 9 - expected 8 + N*21 felts" in msg


def test_aggregated_public_inputs_parse_header():
    pis = np.zeros(8 + LEAF, dtype=np.uint64)
    pis[0] = 2; pis[7] = 42
    rc, msg, hdr, slots, _ = parse_private(pis)
    assert rc == 0, msg
    assert hdr.num_exit_slots == 2 and hdr.block_number == 42 and hdr.n_leaf == 1


def test_aggregated_public_inputs_reject_header_slot_count_mismatch():
    pis = np.zeros(8 + LEAF, dtype=np.uint64)
    pis[0] = 1
    rc, msg, *_ = parse_private(pis)
    assert rc != 0 and "exit slot" in msg


def test_aggregated_public_inputs_reject_oversized_leaf_count():
    rc, msg, *_ = parse_private(np.zeros(8 + 65 * LEAF))
    assert rc != 0 and "exceeds maximum" in msg


def test_public_batch_parser_rejects_oversized_counts():
    header = np.zeros(12, dtype=np.uint64)
    hdr = PubHdr()
    rc, msg = call(L.qpgpu_public_batch_public_inputs_parse, header.ctypes.data, header.size, 1 << 63, 1, ctypes.byref(hdr), None, None)
    assert rc != 0 and "exceeds maximum" in msg
    assert L.qpgpu_public_batch_pi_len(8, 8) == 12 + 14 * 64 and L.qpgpu_public_batch_pi_len(65, 1) == 0
    assert L.qpgpu_private_batch_pi_len(8) == 176


def test_leaf_public_inputs_parse():
    p = leaf_pis(0, 10, 5, nullifier=9, amounts=(100, 7), exits=((1, 2, 3, 4), (5, 6, 7, 8)), number=77)
    out = LeafPis()
    rc, msg = call(L.qpgpu_leaf_public_inputs_parse, p.ctypes.data, p.size, ctypes.byref(out))
    assert rc == 0, msg
    assert (out.asset_id, out.output_amount_1, out.output_amount_2, out.volume_fee_bps, out.block_number) == (0, 100, 7, 10, 77)
    assert bytes(out.exit_account_1) == b"".join(int(v).to_bytes(8, "little") for v in (1, 2, 3, 4))
    assert bytes(out.block_hash)[:8] == (5).to_bytes(8, "little")
    assert "should contain 21 field elements, got 20" in call(L.qpgpu_leaf_public_inputs_parse, p.ctypes.data, 20, ctypes.byref(out))[1]
    p[1] = 1 << 32
    assert "output_amount_1" in call(L.qpgpu_leaf_public_inputs_parse, p.ctypes.data, p.size, ctypes.byref(out))[1]
    p[1] = 0; p[17] = P
    rc, msg = call(L.qpgpu_leaf_public_inputs_parse, p.ctypes.data, p.size, ctypes.byref(out))
    assert rc != 0 and "block_hash" in msg and "Chunk out of field range at index 1" in msg


# ----------------------------------------------------------------------------------------- dummy leaf template sentinel

def template_check(p):
    p = _u64(p)
    return call(L.qpgpu_dummy_leaf_template_check, p.ctypes.data, p.size)


def test_dummy_template_with_zero_sentinel_is_accepted():
    assert template_check(np.zeros(LEAF))[0] == 0


@pytest.mark.parametrize("index,value,needle", [(16, 1, "non-zero block_hash"), (0, 7, "non-zero asset_id"), (1, 5, "non-zero output amounts"),
                                                (8, 1, "non-zero exit account"), (12, 1, "non-zero exit account")])
def test_dummy_template_sentinel_violations_are_rejected(index, value, needle):
    p = np.zeros(LEAF, dtype=np.uint64)
    p[index] = value
    rc, msg = template_check(p)
    assert rc != 0 and needle in msg, msg


# ------------------------------------------------------------------------------------------- leaf batch compatibility

def preflight(rows, slots=None):
    a = _u64(np.stack(rows)) if len(rows) else np.zeros(0, dtype=np.uint64)
    return call(L.qpgpu_private_batch_preflight, a.ctypes.data if a.size else None, len(rows), slots if slots is not None else len(rows))


def test_compatible_leaf_batch_is_accepted():
    rows = [leaf_pis(0, 10, 1, 1), leaf_pis(0, 10, 1, 2), leaf_pis(0, 99, 0)]      # the dummy slot is exempt from block / fee consistency
    assert preflight(rows)[0] == 0


def test_duplicate_real_nullifier_leaf_batch_is_rejected():
    rc, msg = preflight([leaf_pis(0, 10, 1, 7), leaf_pis(0, 10, 1, 7)])
    assert rc != 0 and "same nullifier" in msg


def test_duplicate_dummy_nullifiers_are_exempt():
    assert preflight([leaf_pis(0, 10, 1, 7), leaf_pis(0, 10, 0, 7), leaf_pis(0, 10, 0, 7)])[0] == 0


def test_mixed_block_leaf_batch_is_rejected():
    rc, msg = preflight([leaf_pis(0, 10, 1, 1), leaf_pis(0, 10, 2, 2)])
    assert rc != 0 and "different block" in msg


def test_mixed_fee_leaf_batch_is_rejected():
    rc, msg = preflight([leaf_pis(0, 10, 1, 1), leaf_pis(0, 20, 1, 2)])
    assert rc != 0 and "volume_fee_bps" in msg


def test_all_dummy_leaf_batch_is_rejected():
    rc, msg = preflight([leaf_pis(0, 10, 0), leaf_pis(0, 10, 0)])
    assert rc != 0 and "all-dummy" in msg


def test_mixed_asset_leaf_batch_is_rejected_even_for_dummies():
    rc, msg = preflight([leaf_pis(0, 10, 1), leaf_pis(5, 10, 0)])
    assert rc != 0 and "asset" in msg


def test_commit_count_checks_and_padding_asset_rule():
    assert "no leaf proofs to aggregate" in preflight([], 8)[1]
    assert "too many proofs: got 3, expected at most 2" in preflight([leaf_pis(0, 10, 1, i) for i in range(3)], 2)[1]
    rc, msg = preflight([leaf_pis(3, 10, 1, 1)], 8)       # padding needed: real proofs must use the native asset
    assert rc != 0 and "real proof 0 has asset_id=3, but dummy proofs use asset_id=0" in msg
    assert preflight([leaf_pis(3, 10, 1, 1), leaf_pis(3, 10, 1, 2)], 2)[0] == 0     # a full batch may use any one asset


# -------------------------------------------------------------------------------------- padding, shuffle, preimages

def arrange(count, slots, seed):
    src = np.zeros(slots, dtype=np.uint32); pre = np.zeros(slots * 4, dtype=np.uint64)
    rc, msg = call(L.qpgpu_private_batch_arrange, count, slots, seed, src.ctypes.data, pre.ctypes.data)
    assert rc == 0, msg
    return src, pre.reshape(slots, 4)


def test_arrange_pads_shuffles_and_draws_preimages():
    src, pre = arrange(3, 8, bytes(range(32)))
    assert sorted(src.tolist()) == [0, 1, 2] + [0xFFFFFFFF] * 5
    assert (pre < np.uint64(P)).all() and len({tuple(r) for r in pre.tolist()}) == 8
    src2, pre2 = arrange(3, 8, bytes(range(32)))
    assert (src == src2).all() and (pre == pre2).all()                      # a seed reproduces the arrangement
    src3, pre3 = arrange(3, 8, None)                                         # operating-system entropy
    assert sorted(src3.tolist()) == sorted(src.tolist()) and not (pre3 == pre).all()
    # uniform shuffle: over many seeds the one real proof of a 4-slot batch lands in every slot about equally often
    hits = np.zeros(4, dtype=int)
    for s in range(400):
        hits[int(np.nonzero(arrange(1, 4, s.to_bytes(32, "little"))[0] == 0)[0][0])] += 1
    assert hits.min() > 60, hits
    assert arrange(1, 1, bytes(32))[0].tolist() == [0]


# ---------------------------------------------------------------------------- what the private-batch circuit outputs

EXITS = [tuple(0x1111_0001 * (k + 1) + j for j in range(4)) for k in range(8)]
NULLS = [(0x9000 - 17 * k, k, 2 * k, 3 * k) for k in range(8)]                     # strictly descending in limb 0
BLOCK = (0xB10C, 2, 3, 4)


def real_leaf(i, amounts, block=BLOCK, fee=10, asset=0, number=42):
    p = np.zeros(LEAF, dtype=np.uint64)
    p[0], p[1], p[2], p[3] = asset, amounts[0], amounts[1], fee
    p[4:8] = NULLS[i]
    p[8:12] = EXITS[i]; p[12:16] = EXITS[(i + 1) % 8]
    p[16:20] = block; p[20] = number
    return p


def outputs(rows, preimages=None):
    rows = _u64(np.stack(rows))
    n = rows.shape[0]
    pre = _u64(preimages if preimages is not None else np.arange(1, 4 * n + 1).reshape(n, 4))
    out = np.zeros(LEAF * n + 8, dtype=np.uint64)
    rc, msg = call(L.qpgpu_private_batch_outputs, rows.ctypes.data, n, pre.ctypes.data, out.ctypes.data)
    return rc, msg, out


def double_hash(pre):
    a = _u64(pre); t = np.zeros(4, dtype=np.uint64); o = np.zeros(4, dtype=np.uint64)
    assert L.qpgpu_poseidon2_hash_pad10(None, 0, a.ctypes.data, 4, t.ctypes.data) == 0
    assert L.qpgpu_poseidon2_hash_pad10(None, 0, t.ctypes.data, 4, o.ctypes.data) == 0
    return tuple(o.tolist())


def test_recursive_aggregation_tree():
    rng = np.random.default_rng(41)
    a1 = (rng.integers(0, 1 << 32, 8) >> 4).tolist(); a2 = (rng.integers(0, 1 << 32, 8) >> 4).tolist()
    rows = [real_leaf(i, (a1[i], a2[i])) for i in range(8)]
    rc, msg, pis = outputs(rows)
    assert rc == 0, msg
    assert pis.size == 8 * LEAF + 8 and pis[0] == 16 and pis[1] == 0 and pis[2] == 10
    assert tuple(pis[3:7].tolist()) == BLOCK and pis[7] == 42
    want = {}
    for i in range(8):      # off-circuit reference: output amounts summed per exit account
        want[EXITS[i]] = want.get(EXITS[i], 0) + a1[i]
        want[EXITS[(i + 1) % 8]] = want.get(EXITS[(i + 1) % 8], 0) + a2[i]
    got = {}
    for s in range(16):
        sm, acct = int(pis[8 + 5 * s]), tuple(pis[9 + 5 * s:13 + 5 * s].tolist())
        if sm:
            assert acct not in got
            got[acct] = sm
        else:
            assert acct == (0, 0, 0, 0)          # a duplicate's slot looks like an unused one
    assert got == want
    region = [tuple(pis[88 + 4 * k:92 + 4 * k].tolist()) for k in range(8)]
    assert region == sorted(NULLS)               # canonically sorted, not in slot order (the inputs are descending)
    assert not pis[120:].any()                   # zero padding up to 21 N + 8
    rc, msg, hdr, slots, nulls = parse_private(pis)
    assert rc == 0 and hdr.n_leaf == 8 and hdr.volume_fee_bps == 10, msg


def test_recursive_aggregation_tree_with_dummy_proofs_masks_exits_and_replaces_nullifiers():
    pre = np.arange(100, 116).reshape(4, 4)
    dummy = np.zeros(LEAF, dtype=np.uint64)
    dummy[3] = 99                                # the reusable template's own fee must not constrain the batch
    dummy[4:8] = (7, 7, 7, 7)                    # and its nullifier field is never forwarded
    poisoned = dummy.copy(); poisoned[8:12] = (0xBAD, 1, 2, 3)                  # a dummy leaf may carry arbitrary exit bytes
    rows = [dummy, real_leaf(0, (500, 20)), poisoned, real_leaf(1, (30, 4))]
    rc, msg, pis = outputs(rows, pre)
    assert rc == 0, msg
    assert pis[2] == 10 and tuple(pis[3:7].tolist()) == BLOCK                   # references come from the first NON-dummy slot (slot 1)
    slots = [(int(pis[8 + 5 * s]), tuple(pis[9 + 5 * s:13 + 5 * s].tolist())) for s in range(8)]
    assert slots[0] == slots[1] == slots[4] == slots[5] == (0, (0, 0, 0, 0))    # dummy slots masked to the zero account
    assert slots[2] == (500, EXITS[0]) and slots[3] == (20 + 30, EXITS[1]) and slots[6] == (0, (0, 0, 0, 0)) and slots[7] == (4, EXITS[2])
    region = [tuple(pis[48 + 4 * k:52 + 4 * k].tolist()) for k in range(4)]
    assert region == sorted([double_hash(pre[0]), NULLS[0], double_hash(pre[2]), NULLS[1]])
    assert (7, 7, 7, 7) not in region


def test_recursive_aggregation_real_proof_in_every_slot_succeeds():
    for pos in range(4):
        rows = [np.zeros(LEAF, dtype=np.uint64) for _ in range(4)]
        rows[pos] = real_leaf(3, (11, 0))
        rc, msg, pis = outputs(rows)
        assert rc == 0, msg
        assert tuple(pis[3:7].tolist()) == BLOCK and pis[7] == 42 and int(pis[8 + 5 * 2 * pos]) == 11


def test_recursive_aggregation_tree_all_dummy_proofs_yields_zero_references():
    rc, msg, pis = outputs([np.zeros(LEAF, dtype=np.uint64)] * 2)
    assert rc == 0 and not pis[1:18].any() and pis[0] == 4, msg


@pytest.mark.parametrize("mutate,needle", [
    (lambda r: r[1].__setitem__(slice(16, 20), (1, 1, 1, 1)), "block hash"),
    (lambda r: r[1].__setitem__(0, 5), "asset_id"),
    (lambda r: r[1].__setitem__(3, 11), "volume_fee_bps"),
    (lambda r: r[1].__setitem__(slice(4, 8), NULLS[0]), "same real nullifier"),
])
def test_private_batch_outputs_refuse_what_the_circuit_cannot_prove(mutate, needle):
    rows = [real_leaf(0, (1, 2)), real_leaf(1, (3, 4))]
    mutate(rows)
    rc, msg, _ = outputs(rows)
    assert rc == -4 and needle in msg, msg


def test_recursive_aggregation_tree_exit_sum_overflow_fails():
    rows = [real_leaf(0, (0xFFFFFFFF, 0)), real_leaf(7, (0, 1))]               # leaf 7's second exit is EXITS[0] again
    rc, msg, _ = outputs(rows)
    assert rc == -4 and "32-bit range check" in msg


# ------------------------------------------------------------------------------------------------------- public batch

def private_outputs(leaf_rows):
    rc, msg, pis = outputs(leaf_rows)
    assert rc == 0, msg
    return pis


def test_public_batch_outputs_forward_in_order_and_zero_dummy_inners():
    inner_a = private_outputs([real_leaf(0, (5, 6)), real_leaf(1, (7, 8))])
    inner_dummy = private_outputs([np.zeros(LEAF, dtype=np.uint64)] * 2)      # an all-dummy private batch (its nullifiers are hashes)
    inner_b = private_outputs([real_leaf(2, (9, 1)), real_leaf(3, (2, 3))])
    rows = _u64(np.stack([inner_dummy, inner_a, inner_b]))
    addr = b"".join(int(v).to_bytes(8, "little") for v in (11, 12, 13, 14))
    n = L.qpgpu_public_batch_pi_len(3, 2)
    out = np.zeros(n, dtype=np.uint64)
    rc, msg = call(L.qpgpu_public_batch_outputs, rows.ctypes.data, 3, 2, addr, out.ctypes.data)
    assert rc == 0, msg
    assert out[:4].tolist() == [11, 12, 13, 14] and out[4] == 0 and out[5] == 10 and tuple(out[6:10].tolist()) == BLOCK and out[10] == 42
    assert out[11] == 12
    assert not out[12:32].any() and (out[32:52] == inner_a[8:28]).all() and (out[52:72] == inner_b[8:28]).all()
    assert not out[72:80].any() and (out[80:88] == inner_a[28:36]).all() and (out[88:96] == inner_b[28:36]).all()
    hdr = PubHdr(); slots = (Slot * 12)(); nulls = (ctypes.c_uint8 * (6 * 32))()
    rc, msg = call(L.qpgpu_public_batch_public_inputs_parse, out.ctypes.data, out.size, 3, 2, ctypes.byref(hdr), slots, nulls)
    assert rc == 0 and hdr.total_exit_slots == 12 and hdr.block_number == 42 and slots[4].summed_output_amount == 5, msg
    # the wrong dimensions are refused by length
    assert "expected" in call(L.qpgpu_public_batch_public_inputs_parse, out.ctypes.data, out.size, 2, 2, ctypes.byref(hdr), None, None)[1]
    # a second real inner from another block cannot be proven
    other = private_outputs([real_leaf(4, (1, 1), block=(9, 9, 9, 9)), real_leaf(5, (1, 1), block=(9, 9, 9, 9))])
    rows2 = _u64(np.stack([inner_a, other]))
    out2 = np.zeros(L.qpgpu_public_batch_pi_len(2, 2), dtype=np.uint64)
    rc, msg = call(L.qpgpu_public_batch_outputs, rows2.ctypes.data, 2, 2, addr, out2.ctypes.data)
    assert rc == -4 and "block hash" in msg


def test_public_batch_preflight_and_dummy_template():
    inner_a = private_outputs([real_leaf(0, (5, 6)), real_leaf(1, (7, 8))])
    inner_dummy = private_outputs([np.zeros(LEAF, dtype=np.uint64)] * 2)
    other_block = private_outputs([real_leaf(4, (1, 1), block=(9, 9, 9, 9)), real_leaf(5, (1, 1), block=(9, 9, 9, 9))])
    other_fee = private_outputs([real_leaf(4, (1, 1), fee=12), real_leaf(5, (1, 1), fee=12)])

    def pre(rows, m):
        a = _u64(np.stack(rows)) if rows else np.zeros(0, dtype=np.uint64)
        return call(L.qpgpu_public_batch_preflight, a.ctypes.data if a.size else None, len(rows), 50, m)
    assert pre([inner_a, inner_dummy], 4)[0] == 0
    assert "no private-batch proofs to aggregate" in pre([], 4)[1]
    assert "Expected at most 1 private-batch proofs, but got 2" in pre([inner_a, inner_dummy], 1)[1]
    assert "all-dummy" in pre([inner_dummy, inner_dummy], 4)[1]                   # commit_rejects_all_dummy_batch
    assert "different block" in pre([inner_a, other_block], 4)[1]                # commit_rejects_batch_incompatible_private_batch_proofs
    assert "volume_fee_bps" in pre([inner_a, other_fee], 4)[1]
    a = _u64(inner_a)
    assert "malformed" in call(L.qpgpu_public_batch_preflight, a.ctypes.data, 1, 49, 4)[1]
    # dummy private-batch template: the all-dummy batch passes, a real one and a marked exit account do not
    assert call(L.qpgpu_dummy_private_batch_template_check, _u64(inner_dummy).ctypes.data, inner_dummy.size)[0] == 0
    assert "non-zero block_hash" in call(L.qpgpu_dummy_private_batch_template_check, a.ctypes.data, a.size)[1]
    marked = inner_dummy.copy(); marked[9] = 1
    assert "non-zero exit account at slot 0" in call(L.qpgpu_dummy_private_batch_template_check, marked.ctypes.data, marked.size)[1]
    paying = inner_dummy.copy(); paying[13] = 3
    assert "non-zero payout at slot 1 (3)" in call(L.qpgpu_dummy_private_batch_template_check, paying.ctypes.data, paying.size)[1]


def test_random_field_elements():
    """qpgpu_random_field_elements (RandomValueGenerator's F::rand() for the blinding wires of a zero-knowledge circuit): canonical,
    reproducible under a seed, fresh without one."""
    L.qpgpu_random_field_elements.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p]
    L.qpgpu_random_field_elements.restype = ctypes.c_int
    err = ctypes.create_string_buffer(200)
    def draw(seed, n=4096):
        out = np.empty(n, dtype=np.uint64)
        assert L.qpgpu_random_field_elements(seed, out.ctypes.data, n, err) == 0
        return out
    a, b, c, d = draw(bytes(32)), draw(bytes(32)), draw(bytes([1] * 32)), draw(None)
    assert np.array_equal(a, b) and not np.array_equal(a, c) and not np.array_equal(a, d) and not np.array_equal(d, draw(None))
    for v in (a, c, d):
        assert int(v.max()) < 0xFFFFFFFF00000001 and len(np.unique(v)) == v.size
    assert abs(float((d >> np.uint64(63)).mean()) - 0.5) < 0.05
    assert L.qpgpu_random_field_elements(None, None, 4, err) != 0
