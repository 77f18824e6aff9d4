"""The chain's 4-ary ZK Merkle tree and the leaf circuit's constraints, natively (include/qpgpu_leaf.h, host only).

Cases follow the reference's own tests: common/src/zk_merkle.rs:398-814 (node hashing, position hints, proof verification,
canonicality, depth bounds) and the leaf circuit's negative tests (wormhole/tests/src/circuit/{nullifier_tests.rs:53-58,
block_header_tests.rs:22-95, unspendable_account_tests.rs}), which observe a violated binding as a failed prove; here the same
bindings are evaluated on the host by qpgpu_leaf_check_constraints. Hashes are Poseidon2 with the pinned parameter set (all
seven reference known-answer vectors, tests/test_leaf_witness.py)."""
import ctypes
import json
import os

import numpy as np
import pytest

from test_leaf_witness import LeafInputs, KATS, header_digest

P = 0xFFFFFFFF00000001


@pytest.fixture(scope="module")
def L(pkg):
    lib = pkg.load_library()
    cp, vp, sz = ctypes.c_char_p, ctypes.c_void_p, ctypes.c_size_t
    lib.qpgpu_zk_leaf_hash.argtypes = [cp, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, cp]
    lib.qpgpu_zk_hash_node_presorted.argtypes = [cp, cp]
    lib.qpgpu_zk_hash_node.argtypes = [cp, cp]
    lib.qpgpu_zk_insert_at_position.argtypes = [cp, cp, ctypes.c_uint, cp]
    lib.qpgpu_zk_proof_verify.argtypes = [cp, cp, cp, sz, cp]
    lib.qpgpu_zk_proof_from_unsorted.argtypes = [cp, cp, sz, cp, cp, cp, cp]
    lib.qpgpu_leaf_check_constraints.argtypes = [vp, cp]
    lib.qpgpu_leaf_unspendable_account.argtypes = [vp, sz, cp, vp]
    lib.qpgpu_leaf_nullifier.argtypes = [vp, sz, cp, ctypes.c_uint64, vp]
    lib.qpgpu_leaf_block_hash.argtypes = [vp, sz, cp, ctypes.c_uint32, cp, cp, cp, cp, vp]
    return lib


def h(n):
    """a canonical 32-byte hash: four little-endian limbs below p"""
    return b"".join(((n * 0x9E3779B97F4A7C15 + k * 0x1234567) % P).to_bytes(8, "little") for k in range(4))


def node(L, children, presorted=False):
    out = ctypes.create_string_buffer(32)
    rc = (L.qpgpu_zk_hash_node_presorted if presorted else L.qpgpu_zk_hash_node)(b"".join(children), out)
    return rc, out.raw


def from_unsorted(L, leaf, levels):
    depth = len(levels)
    so = ctypes.create_string_buffer(max(96 * depth, 1)); po = ctypes.create_string_buffer(max(depth, 1))
    root = ctypes.create_string_buffer(32); err = ctypes.create_string_buffer(160)
    rc = L.qpgpu_zk_proof_from_unsorted(leaf, b"".join(b"".join(lv) for lv in levels), depth, so, po, root, err)
    return rc, so.raw[:96 * depth], po.raw[:depth], root.raw, err.value.decode()


def test_hash_node_is_deterministic_and_order_independent(L):
    c = [h(1), h(2), h(3), h(4)]
    rc, a = node(L, c)
    assert rc == 0 and a == node(L, c)[1] and a != bytes(32)
    assert node(L, [c[3], c[1], c[0], c[2]])[1] == a                    # children are sorted before hashing
    assert node(L, sorted(c), presorted=True)[1] == a                   # hash_node_presorted matches hash_node
    assert node(L, [c[3], c[1], c[0], c[2]], presorted=True)[1] != a    # ... and does not sort
    assert all(int.from_bytes(a[8 * k:8 * k + 8], "little") < P for k in range(4))


def test_insert_at_position(L):
    sibs = [h(10), h(20), h(30)]
    cur = h(15)
    for pos in range(4):
        out = ctypes.create_string_buffer(128)
        assert L.qpgpu_zk_insert_at_position(cur, b"".join(sibs), pos, out) == 0
        got = [out.raw[32 * k:32 * k + 32] for k in range(4)]
        want = sibs[:pos] + [cur] + sibs[pos:]
        assert got == want
    assert L.qpgpu_zk_insert_at_position(cur, b"".join(sibs), 4, ctypes.create_string_buffer(128)) == -1     # rejects out of range


def test_proof_verification_depths_and_positions(L):
    leaf = h(100)
    assert L.qpgpu_zk_proof_verify(leaf, None, None, 0, leaf) == 1          # depth 0: the leaf is the root
    for depth in (1, 3, 16):                                                # MAX_DEPTH accepted
        levels = [[h(1000 * d + k) for k in range(3)] for d in range(depth)]
        rc, sorted_sibs, positions, root, msg = from_unsorted(L, leaf, levels)
        assert rc == 0, msg
        assert L.qpgpu_zk_proof_verify(leaf, sorted_sibs, positions, depth, root) == 1
        # from_unsorted computes the positions hash_node's sorting implies
        cur = leaf
        for d in range(depth):
            four = sorted([cur] + levels[d])
            assert positions[d] == four.index(cur)
            assert sorted_sibs[96 * d:96 * d + 96] == b"".join(x for x in four if x != cur)
            cur = node(L, four, presorted=True)[1]
        assert cur == root
        # a wrong root, a wrong sibling, a wrong position hint
        assert L.qpgpu_zk_proof_verify(leaf, sorted_sibs, positions, depth, h(5)) == 0
        bad = bytearray(sorted_sibs); bad[5] ^= 1
        assert L.qpgpu_zk_proof_verify(leaf, bytes(bad), positions, depth, root) == 0
        wrong = bytes([(positions[0] + 1) % 4]) + positions[1:]
        assert L.qpgpu_zk_proof_verify(leaf, sorted_sibs, wrong, depth, root) == 0          # test_wrong_position_hint_fails_default_verify
        assert L.qpgpu_zk_proof_verify(leaf, sorted_sibs, bytes([7]) + positions[1:], depth, root) == 0
    # oversized proofs are rejected before any hashing
    levels = [[h(k), h(k + 1), h(k + 2)] for k in range(17)]
    rc, *_rest, msg = from_unsorted(L, leaf, levels)
    assert rc == -1 and "exceeds MAX_DEPTH" in msg
    assert L.qpgpu_zk_proof_verify(leaf, bytes(96 * 17), bytes(17), 17, leaf) == 0


def test_noncanonical_hash_bytes_are_rejected(L):
    """hash_node_rejects_noncanonical_child_with_error_not_panic, from_unsorted_rejects_noncanonical_bytes,
    verify_rejects_noncanonical_{leaf,sibling}_alias: a limb v and v + p would hash alike."""
    good = h(7)
    limb0 = int.from_bytes(good[:8], "little")
    assert limb0 + P < 1 << 64 or True
    small = (5).to_bytes(8, "little") + good[8:]
    alias = (5 + P).to_bytes(8, "little") + good[8:]                        # same field element, different bytes
    c = [small, h(2), h(3), h(4)]
    assert node(L, c)[0] == 0
    assert node(L, [alias] + c[1:])[0] == -1
    rc, *_x, msg = from_unsorted(L, alias, [[h(1), h(2), h(3)]])
    assert rc == -1 and "leaf hash bytes are noncanonical" in msg
    rc, *_x, msg = from_unsorted(L, small, [[alias, h(2), h(3)]])
    assert rc == -1 and "sibling hash bytes are noncanonical" in msg
    rc, sibs, pos, root, _ = from_unsorted(L, small, [[h(1), h(2), h(3)]])
    assert L.qpgpu_zk_proof_verify(small, sibs, pos, 1, root) == 1
    assert L.qpgpu_zk_proof_verify(alias, sibs, pos, 1, root) == 0
    rc, sibs2, pos2, root2, _ = from_unsorted(L, h(9), [[small, h(2), h(3)]])
    i = [sibs2[32 * k:32 * k + 32] for k in range(3)].index(small)
    aliased = sibs2[:32 * i] + alias + sibs2[32 * i + 32:]
    assert L.qpgpu_zk_proof_verify(h(9), sibs2, pos2, 1, root2) == 1 and L.qpgpu_zk_proof_verify(h(9), aliased, pos2, 1, root2) == 0


# ------------------------------------------------------------------------------- the leaf circuit's constraints

def consistent_inputs(L, depth=5, secret=bytes(range(1, 33)), transfer_count=7, input_amount=100_000, outs=(60_000, 30_000), fee=10):
    """A CircuitInputs every binding of the leaf circuit holds for, derived natively from a secret and a header."""
    x = LeafInputs()
    sec = bytes(b % 250 for b in secret)
    x.secret[:] = sec
    x.transfer_count = transfer_count
    x.asset_id, x.output_amount_1, x.output_amount_2, x.volume_fee_bps, x.input_amount = 0, outs[0], outs[1], fee, input_amount
    buf = ctypes.create_string_buffer(32)
    assert L.qpgpu_leaf_unspendable_account(None, 0, sec, buf) == 0
    x.unspendable_account[:] = buf.raw
    assert L.qpgpu_leaf_nullifier(None, 0, sec, transfer_count, buf) == 0
    x.nullifier[:] = buf.raw
    x.exit_account_1[:] = h(41); x.exit_account_2[:] = h(42)
    assert L.qpgpu_zk_leaf_hash(bytes(x.unspendable_account), transfer_count, 0, input_amount, buf) == 0
    leaf = buf.raw
    levels = [[h(500 + 10 * d + k) for k in range(3)] for d in range(depth)]
    rc, sibs, pos, root, msg = from_unsorted(L, leaf, levels)
    assert rc == 0, msg
    x.zk_merkle_depth = depth
    x.zk_merkle_siblings[:96 * depth] = sibs
    x.zk_merkle_positions[:depth] = pos
    x.zk_tree_root[:] = root
    k = KATS["block_header_kats"][0]
    x.parent_hash[:] = h(77); x.state_root[:] = bytes.fromhex(k["state_root"]); x.extrinsics_root[:] = bytes.fromhex(k["extrinsics_root"])
    x.digest[:] = header_digest()
    x.block_number = 4242
    assert L.qpgpu_leaf_block_hash(None, 0, bytes(x.parent_hash), 4242, bytes(x.state_root), bytes(x.extrinsics_root), root, header_digest(), buf) == 0
    x.block_hash[:] = buf.raw
    return x


def check(L, x):
    err = ctypes.create_string_buffer(160)
    return L.qpgpu_leaf_check_constraints(ctypes.byref(x), err), err.value.decode()


def test_consistent_inputs_satisfy_the_leaf_circuit(L):
    for depth in (0, 1, 5, 16):
        rc, msg = check(L, consistent_inputs(L, depth=depth))
        assert rc == 0, (depth, msg)


def test_each_binding_is_enforced(L):
    x = consistent_inputs(L); x.nullifier[3] ^= 1                           # nullifier_tests: a nullifier not derived from the secret
    assert check(L, x) == (-4, 'nullifier is not H(H("~nullif~" || secret || transfer_count))')
    x = consistent_inputs(L); x.transfer_count += 1                         # ... or from another transfer count (also moves the leaf)
    assert check(L, x)[0] == -4
    x = consistent_inputs(L); x.secret[0] ^= 1                              # unspendable_account_tests: wrong secret
    assert check(L, x) == (-4, 'unspendable_account is not H(H("wormhole" || secret))')
    x = consistent_inputs(L); x.block_hash[9] ^= 1                          # block_header_tests: block hash not the header's
    assert check(L, x) == (-4, "block_hash is not the hash of the header contents")
    x = consistent_inputs(L); x.block_number += 1                           # any header field is committed to
    assert check(L, x)[1] == "block_hash is not the hash of the header contents"
    x = consistent_inputs(L); x.digest[50] ^= 1
    assert check(L, x)[1] == "block_hash is not the hash of the header contents"
    x = consistent_inputs(L); x.zk_merkle_siblings[40] ^= 1                 # a sibling that is not the tree's
    assert check(L, x) == (-4, "ZK Merkle path does not lead to the header's zk_tree_root")
    x = consistent_inputs(L); x.zk_merkle_positions[0] = (x.zk_merkle_positions[0] + 1) % 4
    assert check(L, x)[1] == "ZK Merkle path does not lead to the header's zk_tree_root"
    x = consistent_inputs(L); x.input_amount += 1                           # the leaf commits to the deposit's amount
    assert check(L, x)[1] == "ZK Merkle path does not lead to the header's zk_tree_root"


def test_fee_relation_and_ranges(L):
    assert check(L, consistent_inputs(L, input_amount=100_000, outs=(99_900, 0), fee=10))[0] == 0        # exactly input * (1 - 0.1 %)
    rc, msg = check(L, consistent_inputs(L, input_amount=100_000, outs=(99_901, 0), fee=10))
    assert rc == -4 and "fee constraint" in msg
    rc, msg = check(L, consistent_inputs(L, fee=10_001, outs=(0, 1)))
    assert rc == -4 and "exceeds 10000" in msg
    x = consistent_inputs(L); x.zk_merkle_depth = 17
    assert check(L, x)[0] == -1                                              # malformed inputs, as fill_witness reports them
    x = consistent_inputs(L); x.zk_merkle_positions[2] = 4
    assert check(L, x)[0] == -1


def test_dummy_inputs_skip_the_conditional_bindings(L):
    """Dummy proofs (zero block hash and zero outputs) may carry a random nullifier and no Merkle path, but the unspendable
    account binding stays; a zero block hash with non-zero outputs is not a dummy (circuit.rs:258-283)."""
    x = consistent_inputs(L, outs=(0, 0))
    x.block_hash[:] = bytes(32)
    x.nullifier[:] = h(999)
    x.zk_merkle_depth = 0
    assert check(L, x)[0] == 0
    x.secret[1] ^= 1
    assert check(L, x)[1] == 'unspendable_account is not H(H("wormhole" || secret))'
    y = consistent_inputs(L, outs=(5, 0))
    y.block_hash[:] = bytes(32)
    assert check(L, y)[0] == -4


def test_the_reference_dummy_inputs_satisfy_the_circuit(L):
    """build_dummy_circuit_inputs (wormhole/aggregator/src/dummy_proof.rs:58-84,125-170) — the reference's bench input and the
    padding template's witness — is a dummy whose unspendable account is the first address known-answer vector: the one binding
    that applies to it holds, with values the reference itself supplies."""
    from test_leaf_witness import dummy_inputs
    x = dummy_inputs()
    assert check(L, x) == (0, "")
    x.unspendable_account[0] ^= 1
    assert check(L, x)[0] == -4
