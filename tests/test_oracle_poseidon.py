"""Pins the oracle's Poseidon-v1 permutation, sponge, Merkle tree and challenger; exercises the
Poseidon2 plug and the byte<->felt codecs against the reference's own anchors."""
import ctypes
import json
import os

import numpy as np

from oracle_binding import Challenger

P = 0xFFFFFFFF00000001
GOLD = os.path.join(os.path.dirname(__file__), "golden")
V1 = json.load(open(os.path.join(GOLD, "poseidon_v1.json")))
KATS = json.load(open(os.path.join(GOLD, "poseidon2_kats.json")))


def test_round_constants_anchor(orc):
    rc = orc.round_constants()
    assert [int(x) for x in rc[:12]] == [int(h, 16) for h in V1["round_constants_first12"]]
    assert int(rc.max()) < int(V1["round_constant_bound"], 16)  # upstream's AVX2 precondition
    assert len(set(rc.tolist())) == 360


def test_permutation_vectors(orc):
    for v in V1["vectors"]:
        out = orc.poseidon(np.array(v["input"], dtype=np.uint64))
        assert [int(x) for x in out] == [int(h, 16) for h in v["output"]]


def test_sponge_shapes(orc):
    rng = np.random.default_rng(3)
    x = rng.integers(0, P, 20, dtype=np.uint64)
    # overwrite-mode absorption: first block then permute, compare with a manual walk
    st = np.zeros(12, dtype=np.uint64); st[:8] = x[:8]; st = orc.poseidon(st)
    st[:8] = x[8:16]; st = orc.poseidon(st)
    st[:4] = x[16:20]; st = orc.poseidon(st)
    assert np.array_equal(orc.hash_n_to_m(x, 4), st[:4])
    # squeezing more than one block permutes in between
    out = orc.hash_n_to_m(x, 11)
    assert np.array_equal(out[:8], st[:8]) and np.array_equal(out[8:], orc.poseidon(st)[:3])
    # hash_or_noop copies short rows
    assert orc.hash_or_noop(x[:3]).tolist() == x[:3].tolist() + [0]
    assert np.array_equal(orc.hash_or_noop(x[:5]), orc.hash_n_to_m(x[:5], 4))
    # two_to_one
    st = np.zeros(12, dtype=np.uint64); st[:8] = x[:8]
    assert np.array_equal(orc.two_to_one(x[:4], x[4:8]), orc.poseidon(st)[:4])


def test_merkle_and_paths(orc):
    rng = np.random.default_rng(4)
    leaves = rng.integers(0, P, (64, 7), dtype=np.uint64)
    for cap_h in (0, 2, 4, 6):
        dig, cap = orc.merkle(leaves, cap_h)
        assert cap.shape == (1 << cap_h, 4)
        for idx in (0, 1, 37, 63):
            path = orc.merkle_path(dig, 64, cap_h, idx)
            assert len(path) == 6 - cap_h
            cur, i = orc.hash_or_noop(leaves[idx]), idx
            for sib in path:
                cur = orc.two_to_one(cur, sib) if i % 2 == 0 else orc.two_to_one(sib, cur)
                i >>= 1
            assert np.array_equal(cur, cap[idx >> (6 - cap_h)])


def test_challenger_duplex(orc):
    ch = Challenger(orc)
    ch.observe([1, 2, 3])
    a = ch.get()
    st = np.zeros(12, dtype=np.uint64); st[:3] = [1, 2, 3]; st = orc.poseidon(st)
    assert a == int(st[7])               # pops from the end of the rate block
    assert ch.get() == int(st[6])
    ch.observe([9])                      # observing clears the output buffer
    st[0] = 9; st2 = orc.poseidon(st)
    assert ch.get() == int(st2[7])
    # 8 observations trigger a duplex immediately
    ch2 = Challenger(orc); ch2.observe(list(range(8)))
    s = np.zeros(12, dtype=np.uint64); s[:8] = range(8); s = orc.poseidon(s)
    assert ch2.get() == int(s[7])
    # proof-of-work response equals what observing the nonce then squeezing yields
    ch3 = Challenger(orc); ch3.observe([5, 6, 7, 8, 9])
    r = ch3.pow_response(123456)
    ch3.observe([123456])
    assert ch3.get() == r


def test_codecs_against_reference_anchors(orc):
    a = KATS["encoding_anchors"]
    d = np.empty(4, dtype=np.uint64)
    orc.lib.orc_bytes_to_digest(bytes([0xAB] * 32), d.ctypes.data_as(ctypes.c_void_p))
    assert int(d[0]) == a["felt_of_ab_x8"]
    orc.lib.orc_bytes_to_digest(bytes([0xCD] * 32), d.ctypes.data_as(ctypes.c_void_p))
    assert int(d[0]) == a["felt_of_cd_x8"]
    out = np.empty(64, dtype=np.uint64)
    n = orc.lib.orc_bytes_to_u64s(KATS["salt"].encode(), len(KATS["salt"]), out.ctypes.data_as(ctypes.c_void_p))
    assert n == a["salt_felts"] and out[:3].tolist() == [int.from_bytes(b"worm", "little"), int.from_bytes(b"hole", "little"), 1]
    digest = bytes.fromhex(KATS["digest_hex_head"]) + bytes(KATS["digest_zero_run"]) + bytes.fromhex(KATS["digest_hex_tail"])
    assert len(digest) == a["digest_bytes"]
    n = orc.lib.orc_bytes_to_u64s(digest, len(digest), out.ctypes.data_as(ctypes.c_void_p))
    assert n == a["digest_felts"]


def _p2_params(orc, rc_ext, rc_int, diag, m4, absorb_add=0):
    buf = np.zeros(orc.lib.orc_p2_params_size() // 8, dtype=np.uint64)
    flat = list(np.array(rc_ext, dtype=np.uint64).ravel()) + list(rc_int) + list(diag) + list(np.array(m4, dtype=np.uint64).ravel())
    buf[:len(flat)] = flat
    buf[len(flat)] = absorb_add  # int in the low half of the next word (little endian)
    return buf


def p2_address(orc, params, secret_hex):
    out = np.empty(64, dtype=np.uint64)
    n = orc.lib.orc_bytes_to_u64s(b"wormhole", 8, out.ctypes.data_as(ctypes.c_void_p))
    sec = np.empty(4, dtype=np.uint64)
    orc.lib.orc_bytes_to_digest(bytes.fromhex(secret_hex), sec.ctypes.data_as(ctypes.c_void_p))
    pre = np.concatenate([out[:n], sec])
    h1 = np.empty(4, dtype=np.uint64); h2 = np.empty(4, dtype=np.uint64)
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    orc.lib.orc_p2_hash_pad10(vp(params), vp(pre), pre.size, vp(h1))
    orc.lib.orc_p2_hash_pad10(vp(params), vp(h1), 4, vp(h2))
    b = ctypes.create_string_buffer(32)
    orc.lib.orc_digest_to_bytes(vp(h2), b)
    return b.raw.hex()


def p2_block_hash(orc, params, k):
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    digest = bytes.fromhex(KATS["digest_hex_head"]) + bytes(KATS["digest_zero_run"]) + bytes.fromhex(KATS["digest_hex_tail"])
    parent = bytes.fromhex(k["parent_hash"]) if "parent_hash" in k else bytes(k["parent_hash_bytes"])
    pre = np.zeros(45, dtype=np.uint64)
    d4 = np.empty(4, dtype=np.uint64)
    for off, b in ((0, parent), (5, bytes.fromhex(k["state_root"])), (9, bytes.fromhex(k["extrinsics_root"])), (13, bytes.fromhex(k["zk_tree_root"]))):
        orc.lib.orc_bytes_to_digest(b, vp(d4)); pre[off:off + 4] = d4
    pre[4] = k["block_number"]
    dg = np.zeros(32, dtype=np.uint64)
    assert orc.lib.orc_bytes_to_u64s(digest, len(digest), vp(dg)) == 28
    pre[17:] = dg[:28]
    h = np.empty(4, dtype=np.uint64)
    orc.lib.orc_p2_hash_pad10(vp(params), vp(pre), 45, vp(h))
    b = ctypes.create_string_buffer(32)
    orc.lib.orc_digest_to_bytes(vp(h), b)
    return b.raw


def test_poseidon2_pinned_by_all_seven_reference_kats(orc):
    """qp-poseidon-core 3.1.0's constants are not in the reference tree. The set orc_p2_qp_params derives
    (found by tools/derivation/p2_search.py) reproduces all five address vectors
    (wormhole/tests/src/circuit/unspendable_account_tests.rs:9-24) and both block-header vectors
    (wormhole/tests/test-helpers/src/lib.rs:210-219): Poseidon2, its pad-10 additive sponge and the codecs are PINNED.
    Random constants, and the pinned constants with overwrite absorption, are rejected by the same gate."""
    params = np.zeros(orc.lib.orc_p2_params_size() // 8, dtype=np.uint64)
    orc.lib.orc_p2_qp_params(params.ctypes.data_as(ctypes.c_void_p))
    assert all(p2_address(orc, params, k["secret"]) == k["address"] for k in KATS["address_kats"])
    assert all(p2_block_hash(orc, params, k) == bytes(k["expected_hash_bytes"]) for k in KATS["block_header_kats"])
    assert len(KATS["address_kats"]) == 5 and len(KATS["block_header_kats"]) == 2
    # overwrite absorption: indistinguishable on one-block inputs, wrong on the 45-element header
    over = params.copy(); over[146] = 0
    assert all(p2_address(orc, over, k["secret"]) == k["address"] for k in KATS["address_kats"])
    assert not any(p2_block_hash(orc, over, k) == bytes(k["expected_hash_bytes"]) for k in KATS["block_header_kats"])
    rng = np.random.default_rng(5)
    rnd = _p2_params(orc, rng.integers(0, P, (8, 12), dtype=np.uint64), rng.integers(0, P, 22, dtype=np.uint64),
                     rng.integers(0, P, 12, dtype=np.uint64), [[5, 7, 1, 3], [4, 6, 1, 1], [1, 3, 5, 7], [1, 1, 4, 6]])
    assert not any(p2_address(orc, rnd, k["secret"]) == k["address"] for k in KATS["address_kats"])


def test_fast_partial_tables(orc, pkg):
    """plonky2's FAST_PARTIAL_* tables: the oracle's derivation, the product's derivation and the golden fixture agree,
    and the fixture reproduces the recalled upstream anchors."""
    G = json.load(open(os.path.join(GOLD, "poseidon_fast_partial.json")))
    flat = G["first_round_constant"] + G["round_constants"] + [0]
    for row in G["vs"]: flat += row
    for row in G["w_hats"]: flat += row
    init_t = G["initial_matrix_upstream_layout"]            # upstream stores the transpose of the column-vector matrix
    for c in range(11): flat += [init_t[r][c] for r in range(11)]
    want = np.array(flat, dtype=np.uint64)
    assert np.array_equal(orc.fast_partial(), want)
    rc, fp = pkg.poseidon_constants()
    assert np.array_equal(fp, want)
    assert np.array_equal(rc, orc.round_constants())
    a = G["anchors"]
    assert [hex(x) for x in G["first_round_constant"][:2]] == a["first_round_constant"]
    assert [hex(x) for x in G["round_constants"][:2]] == a["round_constants"]
    assert [hex(x) for x in G["vs"][0][:2]] == a["vs_0"] and [hex(x) for x in G["w_hats"][0][:2]] == a["w_hats_0"]
    assert [hex(x) for x in init_t[0][:2]] == a["initial_matrix_row0"]
    assert int(G["m00"]) == 25
    assert [int(x) for x in orc.poseidon(np.array(G["check_vector"]["input"], dtype=np.uint64))] == G["check_vector"]["output"]


def test_fast_mds_layer_matches_the_definition(orc):
    """orc_poseidon_permute (32-bit-halves MDS) against the definitional circulant-plus-diagonal form."""
    import ctypes
    rng = np.random.default_rng(3)
    for _ in range(50):
        s = rng.integers(0, 0xFFFFFFFF00000001, size=12, dtype=np.uint64)
        a, b = s.copy(), s.copy()
        orc.lib.orc_poseidon_permute(a.ctypes.data_as(ctypes.c_void_p))
        orc.lib.orc_poseidon_permute_naive(b.ctypes.data_as(ctypes.c_void_p))
        assert (a == b).all()
    edge = np.array([0xFFFFFFFF00000000] * 12, dtype=np.uint64)
    a, b = edge.copy(), edge.copy()
    orc.lib.orc_poseidon_permute(a.ctypes.data_as(ctypes.c_void_p))
    orc.lib.orc_poseidon_permute_naive(b.ctypes.data_as(ctypes.c_void_p))
    assert (a == b).all()
