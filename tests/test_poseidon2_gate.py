"""The qp fork's Poseidon2 GATE (gate type 14 of the circuit pack) without a GPU: pack format, the oracle's prover and verifier,
the library's host verifier, and the tie between the gate and the KAT-pinned hash.

Every hash of the Wormhole leaf circuit goes through `hash_n_to_hash_no_pad_p2` (reference
wormhole/circuit/src/zk_merkle_proof.rs:482,504,606, nullifier.rs:298-299, unspendable_account.rs:229-231,
block_header/mod.rs:66). The permutation behind it is pinned by the reference's seven known-answer vectors; the gate's wire
layout lives in un-vendored qp-plonky2 and is carried as data in the pack ("P2GL1") — LAYOUT UNPINNED: these tests run the
default layout (upstream PoseidonGate's, carried over) and a deliberately different one through the same code."""
import ctypes

import numpy as np
import pytest

from oracle_binding import OracleCircuit

P = 0xFFFFFFFF00000001
KW = dict(poseidon=True, base_sum=True, poseidon2=True)


def host_hash(pkg, pre):
    lib = pkg.load_library()
    x = np.ascontiguousarray(pre, dtype=np.uint64); out = np.empty(4, dtype=np.uint64)
    assert lib.qpgpu_poseidon2_hash_pad10(None, 0, x.ctypes.data if x.size else None, x.size, out.ctypes.data) == 0
    return out


def check_sites(pkg, pack, wires, d, npis=21, alt=False):
    """every hash site's in-circuit digest == the library's pad-10 sponge over the same preimage cells"""
    sites = pkg.synth_p2_sites(d, npis, poseidon2=True, p2_alt_layout=alt)
    assert sites
    for site in sites:
        pre, dig = pkg.p2_site_cells(pack, site)
        want = host_hash(pkg, [wires[c, r] for c, r in pre])
        got = [int(wires[c, r]) for c, r in dig]
        assert got == [int(x) for x in want], site
    return sites


@pytest.mark.parametrize("alt", [False, True])
def test_gate_rows_compute_the_kat_pinned_hash(pkg, alt):
    """The digests the Poseidon2-gate rows of a synthetic leaf-profile circuit produce are qpgpu_poseidon2_hash_pad10 of their
    preimages — the function tests/test_leaf_witness.py holds to the reference's seven vectors — for one-block (7, 4, 8 + pad),
    two-block (9) and six-block (45) preimages."""
    d = 8
    pack, wires, _ = pkg.synth_circuit(d, seed=11, p2_alt_layout=alt, **KW)
    sites = check_sites(pkg, pack, wires, d, alt=alt)
    assert [s[0] for s in sites[:6]] == [7, 4, 9, 4, 8, 45] and [s[1] for s in sites[:6]] == [1, 1, 2, 1, 2, 6]
    lay = pkg.pack_p2_layout(pack)
    assert lay is not None and (lay["w_swap"] == pkg.P2_NO_SWAP) == alt
    hdr = pkg.pack_header(pack)
    assert hdr["num_gate_constraints"] == 123     # PoseidonGate's 123 and the Poseidon2 gate's 123 (default) / 118 (no swap)


def test_the_bench_shape_holds_all_61_permutations(pkg):
    """2^13 rows: the whole list of the leaf circuit's application hashes fits — 7, 4 | 9, 4 | 8 | 45 | 16 x 16 (SURVEY.md
    Appendix B: 1+1+2+1+2+6+48 = 61 permutations)."""
    sites = pkg.synth_p2_sites(13, 21, poseidon2=True)
    assert [s[0] for s in sites] == [7, 4, 9, 4, 8, 45] + [16] * 16
    assert sum(s[1] for s in sites) == 61


@pytest.mark.parametrize("alt", [False, True])
def test_oracle_and_library_verifiers_agree_on_poseidon2_rows(pkg, orc, alt):
    """orc_prove -> orc_verify and the library's own host verifier (written separately) accept; a change to any class of the
    gate's wires (input, output, swap, delta, each block of recorded S-box inputs) makes both reject."""
    d = 7
    pack, wires, pis = pkg.synth_circuit(d, seed=12, p2_alt_layout=alt, **KW)
    lay = pkg.pack_p2_layout(pack)
    oc = OracleCircuit(orc, pack); ver = pkg.Verifier(pack)
    try:
        proof = oc.prove(wires, pis)
        assert oc.verify(proof) == 0 and ver.verify(proof)
        row = 8 * pkg.synth_p2_sites(d, 21, poseidon2=True)[2][2] + 3        # first gate row of the two-block hash
        cols = [lay["w_input"] + 9, lay["w_output"] + 5, lay["w_full0"] + 13, lay["w_partial"] + 21, lay["w_full1"] + 47, lay["w_partial"]]
        if lay["w_swap"] != pkg.P2_NO_SWAP:
            cols += [lay["w_swap"], lay["w_delta"] + 2]
        for col in cols:
            bad = wires.copy()
            bad[col, row] = (int(bad[col, row]) + 1) % P
            pf = oc.prove(bad, pis)
            assert oc.verify(pf) != 0 and not ver.verify(pf), col
    finally:
        oc.close(); ver.close()


def test_layout_trailer_round_trip_and_validation(pkg):
    lib = pkg.load_library()
    lib.qpgpu_pack_validate.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p]
    def validate(p):
        p = np.ascontiguousarray(p, dtype=np.uint64)
        err = ctypes.create_string_buffer(256)
        return lib.qpgpu_pack_validate(p.ctypes.data, p.size, err), err.value.decode()
    pack, _, _ = pkg.synth_circuit(6, seed=13, **KW)
    assert validate(pack)[0] == 0
    trailers = pkg.pack_trailers(pack)
    magic, at, cnt = [t for t in trailers if t[0] == 0x314C473250][0]
    assert cnt == 10 and pack[at:at + 10].tolist() == [0, 12, 24, 25, 29, 65, 87, 0, 0, 135]
    # without the trailer the pack is refused (fail closed): the fork's layout is not known offline, so none is assumed
    cut = np.concatenate([pack[:at - 2], pack[at + cnt:]])
    rc, msg = validate(cut)
    assert rc != 0 and "P2GL1" in msg and pkg.pack_p2_layout(cut) is None
    for field, value, why in ((0, 130, "outside"), (1, 70, "routed"), (4, 60, "overlap"), (7, 2, "first_round_wires"), (8, 1, "constraint order"),
                              (9, 200, "end_wire"), (7, 1, "overlap"), (2, pkg.P2_NO_SWAP, "constraint count")):
        bad = pack.copy(); bad[at + field] = value
        rc, msg = validate(bad)
        assert rc != 0 and why in msg, (field, msg)
    short = pack.copy(); short[at - 1] = 9
    assert validate(short[:-1])[0] != 0
    twice = np.concatenate([pack, pack[at - 2:at + cnt]])
    assert validate(twice)[0] != 0
