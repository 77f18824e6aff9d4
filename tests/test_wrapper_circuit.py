"""A wrapper circuit that checks the Merkle half of its inner proofs (include/qpgpu_batch.h: qpgpu_wrapper_circuit_build,
csrc/wrapper_circuit.cpp), on the CPU: the library builds the circuit (host code) from the leaf circuit's pack, the ORACLE makes
the inner leaf proofs, generates the wrapper's witness from them (fill_private_batch_witness's assignments + the query indices of
the host verifier's transcript replay), proves and verifies. What add_recursive_verifiers
(wormhole/aggregator/src/common/recursive.rs:74-102) adds per inner proof, as far as the commitments go; the module says what is
NOT verified in-circuit. tests/test_wrapper_circuit_gpu.py runs the same through the device."""
import numpy as np
import pytest

import leaf_cases as lc
import oracle_binding as ob


@pytest.fixture(scope="module")
def setup(pkg, orc):
    L = pkg.leaf
    leaf = L.LeafCircuit()                                  # the restated leaf circuit at its own size (2^8 rows)
    xs = [lc.real_inputs(L, depth=3), lc.test_inputs(L, 0), lc.dummy_inputs(L)]
    com = [leaf.commit(x) for x in xs]
    op = ob.OracleProver(orc, leaf.pack)
    proofs = op.commit_prove_many(com[0][0], np.stack([c[1] for c in com]), np.stack([c[2] for c in com]))
    op.close()
    ver = pkg.Verifier(leaf.pack)                           # verifier data (constants/sigmas cap) computed on the host
    assert all(ver.verify(p) for p in proofs)
    w = pkg.recursion.WrapperCircuit(leaf.pack, ver, 2)                       # transcript in-circuit (the default)
    w0 = pkg.recursion.WrapperCircuit(leaf.pack, ver, 2, transcript=False)    # the Merkle checks alone: query indices are inputs
    yield leaf, proofs, ver, w, w0
    ver.close()


def test_shape(pkg, setup):
    leaf, proofs, ver, wt, w = setup
    h = pkg.pack_header(leaf.pack)
    L_ = h["degree_bits"] + h["rate_bits"]
    path = L_ - h["cap_height"]
    widths = [h["num_selectors"] + h["num_constants"] + h["num_routed_wires"], 135, 2 * (1 + h["num_partial_products"]), 16]
    perms = lambda wd: -(-wd // 8)
    steps = h["num_arity_rounds"]
    # per query round: hash the four rows, walk four paths, hash every step's coset (2^4 extension values = 32 elements), walk its path
    per_query = sum(perms(wd) for wd in widths) + 4 * path + sum(4 + (L_ - 4 * (s + 1) - h["cap_height"]) for s in range(steps))
    pis_hash = perms(21)
    assert w.info["rows_poseidon"] == 2 * (28 * per_query + pis_hash) + perms(42)      # + the wrapper's own public-input hash
    assert w.info["rows_random_access"] == 2 * 28 * (4 + steps) * 4 // 4               # 4 look-ups per path, 4 copies per row
    assert w.info["rows_base_sum"] == 2 * 28 and w.info["public_inputs"] == 42
    assert w.info["targets_per_proof"] == (len(proofs[0]) - 28 * (4 + steps)) // 8
    # the targets nothing in the Merkle half consumes have no cell: openings (2 per extension element), pow witness, final polynomial, preimages
    nopen = 2 * (h["num_selectors"] + h["num_constants"] + 80 + 135 + 2 + 2 * h["num_partial_products"] + 16 + 2)
    fin = 2 << (h["degree_bits"] - 4 * steps)
    assert int((w.target_map == pkg.recursion.NO_CELL).sum()) == 2 * (nopen + 1 + fin + 4)
    # with the transcript in-circuit: the openings, the final polynomial and the proof-of-work witness are absorbed (they have cells now),
    # the query indices are derived (no cells), and the transcript costs one PoseidonGate row per 8 absorbed elements plus the squeezes
    assert int((wt.target_map == pkg.recursion.NO_CELL).sum()) == 2 * (4 + 28)
    absorbed = [8 + 64, 64, 64, nopen] + [64] * steps + [fin + 1]            # between two challenges: digest + pi hash + cap | cap | cap | openings | FRI caps | final poly + pow witness
    squeezes = 28 // 8                                                        # 28 indices after the proof-of-work response: the buffer of 8 refills three more times
    assert wt.info["rows_poseidon"] == w.info["rows_poseidon"] + 2 * (sum(-(-a // 8) for a in absorbed) + squeezes)
    assert wt.info["rows_base_sum"] == 2 * (28 * 2 + 1)                       # 64-bit splits of the 28 challenges (two gates each) + the proof-of-work range check


def test_valid_inner_proofs_give_a_valid_wrapper_proof(pkg, orc, setup):
    leaf, proofs, ver, w, w0 = setup
    c0 = w0.commit(proofs[:2])
    assert orc.generate_witness(w0.pack, *c0)[0] == orc.WIT_OK                 # indices from the host verifier's replay
    cells, vals, pis = w.commit(proofs[:2])
    rc, wires, _ = orc.generate_witness(w.pack, cells, vals, pis)
    assert rc == orc.WIT_OK
    assert pis.tolist() == np.concatenate([lc.proof_public_inputs(p, 21) for p in proofs[:2]]).tolist()     # inner public inputs forwarded
    oc = ob.OracleCircuit(orc, w.pack)
    proof = oc.prove(wires, pis)
    assert oc.verify(proof) == 0
    oc.close()
    # other inner proofs, same wrapper
    cells2, vals2, pis2 = w.commit([proofs[2], proofs[0]])
    assert np.array_equal(cells2, cells) and orc.generate_witness(w.pack, cells2, vals2, pis2)[0] == orc.WIT_OK


def test_tampered_inner_proofs_are_unsatisfiable(pkg, orc, setup):
    leaf, proofs, ver, w, w0 = setup
    h = pkg.pack_header(leaf.pack)
    base = proofs[1]
    # byte offsets inside the proof: caps 3 x 16 x 32, openings, commit caps, then the query rounds
    n_open = (h["num_selectors"] + h["num_constants"] + 80 + 135 + 2 + 2 + 2 * h["num_partial_products"] + 16) * 16
    q0 = 3 * 16 * 32 + n_open + h["num_arity_rounds"] * 16 * 32
    ncs = h["num_selectors"] + h["num_constants"] + 80
    path = h["degree_bits"] + h["rate_bits"] - h["cap_height"]
    spots = {"constants/sigmas row, first query": q0 + 8 * 3,
             "its first sibling": q0 + 8 * ncs + 1 + 5,
             "wires row, first query": q0 + 8 * ncs + 1 + 32 * path + 8 * 100,
             "a later query round": q0 + 9 * ((len(base) - q0 - 8 * (2 << (h["degree_bits"] - 4 * h["num_arity_rounds"])) - 8 - 8 * 21) // 28) + 40}
    for what, off in spots.items():
        bad = bytearray(base); bad[off] ^= 1
        # the transcript does not absorb query data: the indices are those of the honest proof, the host verifier rejects the proof
        assert not ver.verify(bytes(bad)), what
        for wr in (w, w0):
            cells, vals, pis = wr.commit([proofs[0], bytes(bad)])
            rc, _, cell = orc.generate_witness(wr.pack, cells, vals, pis)
            assert rc == orc.WIT_CONFLICT, what
    # a wrong query index (indices as inputs): the rows are committed, but not at that leaf
    qi = [w0.query_indices(p) for p in proofs[:2]]
    qi[0] = qi[0].copy(); qi[0][5] ^= 1
    cells, vals, pis = w0.commit(proofs[:2], query_indices=qi)
    assert orc.generate_witness(w0.pack, cells, vals, pis)[0] == orc.WIT_CONFLICT
    # with the transcript in-circuit, everything the transcript absorbs is bound too: a cap, an opening, the final polynomial, the
    # proof-of-work witness — the indices (or the proof-of-work response) change and the paths no longer close
    final_poly_at = len(base) - 8 * 21 - 8 - 8 * (2 << (h["degree_bits"] - 4 * h["num_arity_rounds"]))
    for what, off in (("a cap", 40), ("an opening", 3 * 16 * 32 + 8 * 50 + 2), ("the final polynomial", final_poly_at + 9), ("the proof-of-work witness", len(base) - 8 * 21 - 8)):
        bad = bytearray(base); bad[off] ^= 1
        cells, vals, pis = w.commit([proofs[0], bytes(bad)])
        assert orc.generate_witness(w.pack, cells, vals, pis)[0] == orc.WIT_CONFLICT, what
    # (without the transcript an opening is not looked at by the Merkle half)
    bad = bytearray(base); bad[3 * 16 * 32 + 8 * 50 + 2] ^= 1
    c0 = w0.commit([proofs[0], bytes(bad)], query_indices=[w0.query_indices(proofs[0]), w0.query_indices(base)])
    assert orc.generate_witness(w0.pack, *c0)[0] == orc.WIT_OK
    # a proof of the wrong shape is refused before any assignment, with the reference's message (common/utils.rs:295-317)
    with pytest.raises(ValueError) as e:
        w.commit([proofs[0], base[:-8]])
    assert "malformed" in str(e.value)


def test_full_verification_in_circuit(pkg, orc, setup):
    """QPGPU_WRAPPER_VERIFY: the arithmetic half of verify_proof in-circuit too (csrc/verify_math.hpp instantiated over the
    builder). The case the Merkle half and the transcript cannot see: a proof made HONESTLY by the prover from a trace that does
    not satisfy the inner circuit — its rows are committed under its caps, its transcript is its own, its proof of work is valid;
    only the quotient identity at zeta / the FRI consistency fail. The host verifier rejects it; the wrapper without the flag has a
    witness for it; the wrapper with the flag has none."""
    leaf, proofs, ver, w, w0 = setup
    L = pkg.leaf
    wv = pkg.recursion.WrapperCircuit(leaf.pack, ver, 2, verify=True)
    assert wv.info["degree_bits"] == 13 and wv.info["rows_poseidon"] == w.info["rows_poseidon"]
    c = wv.commit(proofs[:2])
    rc, wires, _ = orc.generate_witness(wv.pack, *c)
    assert rc == orc.WIT_OK
    ocw = ob.OracleCircuit(orc, wv.pack)
    proof = ocw.prove(wires, c[2])
    assert ocw.verify(proof) == 0
    ocw.close()
    # traces that violate the leaf circuit: a wire of a used gate slot changed (gate constraint or copy constraint broken); and one
    # that does not: a wire no gate of that row reads
    x = lc.real_inputs(L, depth=3)
    cells, vals, pis = leaf.commit(x)
    rc, lw, _ = orc.generate_witness(leaf.pack, cells, vals, pis)
    assert rc == orc.WIT_OK
    oc = ob.OracleCircuit(orc, leaf.pack)
    for (col, row), valid in (((3, 0), False), ((20, 40), False), ((100, 17), True)):
        bw = lw.copy(); bw[col, row] = (int(bw[col, row]) + 1) % pkg.P
        forged = oc.prove(bw, pis)
        assert bool(ver.verify(forged)) == valid, (col, row)
        c0 = w.commit([proofs[1], forged])
        assert orc.generate_witness(w.pack, *c0)[0] == orc.WIT_OK, (col, row)            # Merkle half + transcript: satisfied
        cv = wv.commit([proofs[1], forged])
        assert orc.generate_witness(wv.pack, *cv)[0] == (orc.WIT_OK if valid else orc.WIT_CONFLICT), (col, row)
    oc.close()
    # the flag needs the in-circuit transcript
    with pytest.raises(pkg.QpGpuError) as e:
        pkg.recursion.WrapperCircuit(leaf.pack, ver, 2, transcript=False, verify=True)
    assert "needs the in-circuit transcript" in str(e.value)


def test_recursion_under_the_poseidon2_hasher(pkg, orc):
    """Were the fork's proof-system hasher Poseidon2, every hash of the recursive verifier — public-input
    hash, transcript, Merkle paths — would be rows of the Poseidon2 gate (with its swap wire) instead of PoseidonGate rows, and the
    inner circuit's own Poseidon2 rows are what the vanishing polynomial evaluates. The same builder code under inner_hasher = 1,
    proof system (oracle and host verifier) switched to Poseidon2 with qp-poseidon-core's parameters."""
    L = pkg.leaf
    qp = pkg.poseidon2_qp_params()
    pkg.set_hasher_poseidon2(*qp); orc.select_poseidon2(*qp)
    try:
        leaf = L.LeafCircuit(inner_hasher=1)
        xs = lc.shared_tree_inputs(L, 2) + [lc.dummy_inputs(L)]
        com = [leaf.commit(x) for x in xs]
        op = ob.OracleProver(orc, leaf.pack)
        proofs = op.commit_prove_many(com[0][0], np.stack([c[1] for c in com]), np.stack([c[2] for c in com]))
        op.close()
        ver = pkg.Verifier(leaf.pack, hasher=1)
        assert all(ver.verify(p) for p in proofs)
        w = pkg.recursion.WrapperCircuit(leaf.pack, ver, 2, inner_hasher=1, logic="private_batch", verify=True)
        assert w.info["rows_poseidon"] == 0 and w.info["degree_bits"] == 13
        pre = np.arange(8, dtype=np.uint64).reshape(2, 4)
        c = w.commit([proofs[0], proofs[2]], preimages=pre)
        rc, wires, _ = orc.generate_witness(w.pack, *c)
        assert rc == orc.WIT_OK
        oc = ob.OracleCircuit(orc, w.pack)
        proof = oc.prove(wires, c[2])
        assert oc.verify(proof) == 0
        oc.close()
        wv = pkg.Verifier(w.pack, hasher=1)
        assert wv.verify(proof)
        wv.close()
        bad = bytearray(proofs[0]); bad[len(bad) // 2] ^= 1
        c = w.commit([bytes(bad), proofs[2]], preimages=pre, public_inputs=c[2])
        assert orc.generate_witness(w.pack, *c)[0] == orc.WIT_CONFLICT
        ver.close()
    finally:
        pkg.set_hasher_poseidon(); orc.select_poseidon()


def test_wire_budget_of_the_wrapper(pkg):
    """The config policy of the wrapper builder (the reference's new_rejects_pathological_circuit_configs, circuit_logic.rs:2015-2040,
    as far as the ABI exposes the config): a routed-wire count the gates of the recursive verifier cannot live in is refused with
    the reason, before any expensive construction."""
    L = pkg.leaf
    fake = L.LeafCircuit(fragment=L.FRAGMENT_FAKE_LEAF)
    ver = pkg.Verifier(fake.pack)
    for routed, needle in ((8, "num_routed_wires outside"), (136, "num_routed_wires outside"), (40, "poseidon_mds_ext: too few routed wires")):
        with pytest.raises(pkg.QpGpuError) as e:
            pkg.recursion.WrapperCircuit(fake.pack, ver, 1, num_routed_wires=routed, verify=True)
        assert needle in str(e.value), routed
    for routed in (48, 60, 80):
        assert pkg.recursion.WrapperCircuit(fake.pack, ver, 1, num_routed_wires=routed, verify=True).info["degree_bits"] == 12
    ver.close()
