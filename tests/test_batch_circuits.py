"""The two batch layers' own circuits on the CPU: build_private_batch_constraints / build_public_batch_constraints
(wormhole/aggregator/src/private_batch/circuit/circuit_logic.rs:171-477, public_batch/circuit/circuit_logic.rs:167-317) restated on
the native builder on top of the wrapper circuit's proof targets (csrc/wrapper_circuit.cpp, QPGPU_WRAPPER_PRIVATE_BATCH /
_PUBLIC_BATCH). The library builds the circuits (host code); the ORACLE makes the inner proofs, generates the witnesses, proves
and verifies. Parity: the public inputs the circuit computes are the ones the host restatement of the same logic
(qpgpu_private_batch_outputs / qpgpu_public_batch_outputs, pinned message for message in tests/test_batch_host.py) predicts — the
witness is unsatisfiable otherwise — and they parse with the reference's parsers' restatements. The cases follow the reference's
own circuit tests (circuit_logic.rs tests: grouping and dedup of exit accounts, dummy masking, replayed leaf, mixed blocks,
nullifiers emitted sorted)."""
import numpy as np
import pytest

import leaf_cases as lc
import oracle_binding as ob

E1, E2, E3 = bytes([4] * 32), bytes([7] * 32), bytes([9] * 32)


@pytest.fixture(scope="module")
def setup(pkg, orc):
    L = pkg.leaf
    leaf = L.LeafCircuit()
    # three real spends of one block: 0 and 1 pay the same first exit account (grouped), 2 pays E3 twice (deduplicated inside one
    # proof); one real spend of ANOTHER block; the reference's dummy
    xs = lc.shared_tree_inputs(L, 3, exits=[(E1, E2), (E1, E3), (E3, E3)], outputs=[(200, 97), (150, 10), (5, 6)])
    xs += [lc.real_inputs(L, depth=2), lc.dummy_inputs(L)]
    com = [leaf.commit(x) for x in xs]
    op = ob.OracleProver(orc, leaf.pack)
    proofs = op.commit_prove_many(com[0][0], np.stack([c[1] for c in com]), np.stack([c[2] for c in com]))
    op.close()
    ver = pkg.Verifier(leaf.pack)
    w = pkg.recursion.WrapperCircuit(leaf.pack, ver, 3, logic="private_batch")
    yield leaf, xs, proofs, ver, w
    ver.close()


def pre(seed, n=3):
    return np.random.default_rng(seed).integers(0, 1 << 63, (n, 4), dtype=np.uint64)


def test_private_batch_layer(pkg, orc, setup):
    leaf, xs, proofs, ver, w = setup
    A = pkg.aggregation
    assert w.info["public_inputs"] == 21 * 3 + 8
    real0, real1, real2, other, dummy = proofs
    p = pre(1)
    cells, vals, pis = w.commit([real1, dummy, real0], preimages=p)        # real and dummy slots in any order
    rc, wires, _ = orc.generate_witness(w.pack, cells, vals, pis)
    assert rc == orc.WIT_OK                                                 # the circuit computes what the host restatement predicts
    rows = np.stack([lc.proof_public_inputs(q, 21) for q in (real1, dummy, real0)])
    assert pis.tolist() == A.private_batch_outputs(rows, p).tolist()
    hdr, slots, nulls = A.parse_private_batch_public_inputs(pis)
    assert hdr["num_exit_slots"] == 6 and hdr["asset_id"] == 0 and hdr["volume_fee_bps"] == lc.DEFAULT_VOLUME_FEE_BPS and hdr["n_leaf"] == 3
    assert hdr["block_hash"] == bytes(xs[0].block_hash) and hdr["block_number"] == xs[0].block_number
    # slots 0/1 = real1 (E1: 150 + 200 grouped, E3: 10), 2/3 = the dummy (masked), 4/5 = real0 (E1 again: zeroed as a duplicate, E2: 97)
    assert [(s, a) for s, a in slots] == [(350, E1), (10, E3), (0, bytes(32)), (0, bytes(32)), (0, bytes(32)), (97, E2)]
    dummy_null = A.dummy_nullifier(p[1])
    assert sorted(nulls) == sorted([bytes(xs[1].nullifier), bytes(xs[0].nullifier), dummy_null])
    key = lambda d: [int.from_bytes(d[8 * i:8 * i + 8], "little") for i in range(4)]
    assert [key(d) for d in nulls] == sorted(key(d) for d in nulls)         # canonical sorted order, limb 0 most significant
    oc = ob.OracleCircuit(orc, w.pack)
    proof = oc.prove(wires, pis)
    assert oc.verify(proof) == 0
    oc.close()
    # two outputs of ONE proof to one account: deduplicated the same way
    cells, vals, pis = w.commit([real2, dummy, dummy], preimages=p)
    assert orc.generate_witness(w.pack, cells, vals, pis)[0] == orc.WIT_OK
    assert A.parse_private_batch_public_inputs(pis)[1][:2] == [(11, E3), (0, bytes(32))]


def test_private_batch_layer_refuses(pkg, orc, setup):
    leaf, xs, proofs, ver, w = setup
    real0, real1, real2, other, dummy = proofs
    p = pre(2)
    good = w.commit([real0, real1, dummy], preimages=p)[2]
    # the same leaf in two real slots (the replay the nullifier-distinctness constraint exists for), real slots of two blocks: the
    # host restatement refuses with the circuit's reason, and the circuit itself has no witness even under public inputs of a valid batch
    for slots, needle in (([real0, real0, dummy], "nullifier"), ([real0, other, dummy], "block")):
        with pytest.raises(pkg.QpGpuError) as e:
            w.commit(slots, preimages=p)
        assert e.value.code == -4 and needle in str(e.value)
        cells, vals, pis = w.commit(slots, preimages=p, public_inputs=good)
        assert orc.generate_witness(w.pack, cells, vals, pis)[0] == orc.WIT_CONFLICT
    # public inputs that are not the circuit's: one sum, one nullifier limb, the padding
    for at in (8, 8 + 6 * 5 + 2, 21 * 3 + 7):
        bad = good.copy(); bad[at] += 1
        cells, vals, pis = w.commit([real0, real1, dummy], preimages=p, public_inputs=bad)
        assert orc.generate_witness(w.pack, cells, vals, pis)[0] == orc.WIT_CONFLICT
    # another preimage than the one the public inputs were computed from
    cells, vals, pis = w.commit([real0, real1, dummy], preimages=p + 1, public_inputs=good)
    assert orc.generate_witness(w.pack, cells, vals, pis)[0] == orc.WIT_CONFLICT
    # a tampered inner proof: the Merkle half and the transcript are still underneath
    bad = bytearray(real1); bad[len(bad) // 2] ^= 1
    cells, vals, pis = w.commit([real0, bytes(bad), dummy], preimages=p)
    assert orc.generate_witness(w.pack, cells, vals, pis)[0] == orc.WIT_CONFLICT


def test_public_batch_layer(pkg, orc, setup):
    leaf, xs, proofs, ver, w = setup
    A = pkg.aggregation
    real0, real1, real2, other, dummy = proofs
    oc = ob.OracleCircuit(orc, w.pack)
    inner = []
    for slots, seed in (([dummy, real2, real0], 3), ([dummy, dummy, dummy], 4), ([other, dummy, dummy], 5)):     # a batch with spends, an all-dummy batch, a batch of another block
        cells, vals, pis = w.commit(slots, preimages=pre(seed))
        rc, wires, _ = orc.generate_witness(w.pack, cells, vals, pis)
        assert rc == orc.WIT_OK
        inner.append(oc.prove(wires, pis))
    oc.close()
    v1 = pkg.Verifier(w.pack)
    assert all(v1.verify(q) for q in inner)
    w2 = pkg.recursion.WrapperCircuit(w.pack, v1, 2, logic="public_batch")
    assert w2.info["public_inputs"] == A.public_batch_pi_len(2, 3) == 12 + 2 * 6 * 5 + 2 * 3 * 4
    addr = bytes(range(1, 33)); addr = bytes(b & 0x7F if i % 8 == 7 else b for i, b in enumerate(addr))
    cells, vals, pis = w2.commit(inner[:2], aggregator_address=addr)
    rc, wires, _ = orc.generate_witness(w2.pack, cells, vals, pis)
    assert rc == orc.WIT_OK
    rows = np.stack([lc.proof_public_inputs(q, 71) for q in inner[:2]])
    assert pis.tolist() == A.public_batch_outputs(rows, 3, addr).tolist()
    hdr, slots, nulls = A.parse_public_batch_public_inputs(pis, 2, 3)
    assert hdr["aggregator_address"] == addr and hdr["block_hash"] == bytes(xs[0].block_hash) and hdr["total_exit_slots"] == 12
    assert slots[:6] == A.parse_private_batch_public_inputs(rows[0])[1] and slots[6:] == [(0, bytes(32))] * 6     # forwarded in order; the dummy batch's segment zeroed
    assert nulls[3:] == [bytes(32)] * 3 and bytes(xs[0].nullifier) in nulls[:3]
    oc2 = ob.OracleCircuit(orc, w2.pack)
    proof = oc2.prove(wires, pis)
    assert oc2.verify(proof) == 0
    oc2.close()
    # batches of two blocks: refused by the host restatement, no witness in the circuit; another aggregator address than the public one: no witness
    with pytest.raises(pkg.QpGpuError) as e:
        w2.commit([inner[0], inner[2]], aggregator_address=addr)
    assert e.value.code == -4 and "block" in str(e.value)
    c = w2.commit([inner[0], inner[2]], aggregator_address=addr, public_inputs=pis)
    assert orc.generate_witness(w2.pack, *c)[0] == orc.WIT_CONFLICT
    bad = pis.copy(); bad[0] ^= 1
    c = w2.commit(inner[:2], aggregator_address=addr, public_inputs=bad)
    assert orc.generate_witness(w2.pack, *c)[0] == orc.WIT_CONFLICT
    v1.close()


def test_shape_checks(pkg, setup):
    """PrivateBatchCircuit::new / PublicBatchCircuit::new refuse an inner circuit of the wrong public-input shape (circuit_logic.rs:
    94-104 / public_batch circuit_logic.rs:74-87)."""
    leaf, xs, proofs, ver, w = setup
    R = pkg.recursion
    with pytest.raises(pkg.QpGpuError) as e:
        R.WrapperCircuit(leaf.pack, ver, 2, logic="public_batch")
    assert "private_batch_common.num_public_inputs (21)" in str(e.value)
    v1 = pkg.Verifier(w.pack)
    with pytest.raises(pkg.QpGpuError) as e:
        R.WrapperCircuit(w.pack, v1, 2, logic="private_batch")
    assert "leaf_common.num_public_inputs (71) != expected wormhole leaf PI len (21)" in str(e.value)
    v1.close()


def test_zero_knowledge_private_layer(pkg, orc, setup):
    """The private layer as the reference configures it (wormhole_private_batch_circuit_config, common/src/circuit.rs:396-402:
    zero-knowledge, 60 routed wires): CircuitBuilder::blind's rows (upstream's counts: per opening of a regular polynomial one
    NoopGate row of random wires, per opening of Z a copy-constrained pair), fresh random wires per proof, salted Merkle leaves
    in the proof; the public layer verifies such a proof completely in-circuit (the salts are hashed, not opened)."""
    leaf, xs, proofs, ver, w = setup
    real0, real1, real2, other, dummy = proofs
    wz = pkg.recursion.WrapperCircuit(leaf.pack, ver, 2, num_routed_wires=60, logic="private_batch", verify=True, zero_knowledge=True)
    h = pkg.pack_header(wz.pack)
    assert h["zero_knowledge"] == 1 and h["num_routed_wires"] == 60
    # CircuitBuilder::blinding_counts, restated: the smallest degree estimate 2^k >= the gate count that also holds its own blinding rows;
    # per estimate: arity-16 reductions while more than 2^5 coefficients remain, 28 queries, D = 2
    gates = wz.info["rows_before_padding"]
    k = max(5, (gates - 1).bit_length())
    while True:
        rounds, d = 0, k
        while d > 5:
            rounds, d = rounds + 1, d - 4
        fri_openings = 28 * (1 + 2 * rounds * 15 + 2 * (1 << d))
        regular, zs = 2 + fri_openings, 4 + fri_openings
        if gates + regular + 2 * zs <= 1 << k:
            break
        k += 1
    assert wz.info["rows_blinding"] == regular + 2 * zs and h["degree_bits"] == (gates + regular + 2 * zs - 1).bit_length()
    assert wz.blinding_cells.size == regular * 135 + zs * 60
    p = pre(9, 2)
    ca = wz.commit([real0, dummy], preimages=p, blinding_seed=bytes([1] * 32))
    cb = wz.commit([real0, dummy], preimages=p, blinding_seed=bytes([2] * 32))
    cc = wz.commit([real0, dummy], preimages=p)                                     # operating-system entropy
    assert np.array_equal(ca[0], cb[0]) and ca[2].tolist() == cb[2].tolist()
    nb = wz.blinding_cells.size
    assert not np.array_equal(ca[1][-nb:], cb[1][-nb:]) and not np.array_equal(ca[1][-nb:], cc[1][-nb:]) and np.array_equal(ca[1][:-nb], cb[1][:-nb])
    assert int(ca[1][-nb:].max()) < pkg.P
    rc, wires, _ = orc.generate_witness(wz.pack, *ca)
    assert rc == orc.WIT_OK
    # the pair rows of the Z blinding carry the same values
    first_pair = wz.info["rows_before_padding"] + regular
    assert np.array_equal(wires[:60, first_pair], wires[:60, first_pair + 1]) and not np.array_equal(wires[:60, first_pair], wires[:60, first_pair + 2])
    oc = ob.OracleCircuit(orc, wz.pack)
    proof = oc.prove(wires, ca[2])
    assert oc.verify(proof) == 0
    oc.close()
    vz = pkg.Verifier(wz.pack)
    assert vz.verify(proof)
    w2 = pkg.recursion.WrapperCircuit(wz.pack, vz, 1, logic="public_batch", verify=True)
    c2 = w2.commit([proof], aggregator_address=bytes(32))
    assert orc.generate_witness(w2.pack, *c2)[0] == orc.WIT_OK
    bad = bytearray(proof); bad[len(bad) // 3] ^= 1
    c2 = w2.commit([bytes(bad)], aggregator_address=bytes(32), public_inputs=c2[2])
    assert orc.generate_witness(w2.pack, *c2)[0] == orc.WIT_CONFLICT
    vz.close()
