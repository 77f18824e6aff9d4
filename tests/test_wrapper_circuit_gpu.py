"""The wrapper circuit that checks the Merkle half of its inner proofs, on the device: leaf proofs made by the device prover from
CircuitInputs, the wrapper's witness generated on the device from them (stage s1: 24 664 assignments -> 4 032 PoseidonGate rows,
280 RandomAccessGate rows; the inner proofs' transcripts replayed in-circuit), the wrapper proof byte-equal to the oracle's and accepted by both verifiers; a byte flipped in an
inner proof's opened row or path -> QPGPU_EUNSAT naming the target. See tests/test_wrapper_circuit.py for what is and is not
verified in-circuit (csrc/wrapper_circuit.cpp)."""
import numpy as np
import pytest

import leaf_cases as lc
import oracle_binding as ob

pytestmark = pytest.mark.gpu


def test_leaf_proofs_verified_in_a_wrapper(pkg, gpu, orc):
    L = pkg.leaf
    leaf = L.LeafCircuit()
    lp = L.LeafProver(pkg, gpu, leaf)
    xs = [lc.real_inputs(L, depth=5, seed=3), lc.test_inputs(L, 1), lc.dummy_inputs(L)]
    proofs = [lp.prove(x)[0] for x in xs]
    ver = pkg.Verifier(leaf.pack, circuit=lp.circ)           # verifier data from the GPU handle's commitment
    assert all(ver.verify(p) for p in proofs)
    w = pkg.recursion.WrapperCircuit(leaf.pack, ver, 2)
    assert w.info["degree_bits"] == 13 and w.info["rows_poseidon"] == 4032
    wc = pkg.Circuit(gpu, w.pack)
    nw, n = 135, 1 << w.info["degree_bits"]
    d = gpu.alloc(nw * n * 8)
    cells, vals, pis = w.commit(proofs[:2])
    wc.generate_witness_partial_dev(cells, vals, pis, d)
    rc, want_wires, _ = orc.generate_witness(w.pack, cells, vals, pis)
    assert rc == orc.WIT_OK and np.array_equal(d.download().reshape(nw, n), want_wires)
    wc.set_witness_check(True)
    proof = wc.prove_dev(d, pis)
    oc = ob.OracleCircuit(orc, w.pack)
    assert proof == oc.prove(want_wires, pis) and oc.verify(proof) == 0
    wv = pkg.Verifier(w.pack, circuit=wc)
    assert wv.verify(proof)
    assert lc.proof_public_inputs(proof, 42).tolist() == np.concatenate([lc.proof_public_inputs(p, 21) for p in proofs[:2]]).tolist()
    # other inner proofs through the same handle (the prepared assignment list is reused)
    cells2, vals2, pis2 = w.commit([proofs[2], proofs[1]])
    wc.generate_witness_partial_dev(cells2, vals2, pis2, d)
    assert wv.verify(wc.prove_dev(d, pis2))
    # tampering with an inner proof: the host verifier rejects it, and the wrapper's witness cannot be generated
    h = pkg.pack_header(leaf.pack)
    n_open = (h["num_selectors"] + h["num_constants"] + 80 + 135 + 2 + 2 + 2 * h["num_partial_products"] + 16) * 16
    q0 = 3 * 16 * 32 + n_open + h["num_arity_rounds"] * 16 * 32
    for off in (q0 + 24, q0 + 8 * (h["num_selectors"] + h["num_constants"] + 80) + 6, len(proofs[1]) - 8 * 21 - 8 - 8 * 2 * 16 - 300):
        bad = bytearray(proofs[1]); bad[off] ^= 1
        assert not ver.verify(bytes(bad))
        c3, v3, p3 = w.commit([proofs[0], bytes(bad)])
        with pytest.raises(pkg.QpGpuError) as e:
            wc.generate_witness_partial_dev(c3, v3, p3, d)
        assert e.value.code == -4 and "set twice with different values" in str(e.value), off
    wv.close(); wc.close(); ver.close(); lp.close(); oc.close()
    d.free(scrub=True)


def test_two_level_tree_attests_its_leaves(pkg, gpu, orc):
    """4 leaves -> 2 first-level wrappers of 2 -> 1 second-level wrapper of 2 (the 64-leaf shape of BASELINE configs[4] at test
    size): every proof accepted by the library's verifier and by the oracle's, the root's public inputs are the four leaves' in
    order, and a root built over a tampered first-level proof cannot be generated."""
    L = pkg.leaf
    tree = pkg.recursion.AttestingTree(pkg, gpu, per_batch=2, batches=2, batch_logic=False)
    xs = [lc.real_inputs(L, depth=2 + i, seed=40 + i, secret_index=i % 2) for i in range(4)]
    leaves, level1, root = tree.run(xs)
    assert all(tree.leaf_ver.verify(p) for p in leaves) and all(tree.w1_ver.verify(p) for p in level1) and tree.w2_ver.verify(root)
    for pack, proof in ((tree.w1.pack, level1[1]), (tree.w2.pack, root)):
        oc = ob.OracleCircuit(orc, pack)
        assert oc.verify(proof) == 0
        oc.close()
    want = np.concatenate([lc.proof_public_inputs(p, 21) for p in leaves])
    assert lc.proof_public_inputs(root, 84).tolist() == want.tolist()
    assert lc.proof_public_inputs(level1[1], 42).tolist() == want[42:].tolist()
    bad = bytearray(level1[0]); bad[len(bad) // 2] ^= 1
    c2 = tree.w2.commit([bytes(bad), level1[1]])
    st = tree.w2_circ.generate_witness_partial_batch_dev(c2[0], c2[1][None], c2[2][None], tree.d_wires)
    assert st == [-4] and "set twice with different values" in gpu.last_error()
    tree.close()


def test_two_level_tree_with_the_batch_layers_logic(pkg, gpu, orc):
    """The same tree with each level's OWN constraints (private-batch logic over the leaves, public-batch logic over the first
    level; tests/test_batch_circuits.py on the CPU): witnesses generated and proofs made on the device; the root's public inputs
    are the PublicBatchPublicInputs the host restatement predicts, the oracle's verifier accepts every level, and leaves the
    layers do not accept (a replayed spend; spends of two blocks) have no witness on the device either."""
    L = pkg.leaf
    A = pkg.aggregation
    addr = bytes([5] * 32)
    tree = pkg.recursion.AttestingTree(pkg, gpu, per_batch=2, batches=2, aggregator_address=addr, zero_knowledge=True)     # first level as the reference's private layer
    assert tree.w1.info["rows_blinding"] > 0 and pkg.pack_header(tree.w1.pack)["zero_knowledge"] == 1
    e1, e2 = bytes([4] * 32), bytes([7] * 32)
    sp = lc.shared_tree_inputs(L, 3, exits=[(e1, e2), (e1, e1), (e2, e1)], outputs=[(200, 97), (1, 2), (30, 40)])
    xs = [sp[0], lc.dummy_inputs(L), sp[1], sp[2]]
    leaves, level1, root = tree.run(xs)
    assert all(tree.leaf_ver.verify(p) for p in leaves) and all(tree.w1_ver.verify(p) for p in level1) and tree.w2_ver.verify(root)
    for pack, proof in ((tree.w1.pack, level1[0]), (tree.w2.pack, root)):
        oc = ob.OracleCircuit(orc, pack)
        assert oc.verify(proof) == 0
        oc.close()
    rows = np.stack([lc.proof_public_inputs(p, 21) for p in leaves])
    n_root = A.public_batch_pi_len(2, 2)
    got = lc.proof_public_inputs(root, n_root)
    assert got.tolist() == tree.expected_root_public_inputs(rows).tolist()
    assert lc.proof_public_inputs(level1[1], 50).tolist() == A.private_batch_outputs(rows[2:], tree.preimages(1)).tolist()
    hdr, slots, nulls = A.parse_public_batch_public_inputs(got, 2, 2)
    assert hdr["aggregator_address"] == addr and hdr["block_hash"] == bytes(sp[0].block_hash) and hdr["total_exit_slots"] == 8
    # batch 0: (200 -> e1, 97 -> e2), the dummy's two slots masked; batch 1: e1 gets 1 + 2 + 40, e2 gets 30, the repeats zeroed
    assert slots == [(200, e1), (97, e2), (0, bytes(32)), (0, bytes(32)), (43, e1), (0, bytes(32)), (30, e2), (0, bytes(32))]
    assert sorted(nulls[:2]) == sorted([bytes(sp[0].nullifier), A.dummy_nullifier(tree.preimages(0)[1])]) and sorted(nulls[2:]) == sorted([bytes(sp[1].nullifier), bytes(sp[2].nullifier)])
    # slots the private-batch layer does not accept: the host restatement says why, and the device witness generator finds no witness
    good1 = lc.proof_public_inputs(level1[1], 50)
    lp = L.LeafProver(pkg, gpu, tree.leaf)
    other = lp.prove(lc.real_inputs(L, depth=2))[0]            # a spend of another block
    lp.close()
    for bad_slots, needle in (([leaves[2], leaves[2]], "nullifier"), ([leaves[2], other], "block")):
        with pytest.raises(pkg.QpGpuError) as e:
            tree.w1.commit(bad_slots, preimages=tree.preimages(1))
        assert e.value.code == -4 and needle in str(e.value)
        c = tree.w1.commit(bad_slots, preimages=tree.preimages(1), public_inputs=good1)
        st = tree.w1_circ.generate_witness_partial_batch_dev(c[0], c[1][None], c[2][None], tree.d_wires)
        assert st == [-4] and "set twice with different values" in gpu.last_error()
    tree.close()


def test_full_verification_on_the_device(pkg, gpu, orc):
    """QPGPU_WRAPPER_VERIFY on the device: the wrapper's witness (ArithmeticExtension / Reducing / ReducingExtension rows and the
    quotient hints of the in-circuit verifier next to the Poseidon / RandomAccess rows) generated by stage s1 equals the oracle's,
    the proof equals the oracle's bytes; an inner proof the device prover made from a trace that violates the leaf circuit has a
    witness in the wrapper WITHOUT the flag and none with it (QPGPU_EUNSAT)."""
    L = pkg.leaf
    leaf = L.LeafCircuit()
    lp = L.LeafProver(pkg, gpu, leaf)
    xs = [lc.real_inputs(L, depth=4, seed=8), lc.dummy_inputs(L)]
    proofs = [lp.prove(x)[0] for x in xs]
    ver = pkg.Verifier(leaf.pack, circuit=lp.circ)
    wv = pkg.recursion.WrapperCircuit(leaf.pack, ver, 2, verify=True)
    w0 = pkg.recursion.WrapperCircuit(leaf.pack, ver, 2)
    wc, w0c = pkg.Circuit(gpu, wv.pack), pkg.Circuit(gpu, w0.pack)
    nw, n = 135, 1 << wv.info["degree_bits"]
    assert w0.info["degree_bits"] == wv.info["degree_bits"] == 13
    d = gpu.alloc(nw * n * 8)
    cells, vals, pis = wv.commit(proofs)
    wc.generate_witness_partial_dev(cells, vals, pis, d)
    rc, want, _ = orc.generate_witness(wv.pack, cells, vals, pis)
    assert rc == orc.WIT_OK and np.array_equal(d.download().reshape(nw, n), want)
    wc.set_witness_check(True)
    proof = wc.prove_dev(d, pis)
    oc = ob.OracleCircuit(orc, wv.pack)
    assert proof == oc.prove(want, pis) and oc.verify(proof) == 0
    oc.close()
    wver = pkg.Verifier(wv.pack, circuit=wc)
    assert wver.verify(proof)
    # a leaf proof of a trace that violates the leaf circuit, made by the device prover (its witness check is off by default)
    pis_leaf = lp.generate_witness(xs[0])
    trace = lp.witness().copy()
    trace[3, 0] = (int(trace[3, 0]) + 1) % pkg.P
    lp.d_wires.upload(trace)
    forged = lp.circ.prove_dev(lp.d_wires, pis_leaf)
    assert not ver.verify(forged)
    c0 = w0.commit([proofs[1], forged])
    w0c.generate_witness_partial_dev(c0[0], c0[1], c0[2], d)                     # Merkle half + transcript: nothing to object to
    cv = wv.commit([proofs[1], forged])
    with pytest.raises(pkg.QpGpuError) as e:
        wc.generate_witness_partial_dev(cv[0], cv[1], cv[2], d)
    assert e.value.code == -4 and "set twice with different values" in str(e.value)
    wver.close(); wc.close(); w0c.close(); ver.close(); lp.close()
    d.free(scrub=True)


def test_blinding_wires_drawn_on_the_device(pkg, gpu, orc):
    """The zero-knowledge private-batch circuit's RandomValueGenerator targets (CircuitBuilder::blind) filled by the device from
    ChaCha20 under a per-witness key (qpgpu_generate_witness_partial_batch_blinded_dev): with those values handed to the oracle as
    ordinary assignments its witness equals the device's; another key gives other blinding wires and the same everything else;
    the proof (salts seeded) equals the oracle's bytes and verifies."""
    L = pkg.leaf
    leaf = L.LeafCircuit()
    lp = L.LeafProver(pkg, gpu, leaf)
    proofs = [lp.prove(x)[0] for x in (lc.real_inputs(L, depth=2, seed=21), lc.dummy_inputs(L))]
    ver = pkg.Verifier(leaf.pack, circuit=lp.circ)
    wz = pkg.recursion.WrapperCircuit(leaf.pack, ver, 2, num_routed_wires=60, logic="private_batch", verify=True, zero_knowledge=True)
    wc = pkg.Circuit(gpu, wz.pack, max_batch=2)
    nw, n = 135, 1 << wz.info["degree_bits"]
    nb = wz.blinding_cells.size
    pre = np.arange(8, dtype=np.uint64).reshape(2, 4) + 5
    c = wz.commit(proofs, preimages=pre, device_blinding=True)
    assert c[0].size == c[1].size + nb and np.array_equal(c[0][-nb:], wz.blinding_cells)
    d = gpu.alloc(2 * nw * n * 8)
    seeds = bytes([7] * 32) + bytes([8] * 32)
    st = wc.generate_witness_partial_batch_blinded_dev(c[0], np.stack([c[1], c[1]]), np.stack([c[2], c[2]]), d, nb, seeds)
    assert st == [0, 0]
    wires = d.download().reshape(2, nw, n)
    cols, rows = (wz.blinding_cells % 135).astype(np.int64), (wz.blinding_cells // 135).astype(np.int64)
    blind = [wires[k][cols, rows] for k in range(2)]
    assert int(max(blind[0].max(), blind[1].max())) < pkg.P and not np.array_equal(blind[0], blind[1])
    assert len(np.unique(blind[0])) == nb                                    # 0.6 M draws from 2^64: no repeats
    assert abs(float((blind[0] >> np.uint64(63)).mean()) - 0.5) < 0.01        # top bit of a uniform element of [0, p)
    # everything that is not a blinding wire or a copy of one is the same witness under both keys
    pair_rows = np.zeros(n, dtype=bool); pair_rows[wz.info["rows_before_padding"]:wz.info["rows_before_padding"] + wz.info["rows_blinding"]] = True
    assert np.array_equal(wires[0][:, ~pair_rows], wires[1][:, ~pair_rows])
    for k in range(2):
        rc, want, _ = orc.generate_witness(wz.pack, c[0], np.concatenate([c[1], blind[k]]), c[2])
        assert rc == orc.WIT_OK and np.array_equal(wires[k], want)
    wc.set_witness_check(True)
    wc.set_blinding_seed(99)
    proof = wc.prove_dev(d, c[2])
    oc = ob.OracleCircuit(orc, wz.pack)
    assert oc.verify(proof) == 0
    oc.close()
    wv = pkg.Verifier(wz.pack, circuit=wc)
    assert wv.verify(proof)
    # OS entropy: two calls, two different sets of blinding wires
    st = wc.generate_witness_partial_batch_blinded_dev(c[0], np.stack([c[1], c[1]]), np.stack([c[2], c[2]]), d, nb)
    w2 = d.download().reshape(2, nw, n)
    assert st == [0, 0] and not np.array_equal(w2[0][cols, rows], w2[1][cols, rows]) and not np.array_equal(w2[0][cols, rows], blind[0])
    # a cell that is not free cannot be a blinding cell
    bad_cells = c[0].copy(); bad_cells[-1] = c[0][0]
    with pytest.raises(pkg.QpGpuError) as e:
        wc.generate_witness_partial_batch_blinded_dev(bad_cells, c[1][None], c[2][None], d, nb)
    assert "blinding cell" in str(e.value)
    wv.close(); wc.close(); ver.close(); lp.close()
    d.free(scrub=True)


def test_recursion_under_the_poseidon2_hasher_on_the_device(pkg, orc):
    """tests/test_wrapper_circuit.py::test_recursion_under_the_poseidon2_hasher through the device: with Poseidon2 as the proof-system
    hasher the wrapper's hashing is Poseidon2-gate rows whose swap wire is a Merkle index bit; stage s1 (the lane-cooperative
    Poseidon2 row generator) and the prover reproduce the oracle's witness and proof bytes."""
    L = pkg.leaf
    qp = pkg.poseidon2_qp_params()
    pkg.set_hasher_poseidon2(*qp); orc.select_poseidon2(*qp)
    try:
        g2 = pkg.QpGpu(0, hasher=qp)
        leaf = L.LeafCircuit(inner_hasher=1)
        lp = L.LeafProver(pkg, g2, leaf)
        proofs = [lp.prove(x)[0] for x in (lc.real_inputs(L, depth=2, seed=9), lc.dummy_inputs(L))]
        ver = pkg.Verifier(leaf.pack, circuit=lp.circ, hasher=1)
        assert all(ver.verify(p) for p in proofs)
        w = pkg.recursion.WrapperCircuit(leaf.pack, ver, 2, inner_hasher=1, logic="private_batch", verify=True)
        assert w.info["rows_poseidon"] == 0
        wc = pkg.Circuit(g2, w.pack)
        nw, n = 135, 1 << w.info["degree_bits"]
        d = g2.alloc(nw * n * 8)
        c = w.commit(proofs, preimages=np.arange(8, dtype=np.uint64).reshape(2, 4))
        wc.generate_witness_partial_dev(c[0], c[1], c[2], d)
        rc, want, _ = orc.generate_witness(w.pack, *c)
        assert rc == orc.WIT_OK and np.array_equal(d.download().reshape(nw, n), want)
        proof = wc.prove_dev(d, c[2])
        oc = ob.OracleCircuit(orc, w.pack)
        assert proof == oc.prove(want, c[2]) and oc.verify(proof) == 0
        oc.close()
        bad = bytearray(proofs[0]); bad[len(bad) // 2] ^= 1
        cb = w.commit([bytes(bad), proofs[1]], preimages=np.arange(8, dtype=np.uint64).reshape(2, 4), public_inputs=c[2])
        with pytest.raises(pkg.QpGpuError) as e:
            wc.generate_witness_partial_dev(cb[0], cb[1], cb[2], d)
        assert e.value.code == -4
        d.free(scrub=True); wc.close(); ver.close(); lp.close(); g2.close()
    finally:
        pkg.set_hasher_poseidon(); orc.select_poseidon()
