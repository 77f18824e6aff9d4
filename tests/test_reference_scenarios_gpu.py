"""The reference's own prover / verifier test scenarios, one for one, through the C ABI on the device — the scenarios of
wormhole/tests/src/prover/prover_tests.rs, wormhole/tests/src/verifier/verifier_tests.rs and the hex hand-over of
wormhole/tests/src/aggregator/aggregator_tests.rs:354-392 that the other test files cover only in passing. Each test names the
reference test it follows. Proof bytes are additionally compared with the oracle's wherever a proof is made."""
import ctypes

import numpy as np
import pytest

import leaf_cases as lc
import oracle_binding as ob

pytestmark = pytest.mark.gpu
P = 0xFFFFFFFF00000001


@pytest.fixture(scope="module")
def L(pkg):
    return pkg.leaf


@pytest.fixture(scope="module")
def full(L):
    return L.LeafCircuit()


@pytest.fixture(scope="module")
def prover(pkg, gpu, L, full):
    p = L.LeafProver(pkg, gpu, full)
    yield p
    p.close()


@pytest.fixture(scope="module")
def verifier(pkg, full, prover):
    v = pkg.Verifier(full.pack, circuit=prover.circ)
    yield v
    v.close()


@pytest.fixture(scope="module")
def proof0(L, prover):
    """prover.commit(&CircuitInputs::test_inputs_0()).unwrap().prove().unwrap() — the proof six reference tests start from"""
    return prover.prove(lc.test_inputs(L, 0))[0]


def with_public_inputs(proof, pis):
    return proof[:len(proof) - 8 * len(pis)] + np.asarray(pis, dtype=np.uint64).tobytes()


# ---- wormhole/tests/src/verifier/verifier_tests.rs ----------------------------------------------------------------------------

def test_verify_simple_proof(verifier, proof0):
    """verifier_tests.rs:39-48"""
    assert verifier.verify(proof0)


def test_borrowed_verify_keeps_proof_available(verifier, proof0):
    """verifier_tests.rs:50-66: verifying does not consume or change the proof; it still carries its 21 public inputs"""
    before = bytes(proof0)
    assert verifier.verify(proof0) and verifier.verify(proof0)
    assert proof0 == before and lc.proof_public_inputs(proof0, 21).size == 21


def test_cannot_verify_with_modified_exit_account(L, verifier, proof0):
    """verifier_tests.rs:86-108: exit_account_1 (public inputs 8..12) replaced by the account [8; 32]"""
    pis = lc.proof_public_inputs(proof0, 21).copy()
    assert pis[8:12].tolist() == lc.digest_felts(lc.test_inputs(L, 0).get32("exit_account_1"))
    pis[8:12] = lc.digest_felts(bytes([8] * 32))
    assert not verifier.verify(with_public_inputs(proof0, pis))


def test_cannot_verify_with_any_public_input_modification(verifier, proof0):
    """verifier_tests.rs:110-128: every public input, every byte of it XORed with 255 cumulatively (168 proofs). A value that leaves
    the field is refused by the decoder, as ProofWithPublicInputs::from_bytes does upstream; either way it does not verify."""
    pis0 = lc.proof_public_inputs(proof0, 21)
    for ix in range(21):
        v = int(pis0[ix])
        for jx in range(8):
            v ^= 255 << (8 * jx)
            pis = pis0.copy(); pis[ix] = np.uint64(v)
            assert not verifier.verify(with_public_inputs(proof0, pis)), (ix, jx)


def test_cannot_verify_with_modified_proof(verifier, proof0, orc, full):
    """verifier_tests.rs:130-152 (ignored upstream for its run time): one byte XORed with 255. Every 61st byte here, and the oracle's
    verifier gives the same verdict on each."""
    oc = ob.OracleCircuit(orc, full.pack)
    assert oc.verify(proof0) == 0
    for ix in range(0, len(proof0), 61):
        b = bytearray(proof0); b[ix] ^= 255
        assert not verifier.verify(bytes(b)), ix
        assert oc.verify(bytes(b)) != 0, ix
    oc.close()


# ---- wormhole/tests/src/prover/prover_tests.rs ---------------------------------------------------------------------------------

def test_commit_and_prove_and_proof_can_be_deserialized(L, full, prover, proof0, orc):
    """prover_tests.rs:15-20, 39-51, 53-61: the proof's public inputs are PublicCircuitInputs of the canonical fixture
    (wormhole/inputs/src/lib.rs:68-80); its bytes are the oracle's"""
    x = lc.test_inputs(L, 0)
    got = lc.proof_public_inputs(proof0, 21)
    assert got[:4].tolist() == [x.asset_id, x.output_amount_1, x.output_amount_2, x.volume_fee_bps]
    assert got[4:8].tolist() == lc.digest_felts(x.get32("nullifier"))
    assert got[8:12].tolist() == lc.digest_felts(x.get32("exit_account_1")) and got[12:16].tolist() == lc.digest_felts(x.get32("exit_account_2"))
    assert got[16:20].tolist() == lc.digest_felts(x.get32("block_hash")) and int(got[20]) == x.block_number
    cells, values, pis = full.commit(x)
    rc, wires, _ = orc.generate_witness(full.pack, cells, values, pis)
    oc = ob.OracleCircuit(orc, full.pack)
    assert rc == orc.WIT_OK and oc.prove(wires, pis) == proof0
    oc.close()


def test_commit_rejects_zk_merkle_proof_exceeding_max_depth(L, full):
    """prover_tests.rs:22-37"""
    x = lc.test_inputs(L, 0)
    x.zk_merkle_depth = 17
    with pytest.raises(ValueError) as e:
        full.commit(x)
    assert "ZK Merkle proof depth" in str(e.value)


def random_tree_case(L, num_leaves, leaf_index, seed):
    """prover_tests.rs:131-233, 286-399 (and 401-615 at 16 / 64 leaves): random secrets, their leaves in a 4-ary tree built bottom
    up with the sorted-children node hash, the proof of one leaf (sorted siblings + positions), verified natively and then by the
    circuit in dummy-header mode (block_hash = 0, outputs = 0: the header fragment is skipped, the tree root is a private input)."""
    rng = np.random.default_rng(seed)
    secrets = []
    for _ in range(num_leaves):
        b = rng.integers(0, 256, 32, dtype=np.uint8); b[7::8] &= 0x7F
        secrets.append(b.tobytes())
    tc, asset, amount = 1, 0, 100
    unsp = [L.unspendable_account(s) for s in secrets]
    level = [L.zk_leaf_hash(u, tc, asset, amount) for u in unsp]
    levels = [level]
    while len(level) > 1:
        level = [L.zk_proof_from_unsorted(level[g], [level[g + 1:g + 4]])[2] for g in range(0, len(level), 4)]
        levels.append(level)
    root = levels[-1][0]
    sibs, idx = [], leaf_index
    for lvl in levels[:-1]:
        g = idx - idx % 4
        sibs.append([lvl[k] for k in range(g, g + 4) if k != idx])
        idx //= 4
    sorted_sibs, positions, r = L.zk_proof_from_unsorted(levels[0][leaf_index], sibs)
    assert r == root                                                          # verify_proof_native
    hk = lc.header_kat(0)
    x = L.LeafInputs()
    x.asset_id, x.output_amount_1, x.output_amount_2, x.volume_fee_bps = asset, 0, 0, 10
    x.transfer_count, x.input_amount, x.block_number = tc, amount, hk[1]
    x.set32("secret", secrets[leaf_index]).set32("unspendable_account", unsp[leaf_index]).set32("nullifier", L.nullifier(secrets[leaf_index], tc))
    x.set32("exit_account_1", bytes([4] * 32)).set32("state_root", hk[2]).set32("extrinsics_root", hk[3]).set32("zk_tree_root", root)
    ctypes.memmove(x.digest, hk[5], 110)
    x.zk_merkle_depth = len(sibs)
    ctypes.memmove(x.zk_merkle_siblings, sorted_sibs, len(sorted_sibs))
    for l, p in enumerate(positions):
        x.zk_merkle_positions[l] = p
    return x


@pytest.mark.parametrize("num_leaves,indices", [(4, range(4)), (16, (0, 5, 10, 15)), (64, (0, 21, 42, 63))])
def test_random_tree_circuit_verification(L, full, prover, verifier, orc, num_leaves, indices):
    """prover_tests.rs:286-399 (depth 1), 401-498 (depth 2), 500-615 (depth 3)"""
    oc = ob.OracleCircuit(orc, full.pack)
    for i in indices:
        x = random_tree_case(L, num_leaves, i, seed=42 + num_leaves)
        proof, pis = prover.prove(x)
        assert verifier.verify(proof), (num_leaves, i)
        cells, values, want_pis = full.commit(x)
        rc, wires, _ = orc.generate_witness(full.pack, cells, values, want_pis)
        assert rc == orc.WIT_OK and oc.prove(wires, want_pis) == proof, (num_leaves, i)
    # what "dummy-header mode" means for this check (zk_merkle_proof.rs:620-623: the root comparison is multiplied by is_not_dummy):
    # the same spend under a root that is NOT the tree's still proves, and once the header is real the very same path is enforced
    # (tests/test_leaf_circuit_gpu.py::test_unsatisfiable_inputs_name_the_target flips a sibling of a real spend)
    y = x.copy(); y.set32("zk_tree_root", bytes([1] * 32))
    assert verifier.verify(prover.prove(y)[0])
    oc.close()


# ---- wormhole/tests/src/aggregator/aggregator_tests.rs:354-392 -------------------------------------------------------------------

def test_aggregate_proofs_from_separate_prover_instances_hex_serialized(pkg, gpu, L, full, orc):
    """Two leaf proofs of one block from two prover instances, handed over as hex text (hex::encode(proof.to_bytes())), decoded,
    aggregated by a private-batch prover built separately; the aggregate verifies."""
    lib = pkg.load_library()
    lib.qpgpu_hex_encode.restype = ctypes.c_size_t
    lib.qpgpu_hex_encode.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]
    lib.qpgpu_hex_decode.restype = ctypes.c_size_t
    lib.qpgpu_hex_decode.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]
    x1, x2 = lc.shared_tree_inputs(L, 2, seed=77)
    texts = []
    for x in (x1, x2):
        p = L.LeafProver(pkg, gpu, L.LeafCircuit())                           # a prover of its own: circuit built again, loaded again
        proof = p.prove(x)[0]
        p.close()
        out = ctypes.create_string_buffer(2 * len(proof) + 1)
        assert lib.qpgpu_hex_encode(proof, len(proof), out, len(out)) == 2 * len(proof)
        texts.append(out.value)
    proofs = []
    for t in texts:
        back = ctypes.create_string_buffer(len(t) // 2)
        assert lib.qpgpu_hex_decode(t, len(t), back, len(back)) == len(t) // 2
        proofs.append(back.raw)
    priv = pkg.recursion.PrivateBatchProver(pkg, gpu, full, 2)
    agg = priv.aggregate(proofs, seed=bytes([5] * 32))
    assert priv.verifier.verify(agg)
    oc = ob.OracleCircuit(orc, priv.circuit.pack)
    assert oc.verify(agg) == 0
    oc.close()
    A = pkg.aggregation
    hdr, slots, nulls = A.parse_private_batch_public_inputs(A.proof_public_inputs(agg, A.private_batch_pi_len(2)))
    assert set(nulls) == {bytes(x1.nullifier), bytes(x2.nullifier)} and hdr["block_hash"] == bytes(x1.block_hash)
    priv.close()


# ---- wormhole/tests/src/aggregator/aggregator_tests.rs: the private-batch prover's preflights -------------------------------------

def test_private_batch_preflights_and_a_full_batch_of_a_non_native_asset(pkg, gpu, L, full, orc):
    """aggregator_tests.rs:289-352, 395-412 with a two-slot private-batch prover: a non-native-asset leaf that would need dummy padding,
    an empty batch, two assets in one batch and a proof with a public input popped are refused at commit, each with the reference's
    reason; a FULL batch of one non-native asset aggregates and verifies."""
    priv = pkg.recursion.PrivateBatchProver(pkg, gpu, full, 2)
    lp = priv.leaf_prover
    a5 = [lp.prove(x)[0] for x in lc.shared_tree_inputs(L, 2, depth=1, seed=5, asset_id=5)]
    mixed = [lp.prove(x)[0] for x in lc.shared_tree_inputs(L, 2, depth=1, seed=6, asset_id=[0, 5])]
    for batch, needle in (([a5[0]], "dummy proofs use asset_id=0"),                    # commit_rejects_nonzero_asset_id_when_dummy_padding_is_needed
                          ([], "no leaf proofs"),                                      # aggregate_rejects_empty_batch
                          (mixed, "asset"),                                            # commit_rejects_batch_incompatible_proofs
                          ([a5[0][:-8], a5[1]], "leaf proof public input length mismatch")):   # private_batch_commit_rejects_malformed_full_batch_at_api_boundary
        with pytest.raises(ValueError) as e:
            priv.commit(batch)
        assert needle in str(e.value), str(e.value)
    agg = priv.aggregate(a5, seed=bytes([8] * 32))                                    # full_batch_of_same_nonzero_asset_aggregates
    assert priv.verifier.verify(agg)
    oc = ob.OracleCircuit(orc, priv.circuit.pack)
    assert oc.verify(agg) == 0
    oc.close()
    A = pkg.aggregation
    hdr, _, _ = A.parse_private_batch_public_inputs(A.proof_public_inputs(agg, A.private_batch_pi_len(2)))
    assert hdr["asset_id"] == 5
    priv.close()
