"""PrivateBatchProver / PublicBatchProver commit + prove on the GPU (aggregation.py over include/qpgpu_batch.h): the reference's
admission behaviour at the commit boundary (private_batch/prover/lib.rs:244-343, public_batch/prover/lib.rs:268-305), padding with
the dummy templates, the private level's shuffle, and proofs whose public inputs parse as the reference's layouts. The CPU
oracle is the cryptographic verifier handed to the provers and the checker of every proof."""
import ctypes
import numpy as np
import pytest

from oracle_binding import OracleCircuit

N_LEAF, N_INNER = 4, 2


@pytest.fixture(scope="module")
def levels(pkg, gpu, orc):
    agg = pkg.aggregation
    rec = dict(poseidon=True, base_sum=True, ext_arith=True, recursion=True)
    leaf = pkg.synth_circuit(7, num_wires=135, num_routed=80, num_public_inputs=21, seed=501, poseidon=True, base_sum=True)
    priv = pkg.synth_circuit(8, num_wires=135, num_routed=60, num_public_inputs=agg.private_batch_pi_len(N_LEAF), seed=502, **rec)
    priv[0][14] = 1
    pub = pkg.synth_circuit(8, num_wires=135, num_routed=80, num_public_inputs=agg.public_batch_pi_len(N_INNER, N_LEAF), seed=503, **rec)
    ocs = {k: OracleCircuit(orc, v[0]) for k, v in (("leaf", leaf), ("priv", priv), ("pub", pub))}
    lp = agg.TemplateProver(gpu, leaf[0], leaf[1])

    def prove_leaf(pis):
        lp.commit(pis)
        return lp.prove()
    dummy_leaf = prove_leaf(np.zeros(21, dtype=np.uint64))
    leaves = [prove_leaf(agg.leaf_public_inputs(i)) for i in range(5)]
    yield dict(agg=agg, leaf=leaf, priv=priv, pub=pub, ocs=ocs, dummy_leaf=dummy_leaf, leaves=leaves, prove_leaf=prove_leaf)
    lp.close()
    for oc in ocs.values():
        oc.close()


@pytest.mark.gpu
def test_private_batch_commit_pads_shuffles_and_proves(pkg, gpu, levels):
    agg, ocs = levels["agg"], levels["ocs"]
    verify_leaf = lambda ps: [ocs["leaf"].verify(p) == 0 for p in ps]
    pp = agg.PrivateBatchProver(gpu, levels["priv"][0], levels["priv"][1], levels["dummy_leaf"], N_LEAF, verify_leaf)
    try:
        pis = pp.commit(levels["leaves"][:3], seed=bytes(range(32)))
        src, pre = pp.arrangement[0]
        assert sorted(src.tolist()) == [0, 1, 2, 0xFFFFFFFF]                      # one padded slot, somewhere
        pp.circ.set_blinding_seed(99)
        w = pp.witness()
        proof = pp.prove()
        assert proof == ocs["priv"].prove(w, pis, 99) and ocs["priv"].verify(proof) == 0
        got = agg.proof_public_inputs(proof, agg.private_batch_pi_len(N_LEAF))
        assert np.array_equal(got, pis) and int(got[0]) == 2 * N_LEAF and tuple(got[3:7].tolist()) == agg.TEST_BLOCK_HASH
        want = {}
        for i in range(3):
            lp = agg.leaf_public_inputs(i)
            for acct, amt in ((tuple(lp[8:12].tolist()), int(lp[1])), (tuple(lp[12:16].tolist()), int(lp[2]))):
                want[acct] = want.get(acct, 0) + amt
        slots = {tuple(got[9 + 5 * k:13 + 5 * k].tolist()): int(got[8 + 5 * k]) for k in range(2 * N_LEAF) if int(got[8 + 5 * k])}
        assert slots == {a: v for a, v in want.items() if v}
        region = [tuple(got[48 + 4 * k:52 + 4 * k].tolist()) for k in range(N_LEAF)]
        assert region == sorted(region) and all(tuple(agg.leaf_public_inputs(i)[4:8].tolist()) in region for i in range(3))
        # another seed, another slot order; OS entropy by default
        pp.commit(levels["leaves"][:3], seed=bytes(range(1, 33)))
        assert not np.array_equal(pp.arrangement[0][1], pre)
        # ---- the commit boundary refuses what the reference refuses, with its words ----
        with pytest.raises(ValueError, match="no leaf proofs to aggregate"):
            pp.commit([])
        with pytest.raises(ValueError, match="too many proofs: got 5, expected at most 4"):
            pp.commit(levels["leaves"])
        with pytest.raises(ValueError, match="same nullifier"):
            pp.commit([levels["leaves"][0], levels["leaves"][0]])
        with pytest.raises(ValueError, match="all-dummy"):
            pp.commit([levels["dummy_leaf"]])
        other_block = agg.leaf_public_inputs(9); other_block[16] += 1
        with pytest.raises(ValueError, match="different block"):
            pp.commit([levels["leaves"][0], levels["prove_leaf"](other_block)])
        tampered = bytearray(levels["leaves"][1]); tampered[100] ^= 1             # commit_rejects_tampered_leaf_proof_before_proving
        with pytest.raises(ValueError, match="leaf proof 1 failed verification"):
            pp.commit([levels["leaves"][0], bytes(tampered)])
        # a padding template that is not a dummy is refused at construction
        with pytest.raises(ValueError, match="non-zero block_hash"):
            agg.PrivateBatchProver(gpu, levels["priv"][0], levels["priv"][1], levels["leaves"][0], N_LEAF)
    finally:
        pp.close()


@pytest.mark.gpu
def test_public_batch_commit_pads_in_order_and_proves(pkg, gpu, levels):
    agg, ocs = levels["agg"], levels["ocs"]
    pp = agg.PrivateBatchProver(gpu, levels["priv"][0], levels["priv"][1], levels["dummy_leaf"], N_LEAF)
    try:
        dummy_batch = pp.prove_dummy_template()
        assert ocs["priv"].verify(dummy_batch) == 0
        pp.commit(levels["leaves"][:4])
        inner = pp.prove()
    finally:
        pp.close()
    verify_inner = lambda ps: [ocs["priv"].verify(p) == 0 for p in ps]
    address = b"".join(v.to_bytes(8, "little") for v in (5, 6, 7, 8))
    pub = agg.PublicBatchProver(gpu, levels["pub"][0], levels["pub"][1], dummy_batch, N_INNER, N_LEAF, verify_inner)
    try:
        pis = pub.commit([inner], address)
        w = pub.witness()
        root = pub.prove()
        assert root == ocs["pub"].prove(w, pis) and ocs["pub"].verify(root) == 0
        ip = agg.proof_public_inputs(inner, agg.private_batch_pi_len(N_LEAF))
        assert pis[:4].tolist() == [5, 6, 7, 8] and int(pis[11]) == N_INNER * 2 * N_LEAF and tuple(pis[6:10].tolist()) == agg.TEST_BLOCK_HASH
        assert np.array_equal(pis[12:52], ip[8:48]) and not pis[52:92].any()          # the real inner's slots, then the dummy's zeroed
        assert np.array_equal(pis[92:108], ip[48:64]) and not pis[108:124].any()
        # parses as PublicBatchPublicInputs
        L = pkg.load_library()
        L.qpgpu_public_batch_public_inputs_parse.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p,
                                                             ctypes.c_void_p, ctypes.c_void_p, ctypes.c_char_p]
        hdr = (ctypes.c_uint8 * 96)(); err = ctypes.create_string_buffer(400)
        assert L.qpgpu_public_batch_public_inputs_parse(pis.ctypes.data, pis.size, N_INNER, N_LEAF, hdr, None, None, err) == 0, err.value
        with pytest.raises(ValueError, match="no private-batch proofs to aggregate"):
            pub.commit([])
        with pytest.raises(ValueError, match="Expected at most 2 private-batch proofs, but got 3"):
            pub.commit([inner, inner, inner])
        with pytest.raises(ValueError, match="all-dummy"):                             # commit_rejects_all_dummy_batch
            pub.commit([dummy_batch])
        bad = bytearray(inner); bad[200] ^= 4
        with pytest.raises(ValueError, match="private-batch proof 0 failed verification"):
            pub.commit([bytes(bad)])
        with pytest.raises(ValueError, match="non-zero block_hash"):                   # direct_constructors_reject_non_dummy_padding_templates
            agg.PublicBatchProver(gpu, levels["pub"][0], levels["pub"][1], inner, N_INNER, N_LEAF)
    finally:
        pub.close()
