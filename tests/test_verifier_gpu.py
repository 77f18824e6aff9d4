"""The product verifier (include/qpgpu_verify.h) on proofs made by the GPU: verifier data = the constants/sigmas cap of the
loaded circuit handle, as VerifierOnlyCircuitData carries it. Covers the exact bench shape (2^13 rows, Poseidon + BaseSum, 80
routed), a lockstep batch verified on several host threads, zero knowledge, and agreement with the oracle's verifier on
tampered proofs."""
import time
import numpy as np
import pytest

from oracle_binding import OracleCircuit


@pytest.mark.gpu
def test_bench_shape_proof_is_accepted_and_bound(pkg, gpu, orc):
    pack, wires, pis = pkg.synth_circuit(13, num_wires=135, num_routed=80, num_public_inputs=21, seed=1000, poseidon=True, base_sum=True)
    circ = pkg.Circuit(gpu, pack)
    v = pkg.Verifier(pack, circuit=circ)
    oc = OracleCircuit(orc, pack)
    try:
        proof = circ.prove(wires, pis)
        assert v.proof_size() == len(proof)
        t0 = time.perf_counter()
        assert v.verify(proof), v.reason
        ms = (time.perf_counter() - t0) * 1e3
        assert ms < 500, ms
        rng = np.random.default_rng(3)
        for pos in list(rng.integers(0, len(proof), 12)) + [len(proof) - 1]:
            b = bytearray(proof); b[pos] ^= 0x10
            assert not v.verify(bytes(b)) and v.reason
            assert oc.verify(bytes(b)) != 0
        # the cap rebuilt from the pack on the host is the handle's
        host = pkg.Verifier(pack)
        assert host.verify(proof)
        host.close()
    finally:
        v.close(); circ.close(); oc.close()


@pytest.mark.gpu
def test_lockstep_batch_verified_on_host_threads(pkg, gpu):
    agg = pkg.aggregation
    pack, wires, _ = pkg.synth_circuit(9, num_wires=135, num_routed=60, num_public_inputs=21, seed=321, poseidon=True, base_sum=True, ext_arith=True, recursion=True)
    pack[14] = 1                                              # zero knowledge: salted openings
    tp = agg.TemplateProver(gpu, pack, wires, max_batch=8)
    v = pkg.Verifier(pack, circuit=tp.circ)
    try:
        tp.commit_many([agg.leaf_public_inputs(i) for i in range(8)])
        proofs = tp.prove_many()
        assert v.verify_many(proofs) == [True] * 8
        bad = bytearray(proofs[5]); bad[40] ^= 1
        res = v.verify_many(proofs[:5] + [bytes(bad)] + proofs[6:], threads=4)
        assert res == [True] * 5 + [False] + [True] * 2 and v.reason.startswith("proof 5:")
    finally:
        v.close(); tp.close()
