"""The product's host-side verifier (include/qpgpu_verify.h, csrc/verifier.cpp): plonky2's VerifierCircuitData::verify over a
circuit pack — the acceptance criterion the reference applies to every proof (wormhole/tests/src/prover/verifier_tests.rs:40-66,
wormhole/aggregator/src/aggregator.rs:224-225, private_batch/prover/lib.rs:274-281).

Checked against the CPU oracle, which plays prover and second opinion: proofs the oracle makes (all fourteen gate types, zero
knowledge, Poseidon2 as the proof-system hasher) are accepted, every tampering the oracle's own verifier rejects is rejected
here too, and the constants/sigmas cap rebuilt from the pack equals the one inside the proofs' Merkle paths. No GPU needed;
tests/test_verifier_gpu.py runs the same verifier on GPU proofs."""
import ctypes
import numpy as np
import pytest

import __graft_entry__ as ge
from oracle_binding import OracleCircuit

pkg_ = ge.load_package()
L = pkg_.load_library()
L.qpgpu_verifier_create.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t,
                                    ctypes.POINTER(ctypes.c_void_p), ctypes.c_char_p]
L.qpgpu_verifier_free.argtypes = [ctypes.c_void_p]
L.qpgpu_verifier_proof_size.argtypes = [ctypes.c_void_p]; L.qpgpu_verifier_proof_size.restype = ctypes.c_size_t
L.qpgpu_verifier_verify.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p]
L.qpgpu_verifier_constants_sigmas_cap.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
EVERIFY = -6


class Verifier:
    def __init__(self, pack, cap=None, hasher=0, params=None):
        pw = np.ascontiguousarray(pack, dtype=np.uint64)
        h = ctypes.c_void_p(); err = ctypes.create_string_buffer(200)
        c = None if cap is None else np.ascontiguousarray(cap, dtype=np.uint64)
        p = None if params is None else np.ascontiguousarray(params, dtype=np.uint64)
        rc = L.qpgpu_verifier_create(pw.ctypes.data, pw.size, None if c is None else c.ctypes.data, 0 if c is None else c.size, hasher,
                                     None if p is None else p.ctypes.data, 0 if p is None else p.size, ctypes.byref(h), err)
        if rc:
            raise ValueError(err.value.decode())
        self.h = h

    def verify(self, proof):
        err = ctypes.create_string_buffer(200)
        rc = L.qpgpu_verifier_verify(self.h, bytes(proof), len(proof), err)
        return rc, err.value.decode()

    def proof_size(self):
        return L.qpgpu_verifier_proof_size(self.h)

    def cap(self, words):
        out = np.zeros(words, dtype=np.uint64)
        assert L.qpgpu_verifier_constants_sigmas_cap(self.h, out.ctypes.data, out.size) == 0
        return out

    def close(self):
        L.qpgpu_verifier_free(self.h)


def roundtrip(pkg, orc, pack, wires, pis, seed=0, tamper=24):
    oc = OracleCircuit(orc, pack)
    v = Verifier(pack)
    try:
        proof = oc.prove(wires, pis, seed)
        assert v.proof_size() == len(proof) == oc.proof_size()
        rc, msg = v.verify(proof)
        assert rc == 0, msg
        rng = np.random.default_rng(5)
        reasons = set()
        for pos in list(rng.integers(0, len(proof), tamper)) + [0, len(proof) - 1]:
            b = bytearray(proof); b[pos] ^= 0x01
            rc, msg = v.verify(bytes(b))
            assert rc == EVERIFY and msg, f"flipping byte {pos} was accepted"
            assert oc.verify(bytes(b)) != 0
            reasons.add(msg.split(":")[-1][:24])
        assert v.verify(proof[:-1])[0] == EVERIFY
        return reasons
    finally:
        v.close(); oc.close()


def test_small_circuit_accepts_and_rejects(pkg, orc):
    pack, wires, pis = pkg.synth_circuit(6, num_wires=24, num_routed=16, num_public_inputs=5, seed=3)
    assert len(roundtrip(pkg, orc, pack, wires, pis, tamper=60)) >= 3           # several different checks fire


def test_unsatisfied_witness_is_rejected_by_the_quotient_identity(pkg, orc):
    pack, wires, pis = pkg.synth_circuit(6, num_wires=24, num_routed=16, num_public_inputs=5, seed=3)
    oc = OracleCircuit(orc, pack); v = Verifier(pack)
    try:
        w = wires.copy(); w[2, 12] ^= 1                                          # a witness that violates a gate still yields bytes
        rc, msg = v.verify(oc.prove(w, pis))
        assert rc == EVERIFY and "quotient identity" in msg
        other = pis.copy(); other[0] ^= 1                                        # and the proof is bound to its public inputs
        proof = bytearray(oc.prove(wires, pis)); proof[-8 * len(pis):] = other.astype("<u8").tobytes()
        assert v.verify(bytes(proof))[0] == EVERIFY
    finally:
        v.close(); oc.close()


@pytest.mark.parametrize("kw", [dict(poseidon=True, base_sum=True),
                                dict(poseidon=True, base_sum=True, ext_arith=True, recursion=True)], ids=["leaf-mix", "recursion-mix"])
def test_standard_shape_all_gate_types(pkg, orc, kw):
    pack, wires, pis = pkg.synth_circuit(8, num_wires=135, num_routed=80, num_public_inputs=21, seed=77, **kw)
    roundtrip(pkg, orc, pack, wires, pis, tamper=10)


def test_zero_knowledge_proofs(pkg, orc):
    pack, wires, pis = pkg.synth_circuit(7, num_wires=135, num_routed=60, num_public_inputs=21, seed=5, poseidon=True, base_sum=True, ext_arith=True, recursion=True)
    pack[14] = 1                                                                  # salted leaves: wider rows in the Merkle openings
    roundtrip(pkg, orc, pack, wires, pis, seed=4242, tamper=8)


def test_cap_is_the_verifier_data(pkg, orc):
    pack, wires, pis = pkg.synth_circuit(6, num_wires=24, num_routed=16, num_public_inputs=5, seed=3)
    oc = OracleCircuit(orc, pack); v = Verifier(pack)
    try:
        proof = oc.prove(wires, pis)
        cap = v.cap(4 << 4)
        again = Verifier(pack, cap=cap)                                           # the cap a GPU circuit handle reports goes in here
        assert again.verify(proof)[0] == 0
        again.close()
        cap[5] ^= 1                                                               # another circuit's verifier data: constants/sigmas paths fail
        wrong = Verifier(pack, cap=cap)
        rc, msg = wrong.verify(proof)
        assert rc == EVERIFY and "initial oracle 0" in msg
        wrong.close()
        with pytest.raises(ValueError, match="cap has 8 words"):
            Verifier(pack, cap=np.zeros(8))
        with pytest.raises(ValueError, match="circuit pack"):
            Verifier(pack[:40])
    finally:
        v.close(); oc.close()


def test_poseidon2_as_the_proof_system_hasher(pkg, orc):
    """Both sides switched to Poseidon2 with the pinned qp-poseidon-core parameters: the verifier accepts the oracle's proof
    and refuses it under the other permutation."""
    blk = np.zeros(146, dtype=np.uint64)
    L.qpgpu_poseidon2_qp_params.argtypes = [ctypes.c_void_p, ctypes.c_size_t]; L.qpgpu_poseidon2_qp_params.restype = ctypes.c_size_t
    assert L.qpgpu_poseidon2_qp_params(blk.ctypes.data, 146) == 146
    parts = (blk[:96], blk[96:118], blk[118:130], blk[130:146])
    pkg.set_hasher_poseidon2(*parts)               # the synthetic circuit's public-input hash follows the process default
    orc.select_poseidon2(*parts)
    try:
        pack, wires, pis = pkg.synth_circuit(6, num_wires=24, num_routed=16, num_public_inputs=5, seed=9)
        oc = OracleCircuit(orc, pack)
        proof = oc.prove(wires, pis)
        assert oc.verify(proof) == 0
        v2 = Verifier(pack, hasher=1)                                             # NULL parameter block = the built-in set
        v2b = Verifier(pack, hasher=1, params=blk)
        assert v2.verify(proof)[0] == 0 and v2b.verify(proof)[0] == 0
        v1 = Verifier(pack, hasher=0)
        assert v1.verify(proof)[0] == EVERIFY
        for v in (v1, v2, v2b):
            v.close()
        oc.close()
    finally:
        orc.select_poseidon(); pkg.set_hasher_poseidon()


def test_every_public_input_bit_flip_fails(pkg, orc):
    """wormhole/tests/src/prover/verifier_tests.rs:110-128: a proof verifies only against the public inputs it was made for."""
    pack, wires, pis = pkg.synth_circuit(7, num_wires=135, num_routed=80, num_public_inputs=21, seed=12, poseidon=True, base_sum=True)
    oc = OracleCircuit(orc, pack); v = Verifier(pack)
    try:
        proof = oc.prove(wires, pis)
        assert v.verify(proof)[0] == 0
        base = len(proof) - 8 * len(pis)
        rng = np.random.default_rng(2)
        for i in range(len(pis)):
            for bit in (0, int(rng.integers(1, 63)), 63):
                b = bytearray(proof)
                b[base + 8 * i + bit // 8] ^= 1 << (bit % 8)
                rc, msg = v.verify(bytes(b))
                assert rc == EVERIFY, (i, bit)
    finally:
        v.close(); oc.close()
