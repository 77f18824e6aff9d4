"""The oracle's full prover (stages s4..s12) and verifier on synthetic circuit packs: the proof verifies,
every tampering is rejected, and the byte layout has the stated size."""
import numpy as np
import pytest

from oracle_binding import OracleCircuit


@pytest.fixture(scope="module")
def small(pkg, orc):
    pack, wires, pis = pkg.synth_circuit(6, num_wires=24, num_routed=16, num_public_inputs=5, seed=3)
    oc = OracleCircuit(orc, pack)
    proof = oc.prove(wires, pis)
    yield oc, pack, wires, pis, proof
    oc.close()


def test_proof_verifies(small):
    oc, pack, wires, pis, proof = small
    assert len(proof) == oc.proof_size()
    assert oc.verify(proof) == 0
    assert oc.prove(wires, pis) == proof           # deterministic (non-ZK, minimum PoW nonce)


def test_tampering_is_rejected(small):
    oc, pack, wires, pis, proof = small
    rng = np.random.default_rng(9)
    bad_codes = set()
    for pos in list(rng.integers(0, len(proof), 40)) + [0, len(proof) - 1, len(proof) - 8 * len(pis)]:
        b = bytearray(proof); b[pos] ^= 0x01
        code = oc.verify(bytes(b))
        assert code != 0, f"flipping byte {pos} was accepted"
        bad_codes.add(code)
    assert len(bad_codes) >= 2
    assert oc.verify(proof[:-1]) == 1


def test_unsatisfied_witness_fails_verification(small, orc):
    oc, pack, wires, pis, proof = small
    w = wires.copy()
    w[3, 10] = (int(w[3, 10]) + 1) % 0xFFFFFFFF00000001   # break an arithmetic output
    assert oc.verify(oc.prove(w, pis)) != 0
    p = pis.copy(); p[0] = (int(p[0]) + 1) % 0xFFFFFFFF00000001   # public inputs no longer match the PI gate
    assert oc.verify(oc.prove(wires, p)) != 0


def test_standard_shape_proof(pkg, orc):
    """standard_recursion_config shape: 135 wires, 80 routed, degree 2^9 keeps the CPU suite fast."""
    pack, wires, pis = pkg.synth_circuit(9, seed=5)
    oc = OracleCircuit(orc, pack)
    proof = oc.prove(wires, pis)
    assert oc.verify(proof) == 0
    assert len(oc.trace("query_indices")) == 28
    assert int(oc.trace("pow_witness")[0]) < 2**24
    oc.close()


def test_zero_knowledge_salts(pkg, orc):
    """Blinded oracles (private-batch config): 4 salt columns per leaf; reproducible under an injected seed,
    different under another seed, and the verifier strips the salts from the FRI combination."""
    pack, wires, pis = pkg.synth_circuit(6, num_wires=24, num_routed=16, num_public_inputs=5, seed=3)
    zk = pack.copy(); zk[14] = 1                      # header word 14 = zero_knowledge
    oc = OracleCircuit(orc, zk)
    plain = OracleCircuit(orc, pack)
    assert oc.proof_size() == plain.proof_size() + 28 * 3 * 4 * 8
    a = oc.prove(wires, pis, seed=11); b = oc.prove(wires, pis, seed=11); c = oc.prove(wires, pis, seed=12)
    assert a == b and a != c
    assert oc.verify(a) == 0 and oc.verify(c) == 0
    cap = 16 * 4 * 8
    assert a[:cap] != plain.prove(wires, pis)[:cap]    # salts change every commitment
    oc.close(); plain.close()


def test_poseidon_gate_circuit(pkg, orc):
    """PoseidonGate rows (degree 7) force a second selector group; a wrong S-box wire breaks the quotient identity."""
    pack, wires, pis = pkg.synth_circuit(7, seed=5, poseidon=True)
    assert int(pack[5]) == 2                          # num_selectors
    oc = OracleCircuit(orc, pack)
    proof = oc.prove(wires, pis)
    assert oc.verify(proof) == 0
    w = wires.copy(); w[70, 8] = (int(w[70, 8]) + 1) % 0xFFFFFFFF00000001    # partial-round S-box input, row 8
    assert oc.verify(oc.prove(w, pis)) == 3
    w = wires.copy(); w[24, 8] = 2                                            # swap must be binary
    assert oc.verify(oc.prove(w, pis)) == 3
    oc.close()


def test_base_sum_gate_circuit(pkg, orc):
    pack, wires, pis = pkg.synth_circuit(7, seed=5, base_sum=True, poseidon=True)
    oc = OracleCircuit(orc, pack)
    assert oc.verify(oc.prove(wires, pis)) == 0
    w = wires.copy(); w[3, 5] = 2                       # a limb that is not a bit (row 5 is a base-sum row)
    assert oc.verify(oc.prove(w, pis)) == 3
    w = wires.copy(); w[0, 5] = (int(w[0, 5]) + 1) % 0xFFFFFFFF00000001   # sum no longer matches its bits
    assert oc.verify(oc.prove(w, pis)) == 3
    oc.close()


def test_extension_arithmetic_gates_circuit(pkg, orc):
    """ArithmeticExtensionGate / MulExtensionGate rows (rows 6, 14, 22, ...) with every other gate present; the
    builder's grouping rule then puts MulExtension next to Poseidon in the second selector polynomial."""
    pack, wires, pis = pkg.synth_circuit(7, seed=6, base_sum=True, poseidon=True, ext_arith=True)
    assert int(pack[5]) == 2 and int(pack[16]) == 8          # 2 selector polynomials, 8 gate types
    oc = OracleCircuit(orc, pack)
    assert oc.verify(oc.prove(wires, pis)) == 0
    for row, col in ((6, 6), (6, 7), (14, 4), (14, 5)):      # output wires of op 0 of an ArithmeticExtension / MulExtension row
        w = wires.copy(); w[col, row] = (int(w[col, row]) + 1) % 0xFFFFFFFF00000001
        assert oc.verify(oc.prove(w, pis)) == 3, (row, col)
    oc.close()
    # without the Poseidon gate everything fits one selector polynomial
    pack, wires, pis = pkg.synth_circuit(6, num_wires=40, num_routed=24, num_public_inputs=3, seed=7, ext_arith=True)
    assert int(pack[5]) == 1 and int(pack[16]) == 6
    oc = OracleCircuit(orc, pack)
    assert oc.verify(oc.prove(wires, pis)) == 0
    oc.close()


def test_recursion_gate_set_circuit(pkg, orc):
    """Reducing / ReducingExtension / RandomAccess / Exponentiation / PoseidonMds / CosetInterpolation rows (rows 7, 15,
    23, 31, 39, 47, ...), all fourteen gate types together: four selector polynomials under the builder's grouping rule."""
    pack, wires, pis = pkg.synth_circuit(8, seed=9, base_sum=True, poseidon=True, ext_arith=True, recursion=True)
    assert int(pack[5]) == 4 and int(pack[16]) == 14
    oc = OracleCircuit(orc, pack)
    assert oc.verify(oc.prove(wires, pis)) == 0
    P = 0xFFFFFFFF00000001
    # (row, wire): Reducing output, ReducingExtension inner accumulator, RandomAccess claimed element and a bit wire,
    # Exponentiation intermediate and output, PoseidonMds output, CosetInterpolation value / intermediate / shifted point / shift
    for row, col in ((7, 0), (15, 6 + 2 * 32 + 3), (23, 1), (23, 74), (31, 2 + 66 + 5), (31, 1 + 66), (39, 24 + 7),
                     (47, 35), (47, 38), (47, 45), (47, 0), (47, 9)):
        w = wires.copy(); w[col, row] = (int(w[col, row]) + 1) % P
        assert oc.verify(oc.prove(w, pis)) == 3, (row, col)
    oc.close()
    # without Poseidon / BaseSum / extension arithmetic, narrower rows
    pack, wires, pis = pkg.synth_circuit(7, num_wires=80, num_routed=48, num_public_inputs=2, seed=10, recursion=True)
    oc = OracleCircuit(orc, pack)
    assert oc.verify(oc.prove(wires, pis)) == 0
    oc.close()


def test_hint_trailer_is_ignored_by_the_prover(pkg, orc):
    """A pack with a witness-hint trailer (stage s1 only) loads in the restatement, and the circuit whose operation inputs
    come from those generators is satisfied by the generator's witness."""
    kw = dict(seed=11, poseidon=True, base_sum=True, ext_arith=True, recursion=True)
    plain, _, _ = pkg.synth_circuit(7, **kw)
    pack, wires, pis = pkg.synth_circuit(7, hints=True, **kw)
    body = plain.size - (2 + pis.size)                   # both packs end with the public-input cell trailer ("PUBI1")
    assert int(plain[body]) == 0x3149425550 and int(plain[body + 1]) == pis.size
    assert pack.size > plain.size and int(pack[body]) == 0x31544E4948 and int(pack[-pis.size - 2]) == 0x3149425550
    oc = OracleCircuit(orc, pack)
    assert oc.verify(oc.prove(wires, pis)) == 0
    oc.close()
    import pytest
    with pytest.raises(ValueError):
        OracleCircuit(orc, pack[:-1])                    # a torn trailer is refused
