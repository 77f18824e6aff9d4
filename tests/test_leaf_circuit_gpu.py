"""The leaf path end to end through the C ABI on the device (SURVEY.md section 8 rows a1 + a2 + s1..s12 on one circuit):

    CircuitInputs -> qpgpu_leaf_commit (fill_witness + target map) -> qpgpu_generate_witness_partial*_dev (stage s1)
                  -> qpgpu_prove_dev / the proving pool (stages s2..s12) -> proof bytes

on the Wormhole leaf circuit restated by qpgpu_leaf_circuit_build. Checked against the oracle on the same inputs: the device
witness equals oracle/witness.c's cell for cell, the proof equals oracle/prove.c's byte for byte, both verifiers accept; the
public inputs parsed out of the proof are the reference's; the block-header fragment's proof carries DEFAULT_BLOCK_HASHES[i]
(wormhole/tests/test-helpers/src/lib.rs:210-219) computed by Poseidon2 gate rows on the device; the reference's negative cases
(block_header_tests.rs:31-95, nullifier_tests.rs:53-58) come back as QPGPU_EUNSAT "set twice with different values"."""
import ctypes

import numpy as np
import pytest

import leaf_cases as lc
import oracle_binding as ob

pytestmark = pytest.mark.gpu


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


def oracle_side(orc, circuit, x):
    cells, values, pis = circuit.commit(x)
    rc, wires, _ = orc.generate_witness(circuit.pack, cells, values, pis)
    assert rc == orc.WIT_OK
    oc = ob.OracleCircuit(orc, circuit.pack)
    proof = oc.prove(wires, pis)
    assert oc.verify(proof) == 0
    oc.close()
    return wires, proof, pis


def test_circuit_inputs_to_proof_bytes(pkg, gpu, orc, L, full, prover):
    ver = pkg.Verifier(full.pack, circuit=prover.circ)
    cases = [("bench input (build_dummy_circuit_inputs)", lc.dummy_inputs(L)), ("test_inputs_0", lc.test_inputs(L, 0)), ("test_inputs_1", lc.test_inputs(L, 1)),
             ("spend, depth 3", lc.real_inputs(L, depth=3)), ("spend, depth 16", lc.real_inputs(L, depth=16, seed=9))]
    for name, x in cases:
        want_wires, want_proof, want_pis = oracle_side(orc, full, x)
        proof, pis = prover.prove(x)
        assert np.array_equal(prover.witness(), want_wires), name            # stage s1 on the device == generate_partial_witness on the CPU
        assert proof == want_proof, name                                     # stages s2..s12: byte parity
        assert ver.verify(proof), name
        got = lc.proof_public_inputs(proof, 21)
        assert got.tolist() == want_pis.tolist() == pis.tolist(), name
        # the reference's PublicCircuitInputs layout (wormhole/inputs/src/lib.rs:68-80)
        assert got[4:8].tolist() == lc.digest_felts(x.get32("nullifier")) and got[16:20].tolist() == lc.digest_felts(x.get32("block_hash"))
        assert int(got[20]) == x.block_number and got[:4].tolist() == [x.asset_id, x.output_amount_1, x.output_amount_2, x.volume_fee_bps]
    ver.close()


def test_unsatisfiable_inputs_name_the_target(pkg, L, full, prover):
    base = lc.real_inputs(L, depth=2)
    def expect_unsat(x):
        with pytest.raises(pkg.QpGpuError) as e:
            prover.generate_witness(x)
        assert e.value.code == -4 and "set twice with different values" in str(e.value), str(e.value)
    for name in ("secret", "nullifier", "block_hash", "state_root", "zk_tree_root"):
        x = base.copy(); getattr(x, name)[0] ^= 1
        expect_unsat(x)
    x = lc.test_inputs(L, 0); x.secret[31] ^= 0x10            # dummies still bind the unspendable account
    expect_unsat(x)
    x = base.copy(); x.output_amount_1 = 250                    # the fee relation's 48-bit range check
    expect_unsat(x)
    x = lc.test_inputs(L, 0); x.output_amount_2 = 3             # zero block hash with a positive output is no dummy
    expect_unsat(x)
    prover.generate_witness(base)                               # and the handle is fine afterwards
    # a proof made from an inconsistent full witness is caught by the optional witness check (row named)
    prover.circ.set_witness_check(True)
    w = prover.witness().copy()
    cell = int(full.target_map[237])                           # asset_id
    w[cell % 135, cell // 135] ^= 1
    with pytest.raises(pkg.QpGpuError) as e:
        prover.circ.prove(w, full.commit(base)[2])
    assert e.value.code == -4
    prover.circ.set_witness_check(False)


def test_batched_partial_witnesses(pkg, gpu, orc, L, full):
    """qpgpu_generate_witness_partial_batch_dev: one pass for several PartialWitnesses; a failing one fails alone."""
    circ = pkg.Circuit(gpu, full.pack, max_batch=8)
    xs = [lc.dummy_inputs(L), lc.test_inputs(L, 0), lc.real_inputs(L, depth=4), lc.test_inputs(L, 1), lc.real_inputs(L, depth=1, seed=2), lc.dummy_inputs(L)]
    bad = lc.real_inputs(L, depth=4); bad.nullifier[9] ^= 2
    xs.insert(3, bad)
    com = [full.commit(x) for x in xs]
    cells = com[0][0]
    assert all(np.array_equal(c[0], cells) for c in com)        # every proof of the circuit assigns the same targets
    circ.witness_partial_prepare(cells, len(xs))
    nw, n = 135, 1 << full.info["degree_bits"]
    d = gpu.alloc(len(xs) * nw * n * 8)
    status = circ.generate_witness_partial_batch_dev(cells, np.stack([c[1] for c in com]), np.stack([c[2] for c in com]), d)
    assert status == [0, 0, 0, -4, 0, 0, 0]
    assert "witness 3" in gpu.last_error() and "set twice with different values" in gpu.last_error()
    got = d.download().reshape(len(xs), nw, n)
    for k, x in enumerate(xs):
        if k == 3:
            continue
        want, _, _ = oracle_side(orc, full, x)
        assert np.array_equal(got[k], want), k
    # and straight into a lockstep batch of proofs
    good = [k for k in range(len(xs)) if k != 3]
    proofs = circ.prove_batch_dev([d.ptr + k * nw * n * 8 for k in good], [com[k][2] for k in good])
    for k, pf in zip(good, proofs):
        assert pf == oracle_side(orc, full, xs[k])[1], k
    d.free(scrub=True); circ.close()


def test_fragment_circuits_on_the_reference_vectors(pkg, gpu, orc, L):
    """The reference's fragment tests on the device: the block-header fragment's proofs carry DEFAULT_BLOCK_HASHES[i] as public
    inputs (computed by the sponge's six Poseidon2 gate rows, bound unconditionally), the unspendable-account fragment proves
    the address vectors, and each of the reference's negative cases is QPGPU_EUNSAT."""
    frag = L.LeafCircuit(fragment=L.FRAGMENT_BLOCK_HEADER)
    pr = L.LeafProver(pkg, gpu, frag)
    for i in (0, 1):
        x = lc.header_inputs(L, i)
        proof, pis = pr.prove(x)
        assert proof == oracle_side(orc, frag, x)[1]
        assert lc.proof_public_inputs(proof, 5)[:4].tobytes() == lc.header_kat(i)[6]          # the reference's 32 bytes
        assert int(lc.proof_public_inputs(proof, 5)[4]) == lc.header_kat(i)[1]
    for name in ("parent_hash", "state_root", "block_hash", "extrinsics_root", "digest"):   # block_header_tests.rs:31-95
        x = lc.header_inputs(L, 0); getattr(x, name)[0] = (getattr(x, name)[0] + 1) % 256
        with pytest.raises(pkg.QpGpuError) as e:
            pr.generate_witness(x)
        assert e.value.code == -4 and "set twice with different values" in str(e.value), name
    x = lc.header_inputs(L, 0); x.block_number += 1
    with pytest.raises(pkg.QpGpuError):
        pr.generate_witness(x)
    pr.close()
    frag = L.LeafCircuit(fragment=L.FRAGMENT_UNSPENDABLE_ACCOUNT)
    pr = L.LeafProver(pkg, gpu, frag)
    for k, kat in enumerate(lc.KATS["address_kats"]):
        x = L.LeafInputs()
        x.set32("secret", bytes.fromhex(kat["secret"])).set32("unspendable_account", bytes.fromhex(kat["address"]))
        proof, _ = pr.prove(x)
        if k == 0:
            assert proof == oracle_side(orc, frag, x)[1]
        x.unspendable_account[0] ^= 1
        with pytest.raises(pkg.QpGpuError) as e:
            pr.generate_witness(x)
        assert e.value.code == -4
    pr.close()
    frag = L.LeafCircuit(fragment=L.FRAGMENT_NULLIFIER)
    pr = L.LeafProver(pkg, gpu, frag)
    x = lc.test_inputs(L, 0)
    proof, pis = pr.prove(x)
    assert proof == oracle_side(orc, frag, x)[1] and pis.tolist() == lc.digest_felts(x.get32("nullifier"))
    x.transfer_count += 1
    with pytest.raises(pkg.QpGpuError):
        pr.generate_witness(x)
    pr.close()


def test_bench_shape_and_other_layout(pkg, gpu, orc, L):
    """The bench's circuit (padded to 2^13 rows) and the second Poseidon2 gate layout: byte parity on the bench input."""
    for kw in (dict(min_degree_bits=13), dict(p2_layout=[12, 0, 0xFFFFFFFF, 0, 94, 24, 46, 0, 0, 130])):
        c = L.LeafCircuit(**kw)
        pr = L.LeafProver(pkg, gpu, c)
        x = lc.dummy_inputs(L)
        proof, pis = pr.prove(x)
        assert proof == oracle_side(orc, c, x)[1], kw
        pr.close()


def test_leaf_circuit_under_the_poseidon2_hasher(pkg, orc, L):
    """Poseidon2 (qp-poseidon-core's parameters) as the proof-system hasher: the public-input hash is built from Poseidon2 gate rows."""
    qp = pkg.poseidon2_qp_params()
    pkg.set_hasher_poseidon2(*qp); orc.select_poseidon2(*qp)
    try:
        c = L.LeafCircuit(inner_hasher=1)
        g2 = pkg.QpGpu(0)
        pr = L.LeafProver(pkg, g2, c)
        x = lc.real_inputs(L, depth=2)
        proof, pis = pr.prove(x)
        assert proof == oracle_side(orc, c, x)[1]
        pr.close(); g2.close()
    finally:
        pkg.set_hasher_poseidon(); orc.select_poseidon()


def test_public_inputs_read_out_of_the_witness(pkg, gpu, orc, L, full, prover):
    """plonky2's prove() does not take public inputs, it reads them out of the generated witness
    (`partition_witness.get_targets(&prover_data.public_inputs)`). The same through the C ABI: stage s1 with public_inputs = NULL —
    the public-input targets are then just what fill_witness assigns and the generators compute — and
    qpgpu_witness_public_inputs_dev afterwards. Same witness, same public inputs, same proof as with the public inputs handed in;
    a synthetic pack whose PublicInputGate takes a host-computed hash cannot derive them (QPGPU_EINVAL)."""
    xs = [lc.real_inputs(L, depth=6, seed=77), lc.dummy_inputs(L), lc.test_inputs(L, 1)]
    com = [full.commit(x) for x in xs]
    cells = com[0][0]; vals = np.stack([c[1] for c in com]); pis = np.stack([c[2] for c in com])
    nw, n = prover.shape
    d = gpu.alloc(3 * nw * n * 8)
    circ = pkg.Circuit(gpu, full.pack, max_batch=3)
    assert circ.generate_witness_partial_batch_dev(cells, vals, pis, d) == [0, 0, 0]
    want = d.download().copy()
    assert circ.generate_witness_partial_batch_dev(cells, vals, None, d) == [0, 0, 0]
    assert np.array_equal(d.download(), want)
    got = circ.witness_public_inputs_dev(d, 3)
    assert got.tolist() == pis.tolist()
    proofs = circ.prove_batch_dev([d.ptr + 8 * k * nw * n for k in range(3)], list(got))
    for k, x in enumerate(xs):
        assert proofs[k] == prover.prove(x)[0]
    # an input the circuit has no witness for is still named (the block hash the header does not hash to)
    bad = xs[0].copy(); bad.block_hash[3] ^= 1
    c_ = full.commit(bad)
    st = circ.generate_witness_partial_batch_dev(c_[0], c_[1][None], None, d)
    assert st == [-4] and "set twice with different values" in gpu.last_error()
    circ.close(); d.free(scrub=True)
    # a pack whose PublicInputGate wires are NOT fed by an in-circuit hash (the synthetic test circuits)
    sp, wires, spis = pkg.synth_circuit(degree_bits=6, num_public_inputs=3, seed=2) if hasattr(pkg, "synth_circuit") else (None, None, None)
    if sp is not None:
        sc = pkg.Circuit(gpu, sp)
        ds = gpu.alloc(wires.size * 8)
        with pytest.raises(pkg.QpGpuError) as e:
            sc.generate_witness_partial_batch_dev(np.zeros(1, dtype=np.uint64), wires[:1, :1], None, ds)
        assert "can only be derived" in str(e.value)
        sc.close(); ds.free()
