"""The Wormhole leaf circuit restated on the native builder (csrc/builder.hpp, csrc/leaf_circuit.cpp; include/qpgpu_leaf.h),
checked on the CPU: the library builds the circuit (host code), the ORACLE generates the witness from the PartialWitness
(oracle/witness.c, plonky2's generate_partial_witness restated), proves and verifies. No GPU; tests/test_leaf_circuit_gpu.py runs
the same cases through the device path and compares bytes.

What pins what: the circuits' digests are computed by Poseidon2 gate rows and must equal the reference-held known-answer vectors
(5 addresses, 2 block hashes) for the witness to be consistent at all — a fragment circuit with the unconditional binding cannot
be satisfied otherwise; the negative cases are the reference's own (wormhole/tests/src/circuit/block_header_tests.rs:34-95,
unspendable_account_tests.rs, nullifier_tests.rs): "set twice with different values"."""
import numpy as np
import pytest

import leaf_cases as lc
import oracle_binding as ob


@pytest.fixture(scope="module")
def L(pkg):
    return pkg.leaf


@pytest.fixture(scope="module")
def full(L):
    return L.LeafCircuit()


def run(orc, circuit, x):
    cells, values, pis = circuit.commit(x)
    rc, wires, bad = orc.generate_witness(circuit.pack, cells, values, pis)
    return rc, wires, pis, bad


def prove_verify(orc, circuit, wires, pis):
    oc = ob.OracleCircuit(orc, circuit.pack)
    try:
        proof = oc.prove(wires, pis)
        assert oc.verify(proof) == 0
        flipped = bytearray(proof); flipped[len(proof) // 3] ^= 1
        assert oc.verify(bytes(flipped)) != 0
    finally:
        oc.close()
    return proof


def test_shape_of_the_restated_circuit(L, full, pkg):
    h = pkg.pack_header(full.pack)
    i = full.info
    # standard_recursion_config (common/src/circuit.rs:378-380), 21 public inputs (wormhole/inputs/src/lib.rs:33)
    assert (h["num_wires"], h["num_routed_wires"], h["num_constants"], h["num_challenges"], h["rate_bits"], h["cap_height"],
            h["proof_of_work_bits"], h["num_query_rounds"], h["zero_knowledge"], h["num_public_inputs"]) == (135, 80, 2, 2, 3, 4, 16, 28, 0, 21)
    # the application hashes: 7 -> 1, 4 -> 1, 9 -> 2, 4 -> 1, 8 -> 2, 45 -> 6, 16 levels x 3 permutations
    assert i["rows_poseidon2"] == 61
    # range checks: 7 x 32 bits, 14, 48, the depth bound and 16 level comparisons on 5 bits, 16 positions on 2 bits, block number
    assert i["rows_base_sum"] == 7 + 1 + 1 + 1 + 16 + 16 + 1
    assert i["rows_poseidon"] == 3 and i["rows_public_input"] == 1            # hash of 21 public inputs: 3 absorptions
    assert i["rows_before_padding"] == sum(i[k] for k in ("rows_arithmetic", "rows_base_sum", "rows_poseidon2", "rows_poseidon", "rows_constant", "rows_public_input"))
    assert (1 << i["degree_bits"]) == i["rows_before_padding"] + i["rows_noop"] and i["degree_bits"] == h["degree_bits"]
    # is_equal: 4 per Merkle level + 4 block-hash limbs + 2 outputs
    assert i["free_standing_generators"] == 4 * 16 + 6
    # every logical target of CircuitTargets has a wire cell, public inputs sit where the trailer says
    assert (full.target_map != L.NO_CELL).all()
    pi_cells = pkg.pack_public_input_cells(full.pack)
    lt = {"asset": 237, "out1": 239, "out2": 240, "fee": 241, "nullifier": 0, "exit1": 242, "exit2": 246, "block_hash": 250, "block_number": 258}
    order = [lt["asset"], lt["out1"], lt["out2"], lt["fee"]] + [lt["nullifier"] + k for k in range(4)] + [lt["exit1"] + k for k in range(4)] + \
            [lt["exit2"] + k for k in range(4)] + [lt["block_hash"] + k for k in range(4)] + [lt["block_number"]]
    assert [int(full.target_map[t]) for t in order] == [int(c) for c in pi_cells]
    # padding to the size the reference states for its circuits (>= 2^12, common/src/circuit.rs:463-467)
    padded = L.LeafCircuit(min_degree_bits=12)
    assert padded.info["degree_bits"] == 12 and padded.info["rows_before_padding"] == i["rows_before_padding"]
    assert (padded.target_map == full.target_map).all()


def test_bench_inputs_prove_and_verify(orc, L, full):
    x = lc.dummy_inputs(L)
    rc, wires, pis, _ = run(orc, full, x)
    assert rc == orc.WIT_OK
    assert pis.tolist() == [0, 0, 0, 10] + [0] * 17
    proof = prove_verify(orc, full, wires, pis)
    assert lc.proof_public_inputs(proof, 21).tolist() == pis.tolist()


def test_reference_test_inputs(orc, L, full):
    for i in (0, 1):
        x = lc.test_inputs(L, i)
        rc, wires, pis, _ = run(orc, full, x)
        assert rc == orc.WIT_OK, i
        # the reference's public-input order (wormhole/inputs/src/lib.rs:68-80)
        assert pis.tolist() == [0, 0, 0, 10] + lc.digest_felts(x.get32("nullifier")) + lc.digest_felts(lc.DEFAULT_EXIT_ACCOUNT) + [0] * 8 + [lc.header_kat(i)[1]]
        if i == 0:
            prove_verify(orc, full, wires, pis)


def test_digests_in_the_trace_are_the_reference_vectors(orc, L, full):
    """A dummy proof still computes every hash: with the KAT header as the private header inputs, the cells behind the block-hash
    comparison hold DEFAULT_BLOCK_HASHES[i], computed by the 6 Poseidon2 gate rows of that sponge; the unspendable-account rows hold
    the address KAT (they are copy-connected to the account target, so a wrong value could not even be generated)."""
    nw = 135
    for i in (0, 1):
        x = lc.test_inputs(L, i)
        hk = lc.header_kat(i)
        x.set32("parent_hash", hk[0]).set32("zk_tree_root", hk[4])     # the vector's header has a zero tree root; a dummy's root is not bound
        rc, wires, pis, _ = run(orc, full, x)
        assert rc == orc.WIT_OK
        want = lc.digest_felts(hk[6])
        # the rows of the Poseidon2 gate (selector value = its index in the gate list): find the row whose outputs are the hash
        outs = {tuple(int(v) for v in wires[12:16, r]) for r in range(wires.shape[1])}
        assert tuple(want) in outs, i
        acct = lc.digest_felts(bytes.fromhex(lc.KATS["address_kats"][i]["address"]))
        assert tuple(acct) in outs
        cell = int(full.target_map[10])       # QPGPU_LT_UNSPENDABLE_ACCOUNT_ID
        assert [int(wires[(int(full.target_map[10 + k]) % nw), int(full.target_map[10 + k]) // nw]) for k in range(4)] == acct and cell != L.NO_CELL


def test_a_real_spend(orc, L, full):
    for depth in (0, 1, 3, 16):
        x = lc.real_inputs(L, depth=depth, seed=depth)
        assert L._lib().qpgpu_leaf_is_not_dummy(__import__("ctypes").byref(x)) == 1
        rc, wires, pis, bad = run(orc, full, x)
        assert rc == orc.WIT_OK, (depth, bad // 135, bad % 135)
        assert pis[16:20].tolist() == lc.digest_felts(x.get32("block_hash")) and int(pis[1]) == 200 and int(pis[2]) == 97
        if depth in (3, 16):
            prove_verify(orc, full, wires, pis)


def test_unsatisfiable_inputs_are_conflicts(orc, L, full):
    import ctypes
    base = lc.real_inputs(L, depth=2)
    err = ctypes.create_string_buffer(160)
    assert L._lib().qpgpu_leaf_check_constraints(ctypes.byref(base), err) == 0

    def conflict(mutate, dummy=False):
        x = (lc.test_inputs(L, 0) if dummy else base).copy()
        mutate(x)
        rc, _, _, bad = run(orc, full, x)
        # the library's native constraint check (host, no circuit) agrees that no proof exists
        assert L._lib().qpgpu_leaf_check_constraints(ctypes.byref(x), err) == -4, err.value
        return rc, bad
    def flip(name, k=0):
        def f(x):
            getattr(x, name)[k] ^= 1
        return f
    # the reference's negative tests observe these as "set twice with different values" panics
    assert conflict(flip("secret"))[0] == orc.WIT_CONFLICT                       # unspendable account no longer H(H(salt || secret))
    assert conflict(flip("secret"), dummy=True)[0] == orc.WIT_CONFLICT           # enforced for dummies too
    assert conflict(flip("nullifier"))[0] == orc.WIT_CONFLICT
    assert conflict(flip("block_hash"))[0] == orc.WIT_CONFLICT
    assert conflict(flip("state_root"))[0] == orc.WIT_CONFLICT                   # header preimage changed, claimed hash not
    assert conflict(flip("zk_tree_root"))[0] == orc.WIT_CONFLICT
    assert conflict(flip("zk_merkle_siblings", 40))[0] == orc.WIT_CONFLICT       # path no longer leads to the root
    def wrong_position(x):
        x.zk_merkle_positions[1] = (x.zk_merkle_positions[1] + 1) % 4
    assert conflict(wrong_position)[0] == orc.WIT_CONFLICT
    def overspend(x):
        x.output_amount_1 = 250                                                   # (250 + 97) * 10000 > 300 * 9990: the 48-bit range check
    assert conflict(overspend)[0] == orc.WIT_CONFLICT
    def fee(x):
        x.volume_fee_bps = 10001                                                  # 10000 - fee wraps: the 14-bit range check
    assert conflict(fee)[0] == orc.WIT_CONFLICT
    # a dummy may carry any nullifier and header (the bindings are multiplied by is_not_dummy = 0) ...
    x = lc.test_inputs(L, 0); x.nullifier[5] ^= 0x40; x.state_root[3] ^= 1
    assert run(orc, full, x)[0] == orc.WIT_OK
    # ... but a zero block hash with a positive output is no dummy
    x = lc.test_inputs(L, 0); x.output_amount_1 = 1
    assert run(orc, full, x)[0] == orc.WIT_CONFLICT


def test_block_header_fragment_carries_the_reference_block_hashes(orc, L):
    """BlockHeader::circuit alone (block_header_tests.rs:8-29): the proof's public inputs ARE DEFAULT_BLOCK_HASHES[i], and the
    unconditional binding means the six Poseidon2 gate rows computed exactly that from the header."""
    frag = L.LeafCircuit(fragment=L.FRAGMENT_BLOCK_HEADER)
    assert frag.info["rows_poseidon2"] == 6 and frag.info["rows_base_sum"] == 1
    for i in (0, 1):
        x = lc.header_inputs(L, i)
        rc, wires, pis, bad = run(orc, frag, x)
        assert rc == orc.WIT_OK, (i, bad)
        assert pis.tolist() == lc.digest_felts(lc.header_kat(i)[6]) + [lc.header_kat(i)[1]]
        proof = prove_verify(orc, frag, wires, pis)
        assert lc.proof_public_inputs(proof, 5)[:4].tobytes() == lc.header_kat(i)[6]      # the reference's 32 bytes, in the proof
    # block_header_tests.rs:31-95: invalid parent hash / state root / block number / block hash / extrinsics root / digest
    def bad_case(mut):
        x = lc.header_inputs(L, 0); mut(x)
        return run(orc, frag, x)[0]
    def m(name):
        def f(x):
            getattr(x, name)[0] = (getattr(x, name)[0] + 1) % 256
        return f
    for name in ("parent_hash", "state_root", "block_hash", "extrinsics_root", "digest"):
        assert bad_case(m(name)) == orc.WIT_CONFLICT, name
    def bn(x):
        x.block_number += 1
    assert bad_case(bn) == orc.WIT_CONFLICT


def test_unspendable_account_fragment_on_the_address_vectors(orc, L):
    """UnspendableAccount::circuit alone on the five address vectors (unspendable_account_tests.rs:9-66)."""
    frag = L.LeafCircuit(fragment=L.FRAGMENT_UNSPENDABLE_ACCOUNT)
    assert frag.info["rows_poseidon2"] == 2
    for k, kat in enumerate(lc.KATS["address_kats"]):
        x = L.LeafInputs()
        x.set32("secret", bytes.fromhex(kat["secret"])).set32("unspendable_account", bytes.fromhex(kat["address"]))
        rc, wires, pis, _ = run(orc, frag, x)
        assert rc == orc.WIT_OK, k
        if k == 0:
            prove_verify(orc, frag, wires, pis)
        x.unspendable_account[31] ^= 1
        assert run(orc, frag, x)[0] == orc.WIT_CONFLICT
        x.unspendable_account[31] ^= 1; x.secret[0] ^= 1
        assert run(orc, frag, x)[0] == orc.WIT_CONFLICT


def test_nullifier_fragment(orc, L):
    frag = L.LeafCircuit(fragment=L.FRAGMENT_NULLIFIER)
    assert frag.info["rows_poseidon2"] == 3
    x = lc.test_inputs(L, 0)
    rc, wires, pis, _ = run(orc, frag, x)
    assert rc == orc.WIT_OK and pis.tolist() == lc.digest_felts(x.get32("nullifier"))
    prove_verify(orc, frag, wires, pis)
    x.transfer_count += 1                                 # nullifier_tests.rs: a preimage that does not match the hash
    assert run(orc, frag, x)[0] == orc.WIT_CONFLICT


def test_other_gate_layout_and_inner_hasher(orc, L, pkg):
    """The Poseidon2 gate's wire layout is data (LAYOUT UNPINNED): the circuit builds and proves under the second layout too; and
    with Poseidon2 as the proof-system hasher the public-input hash is built from Poseidon2 gate rows."""
    alt = [12, 0, 0xFFFFFFFF, 0, 94, 24, 46, 0, 0, 130]
    circ = L.LeafCircuit(p2_layout=alt)
    assert pkg.pack_p2_layout(circ.pack)["w_output"] == 0
    x = lc.test_inputs(L, 1)
    rc, wires, pis, _ = run(orc, circ, x)
    assert rc == orc.WIT_OK
    prove_verify(orc, circ, wires, pis)
    qp = pkg.poseidon2_qp_params()
    pkg.set_hasher_poseidon2(*qp); orc.select_poseidon2(*qp)
    try:
        c2 = L.LeafCircuit(inner_hasher=1)
        assert c2.info["rows_poseidon"] == 0 and c2.info["rows_poseidon2"] == 64
        rc, wires, pis, _ = run(orc, c2, x)
        assert rc == orc.WIT_OK
        prove_verify(orc, c2, wires, pis)
    finally:
        pkg.set_hasher_poseidon(); orc.select_poseidon()
