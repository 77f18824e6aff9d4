"""Hash hints of the leaf front-end (include/qpgpu_leaf.h: qpgpu_leaf_circuit_hash_hint_cells, qpgpu_leaf_hash_hints) on the CPU: the 732
sponge states the host computes for the circuit's 8 hash call sites are what the rows' generators compute — the ORACLE's witness
generator, given the 299 assignments plus the hints, finds no target set twice with different values and produces the same witness;
a hint that is off by one is a conflict."""
import numpy as np
import pytest

import leaf_cases as lc


@pytest.fixture(scope="module")
def L(pkg):
    return pkg.leaf


@pytest.fixture(scope="module")
def full(L):
    return L.LeafCircuit()


def test_hint_cells_are_the_hash_rows_outputs(L, full, pkg):
    cells = full.hash_hint_cells
    assert cells.size == L.HASH_HINTS == 12 * 61 + 64 and full.info["rows_poseidon2"] == 61
    rows = cells // 135
    assert len(set(cells[:732].tolist())) == 732 and rows.max() < (1 << full.info["degree_bits"])
    # the same call sites, cells moved with the rows, in a padded build
    padded = L.LeafCircuit(min_degree_bits=13)
    assert padded.hash_hint_cells.size == cells.size
    with pytest.raises(ValueError):
        L.LeafCircuit(fragment=L.FRAGMENT_NULLIFIER).hash_hint_cells


@pytest.mark.parametrize("case", ["dummy", "test_inputs_0", "test_inputs_1", "depth 1", "depth 7", "depth 16"])
def test_hints_agree_with_the_generators(L, full, orc, case):
    x = {"dummy": lambda: lc.dummy_inputs(L), "test_inputs_0": lambda: lc.test_inputs(L, 0), "test_inputs_1": lambda: lc.test_inputs(L, 1),
         "depth 1": lambda: lc.real_inputs(L, depth=1, seed=2), "depth 7": lambda: lc.real_inputs(L, depth=7, seed=3),
         "depth 16": lambda: lc.real_inputs(L, depth=16, seed=9)}[case]()
    c0, v0, p0 = full.commit(x)
    c1, v1, p1 = full.commit(x, hash_hints=True)
    assert c1.size == c0.size + L.HASH_HINTS and np.array_equal(c1[:c0.size], c0) and np.array_equal(v1[:v0.size], v0) and np.array_equal(p0, p1)
    rc0, w0, _ = orc.generate_witness(full.pack, c0, v0, p0)
    rc1, w1, _ = orc.generate_witness(full.pack, c1, v1, p1)
    assert rc0 == rc1 == orc.WIT_OK and np.array_equal(w0, w1)
    # every hinted value is the witness's value at that cell
    assert [int(w0[int(c) % 135, int(c) // 135]) for c in full.hash_hint_cells] == v1[c0.size:].tolist()
    for k in (0, 100, 731, 795):
        bad = v1.copy(); bad[c0.size + k] = (int(bad[c0.size + k]) + 1) % lc.P
        assert orc.generate_witness(full.pack, c1, bad, p1)[0] == orc.WIT_CONFLICT


def test_hints_refuse_what_fill_witness_refuses(L, full):
    x = lc.test_inputs(L, 0)
    x.zk_merkle_depth = 17
    with pytest.raises(ValueError) as e:
        full.commit(x, hash_hints=True)
    assert "ZK Merkle proof depth" in str(e.value)
