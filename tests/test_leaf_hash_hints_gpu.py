"""Hash hints through stage s1 on the device: the PartialWitness with the 732 host-computed sponge states makes the SAME witness and
the same proof bytes as without them (and as the oracle's), in fewer dependency levels — the 61 hash rows run side by side and are
checked against their hints; a wrong hint, or inputs the circuit does not satisfy, are QPGPU_EUNSAT; lockstep batches mix the cases."""
import numpy as np
import pytest

import leaf_cases as lc
import oracle_binding as ob

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L(pkg):
    return pkg.leaf


def test_same_witness_fewer_levels(pkg, gpu, orc, L):
    leaf = L.LeafCircuit(min_degree_bits=13)
    plain = L.LeafProver(pkg, gpu, leaf)
    hinted = L.LeafProver(pkg, gpu, leaf, hash_hints=True)
    oc = ob.OracleCircuit(orc, leaf.pack)
    for x in (lc.dummy_inputs(L), lc.test_inputs(L, 0), lc.real_inputs(L, depth=1, seed=2), lc.real_inputs(L, depth=9, seed=4), lc.real_inputs(L, depth=16, seed=9)):
        p0, pis0 = plain.prove(x)
        p1, pis1 = hinted.prove(x)
        assert np.array_equal(plain.witness(), hinted.witness()) and p0 == p1 and pis0.tolist() == pis1.tolist()
        cells, values, pis = leaf.commit(x)
        rc, wires, _ = orc.generate_witness(leaf.pack, cells, values, pis)
        assert rc == orc.WIT_OK and oc.prove(wires, pis) == p1
    l0, l1 = plain.circ.witness_info()[1], hinted.circ.witness_info()[1]
    assert l0 == 120 and l1 < 60, (l0, l1)                                 # the plan built for the assignment list is the shallower one
    # a wrong hint is a target set twice; inputs the circuit does not satisfy stay unsatisfiable under honest hints
    x = lc.real_inputs(L, depth=5, seed=6)
    cells, values, pis = leaf.commit(x, hash_hints=True)
    d = gpu.alloc(135 * 8192 * 8)
    for k in (299 + 5, 299 + 400, cells.size - 1):
        bad = values.copy(); bad[k] ^= np.uint64(1)
        with pytest.raises(pkg.QpGpuError) as e:
            hinted.circ.generate_witness_partial_dev(cells, bad, pis, d)
        assert e.value.code == -4 and "set twice with different values" in str(e.value)
    y = x.copy(); y.secret[3] ^= 1
    with pytest.raises(pkg.QpGpuError) as e:
        hinted.prove(y)
    assert e.value.code == -4
    # lockstep batch: honest, one wrong hint, honest
    xs = [lc.real_inputs(L, depth=2 + i, seed=20 + i) for i in range(3)]
    com = [leaf.commit(v, hash_hints=True) for v in xs]
    vals = np.stack([c[1] for c in com]); vals[1, 299 + 77] ^= np.uint64(1)
    circ = pkg.Circuit(gpu, leaf.pack, max_batch=3)
    db = gpu.alloc(3 * 135 * 8192 * 8)
    assert circ.generate_witness_partial_batch_dev(com[0][0], vals, np.stack([c[2] for c in com]), db) == [0, -4, 0]
    got = db.download().reshape(3, 135, 8192)
    for b in (0, 2):
        plain.generate_witness(xs[b])
        assert np.array_equal(got[b], plain.witness())
    circ.close(); db.free(scrub=True); d.free(scrub=True); oc.close(); plain.close(); hinted.close()


def test_hints_through_the_pool(pkg, gpu, orc, L):
    """The proving pool with the hinted assignment list: every proof equals the proof without hints."""
    leaf = L.LeafCircuit()
    xs = [lc.dummy_inputs(L), lc.test_inputs(L, 1), lc.real_inputs(L, depth=3), lc.real_inputs(L, depth=12, seed=7)]
    plain = L.LeafProver(pkg, gpu, leaf)
    want = [plain.prove(x)[0] for x in xs]
    plain.close()
    com = [leaf.commit(x, hash_hints=True) for x in xs]
    pool = pkg.ProvingPool(leaf.pack, workers=2, max_batch=4)
    pool.set_partial_cells(com[0][0])
    tickets = [(k, pool.submit_partial(com[k][1], com[k][2])) for _ in range(3) for k in range(len(xs))]
    for k, t in tickets:
        assert pool.wait(t) == want[k], k
    pool.close()


def test_entries_without_a_list_do_not_run_on_a_plan_built_for_hints(pkg, gpu, L):
    """After a hinted PartialWitness (stage s1's plan built FOR that assignment list) the same circuit handle generates a witness from
    a full matrix of free cells again — on the plan built without a list — and the two witnesses are equal."""
    leaf = L.LeafCircuit()
    x = lc.real_inputs(L, depth=4, seed=12)
    p = L.LeafProver(pkg, gpu, leaf, hash_hints=True)
    p.prove(x)
    assert p.circ.witness_info()[1] < 60
    w = p.witness()
    mask = p.circ.witness_free_mask(*w.shape)
    full = p.circ.generate_witness(np.where(mask == 1, w, 0).astype(np.uint64), leaf.commit(x)[2])
    assert np.array_equal(full, w) and p.circ.witness_info()[1] == 120
    p.prove(x)                                                              # and back
    assert np.array_equal(p.witness(), w) and p.circ.witness_info()[1] < 60
    p.close()
