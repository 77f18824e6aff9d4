"""How far would hash hints take stage s1 of a batch circuit? A probe, not a feature: the private-batch circuit over 8 leaf proofs
(complete in-circuit verification; 2^15 rows without blinding), its witness generated once, then generated again with the outputs of
every PoseidonGate row READ BACK OUT OF THAT WITNESS appended to the assignments as hints (what a host verifier that recorded its
permutations could hand in). Prints dependency levels and stage-s1 time both ways. usage: python tools/wrapper_hint_probe.py"""
import sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
import leaf_cases as lc
L, R = pkg.leaf, pkg.recursion
gpu = pkg.QpGpu(0)
leaf = L.LeafCircuit()
lp = L.LeafProver(pkg, gpu, leaf)
proofs = [lp.prove(x)[0] for x in lc.shared_tree_inputs(L, 6, depth=2, seed=31)] + [lp.prove(lc.dummy_inputs(L))[0]] * 2
ver = pkg.Verifier(leaf.pack, circuit=lp.circ)
w = R.WrapperCircuit(leaf.pack, ver, 8, num_routed_wires=60, logic="private_batch", verify=True)
circ = pkg.Circuit(gpu, w.pack)
n = 1 << w.info["degree_bits"]
d = gpu.alloc(135 * n * 8)
com = w.commit(proofs, preimages=np.arange(32, dtype=np.uint64).reshape(8, 4), derive_public_inputs=True)
cells, values = com[0], com[1]


def timed(c, v, reps=5):
    circ.generate_witness_partial_batch_dev(c, v[None], None, d); gpu.sync()
    t = time.perf_counter()
    for _ in range(reps):
        circ.generate_witness_partial_batch_dev(c, v[None], None, d)
    gpu.sync()
    return (time.perf_counter() - t) / reps * 1e3


t0 = timed(cells, values)
w0 = d.download().reshape(135, n)
print("rows 2^%d, plain: %.3f ms, (instances, levels, free) = %s" % (w.info["degree_bits"], t0, circ.witness_info()))
rows = circ.gate_rows(4)
hc = np.array([int(r) * 135 + 12 + k for r in rows for k in range(12)], dtype=np.uint64)
hv = np.array([w0[12 + k, int(r)] for r in rows for k in range(12)], dtype=np.uint64)
c2, v2 = np.concatenate([cells, hc]), np.concatenate([values, hv])
t1 = timed(c2, v2)
w1 = d.download().reshape(135, n)
print("%d PoseidonGate rows hinted: %.3f ms, %s, witness equal %s" % (rows.size, t1, circ.witness_info(), np.array_equal(w0, w1)))
