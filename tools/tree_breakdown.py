"""Where the 64-leaf attesting tree (recursion.AttestingTree, zero-knowledge first level) spends its time on one GPU: whole tree,
then the first level's host commit, stage s1 and proving, and the second level's. usage: python tools/tree_breakdown.py"""
import sys, time; sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo')
import numpy as np
import __graft_entry__ as g; pkg = g.load_package()
import leaf_cases as lc
gpu = pkg.QpGpu(0)
L = pkg.leaf; R = pkg.recursion
tree = R.AttestingTree(pkg, gpu, per_batch=8, batches=8, zero_knowledge=True)
sp = lc.shared_tree_inputs(L, 48, depth=3, seed=9)
dm = lc.dummy_inputs(L)
xs = []
for b in range(8):
    s = sp[6*b:6*b+6]; s.insert(b % 7, dm); s.insert((3*b+1) % 8, dm); xs += s
tree.run(xs)
t=time.perf_counter(); leaves, l1, root = tree.run(xs); print("tree", time.perf_counter()-t, tree.times)
# first level pieces
pre = [tree.preimages(b) for b in range(8)]
t=time.perf_counter(); com1 = [tree.w1.commit(leaves[k*8:(k+1)*8], preimages=pre[k], device_blinding=True) for k in range(8)]; t1=time.perf_counter()-t
gpu.sync(); t=time.perf_counter(); st = R.generate_wrapper_witnesses(tree.w1_circ, tree.w1, com1, tree.d_wires); gpu.sync(); t2=time.perf_counter()-t
t=time.perf_counter(); pr = tree.w1_circ.prove_batch_dev([tree.d_wires.ptr + 8*k*tree.words[1] for k in range(8)], [c[2] for c in com1]); t3=time.perf_counter()-t
print("first level: commit (1 thread) %.4f  s1 %.4f  prove %.4f" % (t1, t2, t3))
c2 = tree.w2.commit(l1, aggregator_address=bytes(32))
t=time.perf_counter(); tree.w2_circ.generate_witness_partial_batch_dev(c2[0], c2[1][None], c2[2][None], tree.d_wires); gpu.sync(); t4=time.perf_counter()-t
t=time.perf_counter(); tree.w2_circ.prove_batch_dev([tree.d_wires.ptr], [c2[2]]); t5=time.perf_counter()-t
print("second level: s1 %.4f prove %.4f" % (t4, t5))

for name, circ in (("leaf", tree.leaf_circ), ("first level", tree.w1_circ), ("second level", tree.w2_circ)):
    gens, levels, free = circ.witness_info()
    print("%s: %d generator instances in %d dependency levels, %d free cells" % (name, gens, levels, free))
