"""Stage s1 with and without fused runs of narrow dependency levels (QPGPU_WITNESS_FUSE, read when a circuit's plan is built): the
leaf circuit at 2^13 rows (batch 1 and 32) and the zero-knowledge private-batch circuit over 8 leaves (batch 1 and 8).
usage: QPGPU_WITNESS_FUSE=0|1 python tools/witness_fuse_ab.py <label>"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import __graft_entry__ as ge
import leaf_cases as lc
pkg = ge.load_package()
gpu = pkg.QpGpu(0)
L, R = pkg.leaf, pkg.recursion
out = {"label": sys.argv[1], "fuse": os.environ.get("QPGPU_WITNESS_FUSE", "default")}


def timed(fn, reps):
    fn(); gpu.sync()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    gpu.sync()
    return round((time.perf_counter() - t) / reps * 1e3, 3)


leaf = L.LeafCircuit(min_degree_bits=13)
circ = pkg.Circuit(gpu, leaf.pack, max_batch=32)
xs = [lc.real_inputs(L, depth=1 + i % 16, seed=i, secret_index=i % 2) for i in range(32)]
com = [leaf.commit(x) for x in xs]
d = gpu.alloc(32 * 135 * 8192 * 8)
cells = com[0][0]; vals = np.stack([c[1] for c in com]); pis = np.stack([c[2] for c in com])
circ.witness_partial_prepare(cells, 32)
out["leaf_levels"] = int(circ.witness_info()[1])
out["leaf_batch1_ms"] = timed(lambda: circ.generate_witness_partial_batch_dev(cells, vals[:1], pis[:1], d), 50)
out["leaf_batch32_ms"] = timed(lambda: circ.generate_witness_partial_batch_dev(cells, vals, pis, d), 50)
ref = d.download(135 * 8192)
out["leaf_checksum"] = int(np.bitwise_xor.reduce(ref))
d.free(); circ.close()
# the private-batch circuit over 8 leaves of the unpadded leaf circuit
small = L.LeafCircuit()
lp = L.LeafProver(pkg, gpu, small)
proofs = [lp.prove(x)[0] for x in lc.shared_tree_inputs(L, 6, depth=2)] + [lp.prove(lc.dummy_inputs(L))[0]] * 2
ver = pkg.Verifier(small.pack, circuit=lp.circ)
w = R.WrapperCircuit(small.pack, ver, 8, num_routed_wires=60, logic="private_batch", verify=True, zero_knowledge=True)
wc = pkg.Circuit(gpu, w.pack, max_batch=8)
c = w.commit(proofs, preimages=np.arange(32, dtype=np.uint64).reshape(8, 4), device_blinding=True)
dw = gpu.alloc(8 * (135 << w.info["degree_bits"]) * 8)
out["private_batch_levels"] = int(wc.witness_info()[1])
seeds = bytes(range(32)) * 8
out["private_batch_batch1_ms"] = timed(lambda: R.generate_wrapper_witnesses(wc, w, [c], dw, seeds[:32]), 10)
out["private_batch_batch8_ms"] = timed(lambda: R.generate_wrapper_witnesses(wc, w, [c] * 8, dw, seeds), 10)
out["private_batch_checksum"] = int(np.bitwise_xor.reduce(dw.download(135 << w.info["degree_bits"])))
print(json.dumps(out))
