import os, sys, time, subprocess
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
import leaf_cases as lc
L = pkg.leaf
gpu = pkg.QpGpu(0)
leaf = L.LeafCircuit(min_degree_bits=13)
lay = pkg.binding.pack_p2_layout(leaf.pack) if hasattr(pkg, "binding") else None
from importlib import import_module
B = sys.modules[pkg.__name__ + ".binding"] if (pkg.__name__ + ".binding") in sys.modules else None
lay = (B or pkg).pack_p2_layout(leaf.pack)
print("layout", lay)
x = lc.real_inputs(L, depth=7, seed=3)
cells, values, pis = leaf.commit(x)
circ = pkg.Circuit(gpu, leaf.pack)
n = 1 << 13
d = gpu.alloc(135 * n * 8)
def timed(fn, reps=20):
    fn(); gpu.sync(); t = time.perf_counter()
    for _ in range(reps): fn()
    gpu.sync(); return (time.perf_counter() - t) / reps * 1e3
t0 = timed(lambda: circ.generate_witness_partial_dev(cells, values, pis, d))
w0 = d.download().reshape(135, n)
print("plain: %.3f ms" % t0, circ.witness_info())
rows = [int(r) for r in os.environ.get("P2ROWS", "").split()]
print("p2 rows", len(rows))
if rows:
    hc = np.array([r * 135 + lay["w_output"] + gidx for r in rows for gidx in range(12)], dtype=np.uint64)
    hv = np.array([w0[lay["w_output"] + gidx, r] for r in rows for gidx in range(12)], dtype=np.uint64)
    c2 = np.concatenate([cells, hc]); v2 = np.concatenate([values, hv])
    t1 = timed(lambda: circ.generate_witness_partial_dev(c2, v2, pis, d))
    w1 = d.download().reshape(135, n)
    print("with hints: %.3f ms" % t1, circ.witness_info(), "equal", np.array_equal(w0, w1))
    # a wrong hint must be caught
    v3 = v2.copy(); v3[len(values) + 40] ^= np.uint64(1)
    try:
        circ.generate_witness_partial_dev(c2, v3, pis, d); print("WRONG HINT ACCEPTED")
    except pkg.QpGpuError as e:
        print("wrong hint:", str(e)[:120])
