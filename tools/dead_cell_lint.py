"""No witness cell that a generator writes may be dead: for every cell of a generated witness that is NOT routed (advice wires: the
hash gates' round wires, BaseSum limbs, RandomAccess bits, Exponentiation / Reducing / CosetInterpolation intermediates) the cell is
changed (+ 1), and the library's witness check (the filtered gate constraints on every trace row + the permutation product, run before
the quotient stage: qpgpu_circuit_set_witness_check) must refuse the witness. A cell that can change unnoticed means a gate definition
here constrains less than the generator assumes — weaker than plonky2's, whose gates constrain every wire their generators write. Routed
cells are sampled as well (they are additionally held by the permutation argument). Circuits: a synthetic one with all 15 gate types,
the leaf circuit, the builder's gadget circuits, a random gadget program, a private-batch wrapper (sampled).
usage: python tools/dead_cell_lint.py [max cells per circuit]"""
import ctypes, json, sys, time
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import __graft_entry__ as ge
pkg = ge.load_package()
gpu = pkg.QpGpu(0)
P = pkg.P
cap = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
rng = np.random.default_rng(5)
lib = pkg.load_library(); c = ctypes
lib.qpgpu_builder_gadget_circuit.restype = c.c_int
lib.qpgpu_builder_gadget_circuit.argtypes = [c.c_uint, c.c_void_p, c.c_size_t, c.POINTER(c.c_size_t), c.c_void_p, c.c_size_t, c.POINTER(c.c_size_t), c.POINTER(c.c_size_t), c.c_char_p]


def gadget(kind):
    n, ni, no = c.c_size_t(), c.c_size_t(), c.c_size_t(); err = c.create_string_buffer(400)
    assert lib.qpgpu_builder_gadget_circuit(kind, None, 0, c.byref(n), None, 0, c.byref(ni), c.byref(no), err) == 0, err.value
    pack = np.empty(n.value, dtype=np.uint64); cells = np.empty(ni.value + no.value, dtype=np.uint64)
    assert lib.qpgpu_builder_gadget_circuit(kind, pack.ctypes.data, pack.size, c.byref(n), cells.ctypes.data, cells.size, c.byref(ni), c.byref(no), err) == 0
    return pack, cells[:ni.value]


def scan(name, pack, make_witness):
    """make_witness(circ, d_wires) -> public inputs; fills the device matrix"""
    h = pkg.pack_header(pack)
    nw, n, R = h["num_wires"], 1 << h["degree_bits"], h["num_routed_wires"]
    circ = pkg.Circuit(gpu, pack)
    d = gpu.alloc(nw * n * 8)
    pis = make_witness(circ, d)
    circ.set_witness_check(True)
    circ.prove_dev(d, pis)                                                   # the honest witness passes
    wires = d.download().reshape(nw, n)
    mask = circ.witness_free_mask(nw, n)
    adv = np.argwhere(mask[R:] == 0); adv[:, 0] += R
    routed = np.argwhere(mask[:R] == 0)
    if len(adv) > cap:
        adv = adv[rng.choice(len(adv), cap, replace=False)]
    routed = routed[rng.choice(len(routed), min(len(routed), cap // 4), replace=False)] if len(routed) else routed
    undetected = []
    t0 = time.time()
    for kind, cells in (("advice", adv), ("routed", routed)):
        for col, row in cells:
            off = (int(col) * n + int(row)) * 8
            v = np.array([(int(wires[col, row]) + 1) % P], dtype=np.uint64)
            gpu._check(gpu.lib.qpgpu_memcpy_h2d(gpu.ctx, d.ptr + off, v.ctypes.data, 8))
            try:
                circ.prove_dev(d, pis)
                undetected.append((kind, int(row), int(col)))
            except pkg.QpGpuError as e:
                assert e.code == -4, e
            v[0] = wires[col, row]
            gpu._check(gpu.lib.qpgpu_memcpy_h2d(gpu.ctx, d.ptr + off, v.ctypes.data, 8))
    circ.prove_dev(d, pis)                                                   # restored
    out = {"circuit": name, "rows": n, "advice_cells_checked": int(len(adv)), "routed_cells_checked": int(len(routed)), "undetected": len(undetected),
           "seconds": round(time.time() - t0, 1)}
    if undetected:
        out["first_undetected"] = undetected[:12]
    print(json.dumps(out), flush=True)
    circ.close(); d.free(scrub=True)
    return len(undetected)


def from_free_cells(wires, pis):
    def make(circ, d):
        mask = circ.witness_free_mask(*wires.shape)
        full = circ.generate_witness(np.where(mask == 1, wires, 0).astype(np.uint64), pis)
        gpu._check(gpu.lib.qpgpu_memcpy_h2d(gpu.ctx, d.ptr, full.ctypes.data, full.nbytes))
        return pis
    return make


def from_partial(cells, values):
    def make(circ, d):
        assert circ.generate_witness_partial_batch_dev(cells, values[None], None, d) == [0]
        return circ.witness_public_inputs_dev(d, 1)[0]
    return make


bad = 0
pack, wires, pis = pkg.synth_circuit(8, seed=4, poseidon=True, base_sum=True, ext_arith=True, recursion=True, poseidon2=True)
bad += scan("synthetic, all 15 gate types", pack, from_free_cells(wires, pis))
import leaf_cases as lc
L = pkg.leaf
leaf = L.LeafCircuit()
cells, values, lp = leaf.commit(lc.real_inputs(L, depth=5))
def leaf_make(circ, d):
    circ.generate_witness_partial_dev(cells, values, lp, d)
    return lp
bad += scan("leaf circuit", leaf.pack, leaf_make)
for kind in (0, 1, 2, 3, 4, 5, 6, 1003, 1007):
    gp, cin = gadget(kind)
    vals = rng.integers(1, P, cin.size, dtype=np.uint64)
    if kind == 3: vals[0] %= np.uint64(1024)
    if kind == 4: vals[0] %= np.uint64(16); vals[17] %= np.uint64(2)
    if kind == 6: vals[0] %= np.uint64(256); vals[1] %= np.uint64(2)
    bad += scan("gadget circuit %d" % kind, gp, from_partial(cin, vals))
# a private-batch wrapper over two leaf proofs (complete in-circuit verification), sampled
R = pkg.recursion
lpv = L.LeafProver(pkg, gpu, leaf)
sp = lc.shared_tree_inputs(L, 2, seed=31)
proofs = [lpv.prove(x)[0] for x in sp]
ver = pkg.Verifier(leaf.pack, circuit=lpv.circ)
w = R.WrapperCircuit(leaf.pack, ver, 2, num_routed_wires=60, logic="private_batch", verify=True)
com = w.commit(proofs, preimages=np.arange(8, dtype=np.uint64).reshape(2, 4), derive_public_inputs=True)
def wrap_make(circ, d):
    st = R.generate_wrapper_witnesses(circ, w, [com], d)
    assert st == [0]
    return circ.witness_public_inputs_dev(d, 1)[0]
bad += scan("private batch over 2 leaf proofs (in-circuit verifier)", w.pack, wrap_make)
print("undetected total:", bad)
sys.exit(1 if bad else 0)
