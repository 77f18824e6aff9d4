"""Stage-level C ABI (qpgpu_oracle_*, qpgpu_challenger_*, qpgpu_fri_prove): the flow a patched plonky2 prove() would
run — commitments, opening evaluations and the FRI proof on the GPU, gate-dependent stages elsewhere — must reproduce
the CPU restatement's proof byte for byte. The gate-dependent intermediate columns (Z / partial products, quotient
chunks) come from the oracle's stage trace, standing in for the Rust code that would compute them."""
import numpy as np
import pytest

from oracle_binding import OracleCircuit

pytestmark = pytest.mark.gpu
P = 0xFFFFFFFF00000001


def parse_pack(pack):
    names = ["degree_bits", "num_wires", "num_routed_wires", "num_constants", "num_selectors", "num_challenges",
             "quotient_degree_factor", "num_partial_products", "num_public_inputs", "rate_bits", "cap_height",
             "proof_of_work_bits", "num_query_rounds", "zero_knowledge", "num_gate_constraints", "num_gates", "num_arity_rounds"]
    h = {k: int(pack[1 + i]) for i, k in enumerate(names)}
    off = 18
    h["arity_bits"] = [int(x) for x in pack[off:off + h["num_arity_rounds"]]]; off += h["num_arity_rounds"]
    off += 8 * h["num_gates"] + h["num_routed_wires"]
    h["circuit_digest"] = pack[off:off + 4].copy(); off += 4
    n = 1 << h["degree_bits"]
    ncs = h["num_selectors"] + h["num_constants"] + h["num_routed_wires"]
    h["constants_sigmas"] = pack[off:off + ncs * n].reshape(ncs, n)
    return h


def staged_proof(pkg, gpu, orc, oc, pack, wires, pis, seed=0):
    h = parse_pack(pack)
    d, nch, zk = h["degree_bits"], h["num_challenges"], bool(h["zero_knowledge"])
    n = 1 << d
    want = oc.prove(wires, pis, seed=seed)               # fills the stage trace
    kw = dict(rate_bits=h["rate_bits"], cap_height=h["cap_height"])
    zkw = dict(blinding=zk, blinding_seed=seed)
    ch = pkg.Challenger()
    o_cs = pkg.PolyOracle(gpu, h["constants_sigmas"], **kw)
    o_w = pkg.PolyOracle(gpu, wires, blinding_stream=1, **kw, **zkw)
    ch.observe(h["circuit_digest"]); ch.observe(oc.trace("pi_hash")); ch.observe(o_w.cap())
    betas, gammas = ch.get_n(nch), ch.get_n(nch)
    assert betas == list(oc.trace("betas")) and gammas == list(oc.trace("gammas"))
    zs_vals = oc.trace("zs_pp_values").reshape(-1, n)
    o_zs = pkg.PolyOracle(gpu, zs_vals, blinding_stream=2, **kw, **zkw)
    ch.observe(o_zs.cap())
    assert ch.get_n(nch) == list(oc.trace("alphas"))
    o_q = pkg.PolyOracle(gpu, oc.trace("quotient_chunk_coeffs").reshape(-1, n), coeffs=True, blinding_stream=3, **kw, **zkw)
    ch.observe(o_q.cap())
    zeta = ch.get_n(2)
    assert zeta == list(oc.trace("zeta"))
    g = orc.root(d)
    g_zeta = [orc.mul(zeta[0], g), orc.mul(zeta[1], g)]
    oracles = [o_cs, o_w, o_zs, o_q]
    opens = [o.eval(zeta) for o in oracles]
    zs_next = o_zs.eval(g_zeta, 0, nch)
    ch.observe(np.concatenate(opens)); ch.observe(zs_next)
    fri = pkg.fri_prove(gpu, oracles, [(zeta, [(0, 0, o_cs.num_polys), (1, 0, o_w.num_polys), (2, 0, o_zs.num_polys), (3, 0, o_q.num_polys)]),
                                       (g_zeta, [(2, 0, nch)])],
                        ch, h["arity_bits"], proof_of_work_bits=h["proof_of_work_bits"], num_query_rounds=h["num_query_rounds"], **kw)
    # ProofWithPublicInputs::to_bytes: caps, openings (zs, zs_next, partial products split out of the Z/PP oracle), FRI, PIs
    zs_open = opens[2]
    parts = [o_w.cap(), o_zs.cap(), o_q.cap(), opens[0], opens[1], zs_open[:nch], zs_next, zs_open[nch:], opens[3]]
    got = b"".join(np.ascontiguousarray(x, dtype=np.uint64).tobytes() for x in parts) + fri + \
        (np.asarray(pis, dtype=np.uint64) % np.uint64(P)).tobytes()
    for o in oracles:
        o.close()
    return got, want


def test_staged_flow_reproduces_the_proof(pkg, gpu, orc):
    for d, kwargs in ((8, dict(seed=61)), (10, dict(seed=62, poseidon=True, base_sum=True)), (6, dict(seed=63, num_wires=24, num_routed=16, num_public_inputs=3))):
        pack, wires, pis = pkg.synth_circuit(d, **kwargs)
        oc = OracleCircuit(orc, pack)
        got, want = staged_proof(pkg, gpu, orc, oc, pack, wires, pis)
        assert len(got) == len(want)
        assert got == want, f"staged proof differs at byte {next(i for i in range(len(got)) if got[i] != want[i])}"
        assert oc.verify(got) == 0
        oc.close()


def test_staged_flow_zero_knowledge(pkg, gpu, orc):
    pack, wires, pis = pkg.synth_circuit(7, seed=64, poseidon=True)
    pack = pack.copy(); pack[14] = 1
    oc = OracleCircuit(orc, pack)
    got, want = staged_proof(pkg, gpu, orc, oc, pack, wires, pis, seed=4242)
    assert got == want
    oc.close()


def test_oracle_read_and_eval(pkg, gpu, orc):
    rng = np.random.default_rng(5)
    vals = rng.integers(0, P, size=(5, 256), dtype=np.uint64)
    o = pkg.PolyOracle(gpu, vals, rate_bits=2, cap_height=3)
    coeffs = o.read()
    ref_coeffs, ref_lde = orc.lde_batch(vals, 8, 2, pkg.MULT_GEN)      # natural-order values on the coset g<w>
    assert (coeffs == ref_coeffs).all()
    lde = o.read(lde=True)
    assert lde.shape == (5, 1024)
    brev = np.array([int(format(i, "010b")[::-1], 2) for i in range(1024)])
    assert (lde == ref_lde[:, brev]).all()                               # slot s holds the value at g*w^bitrev(s)
    leaves = np.ascontiguousarray(lde.T)
    _, cap = orc.merkle(leaves, 3)
    assert (o.cap() == cap).all()
    # Horner in the extension field F[x]/(x^2 - 7) against the device evaluation
    z = [int(v) for v in rng.integers(0, P, size=2, dtype=np.uint64)]
    got = o.eval(z, 1, 3)

    def ext_mul(x, y):
        a = orc.add(orc.mul(x[0], y[0]), orc.mul(7, orc.mul(x[1], y[1])))
        return [a, orc.add(orc.mul(x[0], y[1]), orc.mul(x[1], y[0]))]
    for j in range(3):
        acc = [0, 0]
        for c in coeffs[1 + j][::-1]:
            acc = ext_mul(acc, z)
            acc[0] = orc.add(acc[0], int(c))
        assert list(map(int, got[j])) == acc
    # a second oracle from the coefficients commits to the same tree
    o2 = pkg.PolyOracle(gpu, coeffs, rate_bits=2, cap_height=3, coeffs=True)
    assert (o.cap() == o2.cap()).all()
    o.close(); o2.close()


def test_stage_api_argument_errors(pkg, gpu):
    vals = np.zeros((2, 64), dtype=np.uint64)
    with pytest.raises(pkg.QpGpuError):
        pkg.PolyOracle(gpu, vals, rate_bits=3, cap_height=12)      # cap above the tree
    o = pkg.PolyOracle(gpu, vals, rate_bits=3, cap_height=2)
    with pytest.raises(pkg.QpGpuError):
        o.eval([1, 2], 1, 5)
    ch = pkg.Challenger()
    with pytest.raises(pkg.QpGpuError):
        pkg.fri_prove(gpu, [o], [([3, 4], [(0, 0, 3)])], ch, [4], cap_height=2)    # range outside the oracle
    with pytest.raises(pkg.QpGpuError):
        pkg.fri_prove(gpu, [o], [([3, 4], [(0, 0, 2)])], ch, [4, 4], cap_height=2)  # reduction deeper than the degree
    o.close()
