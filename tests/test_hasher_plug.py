"""The proof-system permutation is a plug (the fork's Merkle / Fiat-Shamir hasher may be Poseidon
or Poseidon2, and qp-poseidon-core's constants are not available offline). With a Poseidon2 parameter block selected on
both sides, host transcript, hashing kernels and whole proofs must agree with the CPU restatement. The parameters here are
placeholders (seeded random constants, the HorizenLabs 4x4 block): PARITY UNPINNED for the real constants."""
import numpy as np
import pytest

from oracle_binding import Challenger as OracleChallenger, OracleCircuit

P = 0xFFFFFFFF00000001


def placeholder_params():
    rng = np.random.default_rng(2)
    return (rng.integers(0, P, (8, 12), dtype=np.uint64), rng.integers(0, P, 22, dtype=np.uint64),
            rng.integers(1, P, 12, dtype=np.uint64), np.array([[5, 7, 1, 3], [4, 6, 1, 1], [1, 3, 5, 7], [1, 1, 4, 6]], dtype=np.uint64))


@pytest.fixture()
def poseidon2_selected(pkg, orc):
    prm = placeholder_params()
    pkg.set_hasher_poseidon2(*prm)
    orc.select_poseidon2(*prm)
    yield prm
    pkg.set_hasher_poseidon()
    orc.select_poseidon()


def test_host_transcript_under_poseidon2(pkg, orc, poseidon2_selected):
    """qpgpu_challenger_* (host code of the product) against the restated challenger, both on the plugged permutation."""
    rng = np.random.default_rng(12)
    a, b = pkg.Challenger(), OracleChallenger(orc)
    for step in range(30):
        k = int(rng.integers(0, 20))
        if k:
            xs = rng.integers(0, P, size=k, dtype=np.uint64)
            a.observe(xs); b.observe(xs)
        m = int(rng.integers(0, 11))
        assert a.get_n(m) == b.get_n(m), step


def test_selection_is_visible_and_reversible(pkg, orc):
    lib = pkg.load_library()
    assert lib.qpgpu_get_hasher() == 0
    base = pkg.Challenger(); base.observe([1, 2, 3]); v1 = base.get()
    prm = placeholder_params()
    pkg.set_hasher_poseidon2(*prm)
    try:
        assert lib.qpgpu_get_hasher() == 1
        c = pkg.Challenger(); c.observe([1, 2, 3])
        assert c.get() != v1
        with pytest.raises(pkg.QpGpuError):
            pkg.set_hasher_poseidon2(prm[0], prm[1][:5], prm[2], prm[3])       # wrong block size
    finally:
        pkg.set_hasher_poseidon()
    again = pkg.Challenger(); again.observe([1, 2, 3])
    assert again.get() == v1 and lib.qpgpu_get_hasher() == 0


@pytest.fixture()
def gpu2(pkg, poseidon2_selected):
    """A context created while Poseidon2 is the process default: the hasher is a property of the context."""
    g = pkg.QpGpu(0)
    yield g
    g.close()


@pytest.mark.gpu
def test_hashing_kernels_under_poseidon2(pkg, gpu2, orc, poseidon2_selected):
    gpu = gpu2
    rng = np.random.default_rng(13)
    st = rng.integers(0, P, (300, 12), dtype=np.uint64)
    got = gpu.poseidon_permute(st)
    assert all((got[i] == orc.poseidon(st[i])).all() for i in (0, 1, 150, 299))
    # Merkle tree through the polynomial-batch path: 2^12 leaves of 21 elements, cap height 3
    vals = rng.integers(0, P, (21, 512), dtype=np.uint64)
    o = pkg.PolyOracle(gpu, vals, rate_bits=3, cap_height=3)
    _, cap = orc.merkle(np.ascontiguousarray(o.read(lde=True).T), 3)
    assert (o.cap() == cap).all()
    o.close()


@pytest.mark.gpu
def test_proofs_under_poseidon2(pkg, gpu2, orc, poseidon2_selected):
    """Whole proofs with the plugged hasher: byte parity, verification, zero-knowledge salts, the (Poseidon) gate rows
    keep their own constants."""
    gpu = gpu2
    for d, kw, zk in ((8, dict(seed=81, num_wires=24, num_routed=16, num_public_inputs=3), False),
                      (9, dict(seed=82, poseidon=True, base_sum=True, ext_arith=True, recursion=True), True)):
        pack, wires, pis = pkg.synth_circuit(d, **kw)
        if zk:
            pack[14] = 1
        circ = pkg.Circuit(gpu, pack); oc = OracleCircuit(orc, pack)
        try:
            circ.set_blinding_seed(7)
            got = circ.prove(wires, pis)
            assert got == oc.prove(wires, pis, seed=7)
            assert oc.verify(got) == 0
        finally:
            circ.close(); oc.close()


@pytest.mark.gpu
def test_back_to_poseidon_after_the_plug(pkg, gpu, orc):
    """Runs after the Poseidon2 tests in this module: the default hasher is in force again on host and device."""
    pack, wires, pis = pkg.synth_circuit(6, num_wires=24, num_routed=16, num_public_inputs=1, seed=83)
    circ = pkg.Circuit(gpu, pack); oc = OracleCircuit(orc, pack)
    assert circ.prove(wires, pis) == oc.prove(wires, pis)
    circ.close(); oc.close()


@pytest.mark.gpu
def test_two_hashers_coexist_in_one_process(pkg, gpu, orc):
    """The hasher belongs to the context (no global state): a Poseidon context and a Poseidon2 context prove
    interleaved in one process, each byte-equal to the restatement under its own permutation."""
    prm = placeholder_params()
    g2 = pkg.QpGpu(0, hasher=prm)
    lib = pkg.load_library()
    assert lib.qpgpu_ctx_get_hasher(gpu.ctx) == 0 and lib.qpgpu_ctx_get_hasher(g2.ctx) == 1 and lib.qpgpu_get_hasher() == 0
    # no public inputs: the public-input hash is the zero digest under every hasher, so one witness serves both contexts
    pack, wires, pis = pkg.synth_circuit(7, num_wires=40, num_routed=24, num_public_inputs=0, seed=91, base_sum=True)
    c1, c2 = pkg.Circuit(gpu, pack), pkg.Circuit(g2, pack)
    try:
        p1a = c1.prove(wires, pis); p2a = c2.prove(wires, pis); p1b = c1.prove(wires, pis); p2b = c2.prove(wires, pis)
        assert p1a == p1b and p2a == p2b and p1a != p2a
        oc = OracleCircuit(orc, pack)
        assert p1a == oc.prove(wires, pis) and oc.verify(p1a) == 0 and oc.verify(p2a) != 0
        oc.close()
        orc.select_poseidon2(*prm)
        try:
            oc = OracleCircuit(orc, pack)
            assert p2a == oc.prove(wires, pis) and oc.verify(p2a) == 0
            oc.close()
        finally:
            orc.select_poseidon()
        # the context's challenger follows the context's hasher
        a, b = pkg.Challenger(gpu), pkg.Challenger(g2)
        a.observe([1, 2, 3]); b.observe([1, 2, 3])
        assert a.get() != b.get()
        # choosing the hasher after a circuit exists is refused
        with pytest.raises(pkg.QpGpuError):
            g2._check(lib.qpgpu_ctx_set_hasher(g2.ctx, 0, None, 0))
    finally:
        c1.close(); c2.close(); g2.close()
