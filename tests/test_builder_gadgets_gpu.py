"""tests/test_builder_gadgets.py's gadget circuits through stage s1 on the device: for every gadget family the device's wire matrix
(public inputs read out of the witness) equals the oracle's on random inputs, batched."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
P = 0xFFFFFFFF00000001


def gadget_circuit(pkg, kind):
    L = pkg.load_library()
    c = ctypes
    L.qpgpu_builder_gadget_circuit.restype = c.c_int
    L.qpgpu_builder_gadget_circuit.argtypes = [c.c_uint, c.c_void_p, c.c_size_t, c.POINTER(c.c_size_t), c.c_void_p, c.c_size_t, c.POINTER(c.c_size_t), c.POINTER(c.c_size_t), c.c_char_p]
    n, ni, no = c.c_size_t(), c.c_size_t(), c.c_size_t()
    err = c.create_string_buffer(400)
    assert L.qpgpu_builder_gadget_circuit(kind, None, 0, c.byref(n), None, 0, c.byref(ni), c.byref(no), err) == 0
    pack = np.empty(n.value, dtype=np.uint64); cells = np.empty(ni.value + no.value, dtype=np.uint64)
    assert L.qpgpu_builder_gadget_circuit(kind, pack.ctypes.data, pack.size, c.byref(n), cells.ctypes.data, cells.size, c.byref(ni), c.byref(no), err) == 0
    return pack, cells, ni, no


@pytest.mark.parametrize("kind", range(7))
def test_gadget_circuit_on_the_device(pkg, gpu, orc, kind):
    pack, cells, ni, no = gadget_circuit(pkg, kind)
    cin = cells[:ni.value]
    rng = np.random.default_rng(40 + kind)
    B = 4
    vals = rng.integers(0, P, (B, cin.size), dtype=np.uint64)
    if kind == 2:
        vals[:, 0] |= np.uint64(1)                         # a non-zero shift
    if kind == 3:
        vals[:, 0] %= np.uint64(1024)
    if kind == 4:
        vals[:, 0] %= np.uint64(16); vals[:, 17] %= np.uint64(2); vals[1, 19] = vals[1, 18]
    if kind == 6:
        vals[:, 0] %= np.uint64(256); vals[:, 1] %= np.uint64(2); vals[0, 2] = 0; vals[1, 2] = P - 1
    circ = pkg.Circuit(gpu, pack, max_batch=B)
    nw, rows = 135, 1 << int(pack[1])
    d = gpu.alloc(B * nw * rows * 8)
    assert circ.generate_witness_partial_batch_dev(cin, vals, None, d) == [0] * B
    got = d.download().reshape(B, nw, rows)
    pis = circ.witness_public_inputs_dev(d, B)
    for b in range(B):
        rc, want, _ = orc.generate_witness(pack, cin, vals[b], None)
        assert rc == orc.WIT_OK and np.array_equal(got[b], want), (kind, b)
        cout = cells[ni.value:]
        assert pis[b].tolist() == [int(want[int(x) % 135, int(x) // 135]) for x in cout]
    proof = circ.prove_dev(d, pis[0])
    ver = pkg.Verifier(pack, circuit=circ)
    assert ver.verify(proof)
    ver.close(); circ.close(); d.free(scrub=True)


@pytest.mark.parametrize("seed", range(12))
def test_random_gadget_programs_on_the_device(pkg, gpu, orc, seed):
    """qpgpu_builder_gadget_circuit(1000 + seed) — a random program of 40-80 gadget applications — through the device: stage s1's
    wire matrices equal the oracle's for a batch of inputs, the public inputs are read out of the witness, and the proof's bytes equal
    the oracle's proof of the same witness."""
    import oracle_binding as ob
    pack, cells, ni, no = gadget_circuit(pkg, 1000 + seed)
    cin, cout = cells[:ni.value], cells[ni.value:]
    rng = np.random.default_rng(900 + seed)
    B = 3
    vals = rng.integers(1, P, (B, cin.size), dtype=np.uint64)
    circ = pkg.Circuit(gpu, pack, max_batch=B)
    nw, rows = 135, 1 << int(pack[1])
    d = gpu.alloc(B * nw * rows * 8)
    assert circ.generate_witness_partial_batch_dev(cin, vals, None, d) == [0] * B
    got = d.download().reshape(B, nw, rows)
    pis = circ.witness_public_inputs_dev(d, B)
    oc = ob.OracleCircuit(orc, pack)
    for b in range(B):
        rc, want, _ = orc.generate_witness(pack, cin, vals[b], None)
        assert rc == orc.WIT_OK and np.array_equal(got[b], want), (seed, b)
        assert pis[b].tolist() == [int(want[int(x) % 135, int(x) // 135]) for x in cout]
    want_proof = oc.prove(got[0], pis[0])
    assert circ.prove_dev(d, pis[0]) == want_proof and oc.verify(want_proof) == 0
    oc.close(); circ.close(); d.free(scrub=True)


def test_a_generator_that_inverts_zero_fails_its_witness_alone(pkg, gpu, orc):
    """A quotient's zero denominator / an interpolation's zero coset shift (plonky2: "Tried to invert zero"): that witness of the
    batch gets QPGPU_EUNSAT and the message names the target; the others are generated; the oracle says the same."""
    for kind, zero in ((0, (2, 3)), (2, (0,))):
        pack, cells, ni, no = gadget_circuit(pkg, kind)
        cin = cells[:ni.value]
        rng = np.random.default_rng(60 + kind)
        vals = rng.integers(1, P, (3, cin.size), dtype=np.uint64)
        for z in zero:
            vals[1, z] = 0
        circ = pkg.Circuit(gpu, pack, max_batch=3)
        nw, rows = 135, 1 << int(pack[1])
        d = gpu.alloc(3 * nw * rows * 8)
        assert circ.generate_witness_partial_batch_dev(cin, vals, None, d) == [0, -4, 0]
        assert "a generator inverts target" in gpu.last_error() and "witness 1" in gpu.last_error()
        got = d.download().reshape(3, nw, rows)
        for b in (0, 2):
            rc, want, _ = orc.generate_witness(pack, cin, vals[b], None)
            assert rc == orc.WIT_OK and np.array_equal(got[b], want)
        assert orc.generate_witness(pack, cin, vals[1], None)[0] == orc.WIT_ZERO_INVERSE
        circ.close(); d.free(scrub=True)
