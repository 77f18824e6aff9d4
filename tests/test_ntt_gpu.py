"""GPU parity: the HIP NTT / iNTT / coset-LDE (through the C ABI) against the CPU oracle, bit for bit,
plus golden vectors and size-independent properties at the BASELINE size (2^20)."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

P = 0xFFFFFFFF00000001
MULT_GEN = 14293326489335486720
G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "field_ntt.json")))


def bitrev_perm(log_n):
    n = 1 << log_n
    idx = np.arange(n, dtype=np.uint64)
    rev = np.zeros(n, dtype=np.uint64)
    for b in range(log_n):
        rev |= ((idx >> np.uint64(b)) & np.uint64(1)) << np.uint64(log_n - 1 - b)
    return rev.astype(np.int64)


def splitmix_columns(log_n, batch, seed=0x9E3779B97F4A7C15):
    """SURVEY §8(d) input: splitmix64(seed ^ column), rejected into [0, p). Vectorised."""
    n = 1 << log_n
    out = np.empty((batch, n), dtype=np.uint64)
    with np.errstate(over="ignore"):
        for c in range(batch):
            x = np.uint64(seed ^ c) + np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
            z = x
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            z = z ^ (z >> np.uint64(31))
            z = np.where(z >= np.uint64(P), z - np.uint64(P), z)
            out[c] = z
    return out


@pytest.mark.parametrize("case", G["cases"], ids=lambda c: f"log{c['log_n']}")
def test_golden_vectors(gpu, case):
    log_n = case["log_n"]
    c = np.array([case["coeffs"]], dtype=np.uint64)
    assert gpu.ntt_host(c, log_n)[0].tolist() == case["fft"]
    assert gpu.ntt_host(c, log_n, inverse=True)[0].tolist() == case["ifft_of_coeffs"]
    assert gpu.ntt_host(c, log_n, coset_shift=MULT_GEN)[0].tolist() == case["coset_fft_g"]


@pytest.mark.parametrize("case", G["edge"], ids=lambda c: f"{c['name']}{c['log_n']}")
def test_golden_edges(gpu, case):
    c = np.array([case["coeffs"]], dtype=np.uint64)
    assert gpu.ntt_host(c, case["log_n"])[0].tolist() == case["fft"]


@pytest.mark.parametrize("log_n", list(range(0, 21)))
def test_forward_inverse_vs_oracle(gpu, orc, log_n):
    batch = 3 if log_n >= 16 else 5
    rng = np.random.default_rng(100 + log_n)
    a = rng.integers(0, P, (batch, 1 << log_n), dtype=np.uint64)
    # edge columns: all p-1 and non-canonical inputs (reduced on load)
    a[0, :] = P - 1
    got = gpu.ntt_host(a, log_n)
    want = orc.fft_batch(a, log_n)
    assert np.array_equal(got, want)
    inv = gpu.ntt_host(a, log_n, inverse=True)
    assert np.array_equal(inv, orc.fft_batch(a, log_n, inverse=True))
    assert np.array_equal(gpu.ntt_host(got, log_n, inverse=True), a)
    # bit-reversed output order = reverse_index_bits of the natural result
    br = gpu.ntt_host(a, log_n, bitrev=True)
    assert np.array_equal(br, want[:, bitrev_perm(log_n)])


def test_noncanonical_inputs_are_reduced(gpu, orc):
    log_n = 6
    a = np.full((2, 64), 2**64 - 1, dtype=np.uint64)
    a[1] = np.arange(64, dtype=np.uint64) + np.uint64(P)
    red = np.where(a >= np.uint64(P), a - np.uint64(P), a)
    assert np.array_equal(gpu.ntt_host(a, log_n), orc.fft_batch(red, log_n))


@pytest.mark.parametrize("log_n,batch", [(1, 1), (4, 135), (10, 135), (12, 135), (13, 20), (16, 9)])
def test_lde_vs_oracle(gpu, orc, log_n, batch):
    """values -> ifft -> coset LDE x8, as PolynomialBatch::from_values does before hashing."""
    rate_bits = 3
    rng = np.random.default_rng(200 + log_n)
    vals = rng.integers(0, P, (batch, 1 << log_n), dtype=np.uint64)
    coeffs_want, lde_want = orc.lde_batch(vals, log_n, rate_bits, MULT_GEN)
    d_vals = gpu.to_device(vals)
    d_coeffs = gpu.alloc(vals.nbytes)
    d_lde = gpu.alloc(vals.nbytes << rate_bits)
    gpu.ntt_dev(d_vals, d_coeffs, log_n, batch, inverse=True)
    gpu.lde_dev(d_coeffs, d_lde, log_n, rate_bits, batch, coset_shift=MULT_GEN)
    gpu.sync()
    assert np.array_equal(d_coeffs.download().reshape(batch, -1), coeffs_want)
    got = d_lde.download().reshape(batch, -1)
    assert np.array_equal(got, lde_want)
    gpu.lde_dev(d_coeffs, d_lde, log_n, rate_bits, batch, coset_shift=MULT_GEN, bitrev=True)
    gpu.sync()
    got_br = d_lde.download().reshape(batch, -1)
    assert np.array_equal(got_br, lde_want[:, bitrev_perm(log_n + rate_bits)])
    for b in (d_vals, d_coeffs, d_lde):
        b.free()


def test_baseline_size_properties(gpu, orc):
    """BASELINE config 2: 2^20 points. Column 0 is checked against the oracle bit for bit; the whole
    batch through the round trip and linearity (size-independent properties)."""
    log_n, batch = 20, 16
    a = splitmix_columns(log_n, batch)
    d_a = gpu.to_device(a)
    d_f = gpu.alloc(a.nbytes)
    gpu.ntt_dev(d_a, d_f, log_n, batch)
    gpu.sync()
    f = d_f.download().reshape(batch, -1)
    assert np.array_equal(f[0], orc.fft(a[0], log_n))
    assert np.array_equal(f[batch - 1], orc.fft(a[batch - 1], log_n))
    assert int(f.max()) < P
    # linearity: NTT(a0 + a1) == NTT(a0) + NTT(a1)
    s = ((a[0].astype(object) + a[1].astype(object)) % P).astype(np.uint64)
    fs = gpu.ntt_host(s[None, :], log_n)[0]
    want = ((f[0].astype(object) + f[1].astype(object)) % P).astype(np.uint64)
    assert np.array_equal(fs, want)
    # round trip in place
    gpu.ntt_dev(d_f, d_f, log_n, batch, inverse=True)
    gpu.sync()
    assert np.array_equal(d_f.download().reshape(batch, -1), a)
    d_a.free(); d_f.free()


@pytest.mark.parametrize("log_n", [21, 22, 23])
def test_three_pass_sizes_vs_oracle(gpu, orc, log_n):
    """2^21..2^23 points (private batches of 9..64 leaves reach an LDE of 2^22): outer pass over the top three index bits,
    then eight two-pass transforms. Forward, inverse, bit-reversed order and the in-place round trip against the oracle."""
    batch = 2
    rng = np.random.default_rng(300 + log_n)
    a = rng.integers(0, P, (batch, 1 << log_n), dtype=np.uint64)
    a[1, ::3] = P - 1
    want = orc.fft_batch(a, log_n)
    d_a = gpu.to_device(a)
    d_f = gpu.alloc(a.nbytes)
    gpu.ntt_dev(d_a, d_f, log_n, batch)
    gpu.sync()
    f = d_f.download().reshape(batch, -1)
    assert np.array_equal(f, want)
    gpu.ntt_dev(d_a, d_f, log_n, batch, bitrev=True)
    gpu.sync()
    assert np.array_equal(d_f.download().reshape(batch, -1), want[:, bitrev_perm(log_n)])
    gpu.ntt_dev(d_a, d_f, log_n, batch, inverse=True)
    gpu.sync()
    assert np.array_equal(d_f.download().reshape(batch, -1), orc.fft_batch(a, log_n, inverse=True))
    d_f.upload(want)
    gpu.ntt_dev(d_f, d_f, log_n, batch, inverse=True)          # in place
    gpu.sync()
    assert np.array_equal(d_f.download().reshape(batch, -1), a)
    d_a.free(); d_f.free()


@pytest.mark.parametrize("log_n,batch", [(18, 3), (19, 2)])
def test_three_pass_lde_vs_oracle(gpu, orc, log_n, batch):
    rate_bits = 3
    rng = np.random.default_rng(400 + log_n)
    vals = rng.integers(0, P, (batch, 1 << log_n), dtype=np.uint64)
    coeffs_want, lde_want = orc.lde_batch(vals, log_n, rate_bits, MULT_GEN)
    d_coeffs = gpu.to_device(coeffs_want)
    d_lde = gpu.alloc(vals.nbytes << rate_bits)
    gpu.lde_dev(d_coeffs, d_lde, log_n, rate_bits, batch, coset_shift=MULT_GEN)
    gpu.sync()
    assert np.array_equal(d_lde.download().reshape(batch, -1), lde_want)
    gpu.lde_dev(d_coeffs, d_lde, log_n, rate_bits, batch, coset_shift=MULT_GEN, bitrev=True)
    gpu.sync()
    assert np.array_equal(d_lde.download().reshape(batch, -1), lde_want[:, bitrev_perm(log_n + rate_bits)])
    d_coeffs.free(); d_lde.free()


def test_baseline_size_edge_vectors(gpu, orc):
    """BASELINE config 2 at B = 1 and the edge inputs SURVEY 8(d) lists, at the full 2^20 points: all-zero, all p-1,
    delta, constant, and a splitmix column; forward bit for bit against the oracle, inverse back to the input."""
    log_n = 20
    n = 1 << log_n
    cols = {"zero": np.zeros(n, dtype=np.uint64), "p_minus_1": np.full(n, P - 1, dtype=np.uint64),
            "delta": np.zeros(n, dtype=np.uint64), "constant": np.full(n, 0x123456789ABCDEF, dtype=np.uint64),
            "splitmix": splitmix_columns(log_n, 1)[0]}
    cols["delta"][1] = 1
    for name, a in cols.items():
        f = gpu.ntt_host(a[None, :], log_n)[0]                    # B = 1
        assert np.array_equal(f, orc.fft(a, log_n)), name
        assert np.array_equal(gpu.ntt_host(f[None, :], log_n, inverse=True)[0], a), name
    assert not cols["zero"].any() and not gpu.ntt_host(cols["zero"][None, :], log_n)[0].any()
    # delta at index 1 transforms to the powers of the primitive 2^20-th root of unity
    f = gpu.ntt_host(cols["delta"][None, :], log_n)[0]
    w = orc.root(log_n)
    assert int(f[0]) == 1 and int(f[1]) == w and int(f[2]) == orc.mul(w, w)


def test_empty_and_bad_arguments(gpu, pkg):
    assert gpu.ntt_host(np.zeros((0, 8), dtype=np.uint64), 3).shape == (0, 8)
    with pytest.raises(pkg.QpGpuError):
        gpu.ntt_host(np.zeros((1, 1 << 24), dtype=np.uint64), 24)  # beyond the supported size: loud error
