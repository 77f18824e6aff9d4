"""Pins the CPU oracle (field + transforms) against definition-level golden vectors
(tests/golden/field_ntt.json, produced by tests/golden/gen_golden.py with Python big integers)."""
import json
import os

import numpy as np
import pytest

P = 0xFFFFFFFF00000001
G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "field_ntt.json")))


def test_field_constants(orc):
    f = G["field"]
    assert f["p"] == P
    for k, v in f["roots"].items():
        assert orc.root(int(k)) == v
    # plonky2's 2^k-th roots of unity for k <= 6 are powers of two (what the twiddle-free rounds rely on)
    for k, s in f["pow2_roots_log"].items():
        assert orc.root(int(k)) == pow(2, s, P)
    assert pow(2, 96, P) == P - 1
    # extension non-residue
    assert pow(7, (P - 1) // 2, P) == P - 1


def test_field_ops(orc):
    for a, b, c in G["field"]["mul"]:
        assert orc.mul(a, b) == c
    rng = np.random.default_rng(1)
    for _ in range(200):
        a, b = int(rng.integers(0, P, dtype=np.uint64)), int(rng.integers(0, P, dtype=np.uint64))
        assert orc.mul(a, b) == a * b % P
        assert orc.add(a, b) == (a + b) % P
        assert orc.sub(a, b) == (a - b) % P
        if a:
            assert orc.mul(a, orc.inv(a)) == 1
    for a in (0, 1, P - 1, 2**32 - 1, 2**32, P - 2**32):
        for b in (0, 1, P - 1, 2**32 - 1, 2**32, P - 2**32):
            assert orc.mul(a, b) == a * b % P
            assert orc.add(a, b) == (a + b) % P
            assert orc.sub(a, b) == (a - b) % P


@pytest.mark.parametrize("case", G["cases"], ids=lambda c: f"log{c['log_n']}")
def test_fft_conventions(orc, case):
    log_n = case["log_n"]
    c = np.array(case["coeffs"], dtype=np.uint64)
    assert orc.fft(c, log_n).tolist() == case["fft"]
    assert orc.dft_naive(c, log_n).tolist() == case["fft"]
    assert orc.ifft(c, log_n).tolist() == case["ifft_of_coeffs"]
    assert orc.coset_fft(c, log_n, G["field"]["mult_gen"]).tolist() == case["coset_fft_g"]
    assert orc.coset_ifft(np.array(case["coset_fft_g"], dtype=np.uint64), log_n, G["field"]["mult_gen"]).tolist() == case["coeffs"]
    if "lde3_g" in case:
        vals = np.array(case["fft"], dtype=np.uint64)  # values on the subgroup
        coeffs, lde = orc.lde_batch(vals, log_n, 3, G["field"]["mult_gen"])
        assert coeffs[0].tolist() == case["coeffs"]
        assert lde[0].tolist() == case["lde3_g"]


@pytest.mark.parametrize("case", G["edge"], ids=lambda c: f"{c['name']}{c['log_n']}")
def test_fft_edges(orc, case):
    assert orc.fft(np.array(case["coeffs"], dtype=np.uint64), case["log_n"]).tolist() == case["fft"]


def test_fft_roundtrip_and_linearity(orc):
    rng = np.random.default_rng(2)
    for log_n in (0, 1, 7, 12, 16):
        n = 1 << log_n
        a = rng.integers(0, P, n, dtype=np.uint64)
        b = rng.integers(0, P, n, dtype=np.uint64)
        fa, fb = orc.fft(a, log_n), orc.fft(b, log_n)
        assert np.array_equal(orc.ifft(fa, log_n), a)
        s = np.array([(int(x) + int(y)) % P for x, y in zip(a[:64], b[:64])], dtype=np.uint64)
        if n <= 64:
            assert np.array_equal(orc.fft(s, log_n), np.array([(int(x) + int(y)) % P for x, y in zip(fa, fb)], dtype=np.uint64))
    # fft agrees with the O(n^2) definition at 2^9
    a = rng.integers(0, P, 512, dtype=np.uint64)
    assert np.array_equal(orc.fft(a, 9), orc.dft_naive(a, 9))
