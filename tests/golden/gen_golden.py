#!/usr/bin/env python3
"""Generates tests/golden/*.json.

No part of the reference can be executed in this image (it is Rust, un-vendored qp-plonky2 1.5.5), so
the fixtures come from three independent sources, none of which is oracle/ or the HIP code:
  1. transcriptions of the reference's own known-answer vectors (poseidon2_kats.json; sources cited
     inside the file),
  2. upstream plonky2 constants / test vectors recalled for Poseidon-v1 (poseidon_v1.json): the first
     twelve round constants and the permutation outputs for inputs 0^12 and 0..11. They pin the
     ChaCha8-seed-0 derivation used by both the oracle and the product,
  3. definition-level big-integer arithmetic in this script (field_ntt.json): O(n^2) DFTs and the
     plonky2 ifft / coset / LDE conventions of SURVEY.md Appendix A.2.
Run from the repo root: python3 tests/golden/gen_golden.py
"""
import json
import os

P = 0xFFFFFFFF00000001
ROOT_2_32 = 7277203076849721926
G = 14293326489335486720
HERE = os.path.dirname(os.path.abspath(__file__))


def root(log_n):
    return pow(ROOT_2_32, 1 << (32 - log_n), P)


def splitmix(seed):
    x = seed
    while True:
        x = (x + 0x9E3779B97F4A7C15) & (2**64 - 1)
        z = x
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & (2**64 - 1)
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & (2**64 - 1)
        z ^= z >> 31
        if z < P:
            yield z


def dft(c, log_n):
    n = 1 << log_n
    w = root(log_n)
    return [sum(c[j] * pow(w, i * j, P) for j in range(n)) % P for i in range(n)]


def idft(v, log_n):
    n = 1 << log_n
    f = dft(v, log_n)
    ninv = pow(n, P - 2, P)
    return [f[(n - i) % n] * ninv % P for i in range(n)]


def main():
    gen = splitmix(0x9E3779B97F4A7C15)
    cases = []
    for log_n in (1, 2, 3, 4, 5, 6, 8):
        n = 1 << log_n
        c = [next(gen) for _ in range(n)]
        case = {"log_n": log_n, "coeffs": c, "fft": dft(c, log_n), "ifft_of_coeffs": idft(c, log_n)}
        shifted = [c[i] * pow(G, i, P) % P for i in range(n)]
        case["coset_fft_g"] = dft(shifted, log_n)
        if log_n <= 5:
            rate = 3
            padded = shifted + [0] * (n * 7)
            case["lde3_g"] = dft(padded, log_n + rate)
        cases.append(case)
    edge = []
    for log_n in (3, 5):
        n = 1 << log_n
        for name, c in (("zero", [0] * n), ("pm1", [P - 1] * n), ("delta", [1] + [0] * (n - 1)), ("const", [5] * n)):
            edge.append({"log_n": log_n, "name": name, "coeffs": c, "fft": dft(c, log_n)})
    field = {
        "p": P, "root_2_32": ROOT_2_32, "mult_gen": G,
        "roots": {str(k): root(k) for k in range(0, 33)},
        "pow2_roots_log": {"1": 96, "2": 48, "3": 24, "4": 12, "5": 6, "6": 3},
        "mul": [[a, b, a * b % P] for a, b in [(P - 1, P - 1), (2**32, 2**32), (0xFFFFFFFF, 0xFFFFFFFF00000000), (next(gen), next(gen)), (next(gen), next(gen))]],
        "ext_w": 7,
    }
    json.dump({"field": field, "cases": cases, "edge": edge}, open(os.path.join(HERE, "field_ntt.json"), "w"))
    print("wrote field_ntt.json")


if __name__ == "__main__":
    main()
