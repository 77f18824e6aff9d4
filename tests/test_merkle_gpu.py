"""GPU parity for stage s3: Poseidon permutation, leaf hashing and Merkle trees vs the CPU oracle and the
upstream permutation vectors."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

P = 0xFFFFFFFF00000001
V1 = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "poseidon_v1.json")))


def test_permutation_golden_and_oracle(gpu, orc):
    states = [np.array(v["input"], dtype=np.uint64) for v in V1["vectors"]]
    rng = np.random.default_rng(11)
    states += [rng.integers(0, P, 12, dtype=np.uint64) for _ in range(300)]
    states.append(np.full(12, P - 1, dtype=np.uint64))
    states.append(np.full(12, 2**64 - 1, dtype=np.uint64))   # non-canonical input is reduced on load
    got = gpu.poseidon_permute(np.stack(states))
    for v, g in zip(V1["vectors"], got):
        assert [int(x) for x in g] == [int(h, 16) for h in v["output"]]
    for s, g in zip(states, got):
        red = np.where(s >= np.uint64(P), s - np.uint64(P), s)
        assert np.array_equal(g, orc.poseidon(red))


@pytest.mark.parametrize("log_leaves,width,cap_h", [(4, 1, 0), (4, 4, 2), (5, 5, 4), (6, 8, 0), (6, 9, 4),
                                                     (8, 16, 4), (10, 135, 4), (12, 20, 4), (3, 32, 3), (13, 85, 4)])
def test_merkle_tree_vs_oracle(gpu, orc, log_leaves, width, cap_h):
    n = 1 << log_leaves
    rng = np.random.default_rng(1000 + log_leaves * 7 + width)
    leaves = rng.integers(0, P, (n, width), dtype=np.uint64)
    dig_want, cap_want = orc.merkle(leaves, cap_h)
    cols = np.ascontiguousarray(leaves.T)                 # column-major, leaf order
    d_cols = gpu.to_device(cols)
    total = gpu.merkle_digest_count(log_leaves, cap_h)
    assert total == dig_want.shape[0]
    d_dig = gpu.alloc(total * 32)
    cap = gpu.merkle_build_dev(d_cols, n, width, log_leaves, cap_h, d_dig)
    assert np.array_equal(cap, cap_want)
    assert np.array_equal(d_dig.download().reshape(-1, 4), dig_want)
    # row-major entry gives the same tree
    d_rows = gpu.to_device(leaves)
    cap2 = gpu.merkle_build_rows_dev(d_rows, width, log_leaves, cap_h, d_dig)
    assert np.array_equal(cap2, cap_want)
    assert np.array_equal(d_dig.download().reshape(-1, 4), dig_want)
    for b in (d_cols, d_dig, d_rows):
        b.free()


def test_lde_commit_matches_reference_order(gpu, orc):
    """PolynomialBatch::from_values: ifft -> coset LDE -> transpose -> reverse_index_bits -> MerkleTree.
    The GPU keeps the LDE column-major in bit-reversed slots; the tree must equal the oracle's tree over
    the explicitly transposed and bit-reversed rows."""
    log_n, rate_bits, ncols, cap_h = 9, 3, 20, 4
    G = 14293326489335486720
    rng = np.random.default_rng(77)
    vals = rng.integers(0, P, (ncols, 1 << log_n), dtype=np.uint64)
    _, lde = orc.lde_batch(vals, log_n, rate_bits, G)
    L = log_n + rate_bits
    rev = np.array([int(format(i, f"0{L}b")[::-1], 2) for i in range(1 << L)])
    rows = np.ascontiguousarray(lde.T[rev])               # leaf j = all columns at point rev(j)
    dig_want, cap_want = orc.merkle(rows, cap_h)
    d_vals = gpu.to_device(vals); d_coeffs = gpu.alloc(vals.nbytes); d_lde = gpu.alloc(vals.nbytes << rate_bits)
    gpu.ntt_dev(d_vals, d_coeffs, log_n, ncols, inverse=True)
    gpu.lde_dev(d_coeffs, d_lde, log_n, rate_bits, ncols, coset_shift=G, bitrev=True)
    d_dig = gpu.alloc(gpu.merkle_digest_count(L, cap_h) * 32)
    cap = gpu.merkle_build_dev(d_lde, 1 << L, ncols, L, cap_h, d_dig)
    assert np.array_equal(cap, cap_want)
    assert np.array_equal(d_dig.download().reshape(-1, 4), dig_want)
    for b in (d_vals, d_coeffs, d_lde, d_dig):
        b.free()


def test_merkle_bad_arguments(gpu, pkg):
    d = gpu.alloc(64 * 8)
    with pytest.raises(pkg.QpGpuError):
        gpu.merkle_build_dev(d, 4, 1, 2, 3, d)   # cap above the leaves
    d.free()


def test_throughput_build_and_rare_folds_vs_oracle(gpu, orc, pkg):
    """Launches of 2^18 threads or more run the throughput build of the hashing kernels (merkle_kernels_tp.hip: S-box products
    as rare-fold groups, gl64.hpp). 2^19 leaves put the leaf kernels and the first tree level there. A sprinkling of leaves
    is built so that S-box inputs of the first round are multiples of 2^32: their squares have a zero low half and a top
    word above it, which is exactly the borrow the lazy products fold behind their wave-uniform branch (a random product
    sees it once in 2^32). Digests, every tree level and the cap must equal the oracle's."""
    log_leaves, width, cap_h = 19, 9, 4
    n = 1 << log_leaves
    rng = np.random.default_rng(2024)
    leaves = rng.integers(0, P, (n, width), dtype=np.uint64)
    rc, _ = pkg.poseidon_constants()
    hit = rng.choice(n, 5000, replace=False)
    for j in hit:
        lanes = rng.choice(8, int(rng.integers(1, 9)), replace=False)
        for i in lanes:
            m = int(rng.integers(1 << 16, 1 << 32))
            leaves[j, i] = ((m << 32) - int(rc[i])) % P          # first-round S-box input = m * 2^32
    dig_want, cap_want = orc.merkle(leaves, cap_h)
    total = gpu.merkle_digest_count(log_leaves, cap_h)
    d_dig = gpu.alloc(total * 32)
    d_rows = gpu.to_device(leaves)
    cap = gpu.merkle_build_rows_dev(d_rows, width, log_leaves, cap_h, d_dig)
    assert np.array_equal(cap, cap_want)
    assert np.array_equal(d_dig.download().reshape(-1, 4), dig_want)
    d_rows.free()
    d_cols = gpu.to_device(np.ascontiguousarray(leaves.T))
    cap = gpu.merkle_build_dev(d_cols, n, width, log_leaves, cap_h, d_dig)
    assert np.array_equal(cap, cap_want)
    assert np.array_equal(d_dig.download().reshape(-1, 4), dig_want)
    d_cols.free(); d_dig.free()


def test_matrix_build_vs_oracle(gpu, orc):
    """Leaf sponges and tree levels of 2^18 hashes or more under the Poseidon hasher run the matrix-pipe build
    (merkle_kernels_mx.hip: the 22 partial rounds as one int8 GEMM, poseidon_mfma.hpp). A wires-shaped commitment (135 columns:
    17 permutations per leaf, the last block ragged) on 2^18 leaves, with runs of extreme elements (0, p - 1, 2^32 - 1, 2^63):
    digests, every level and the cap equal the oracle's."""
    log_leaves, width, cap_h = 18, 135, 4
    n = 1 << log_leaves
    rng = np.random.default_rng(31337)
    leaves = rng.integers(0, P, (n, width), dtype=np.uint64)
    ext = np.array([0, P - 1, 2**32 - 1, 2**63, 2**32, P - 2**32], dtype=np.uint64)
    for j in rng.choice(n, 4000, replace=False):
        k = int(rng.integers(1, width))
        leaves[j, rng.choice(width, k, replace=False)] = rng.choice(ext, k)
    dig_want, cap_want = orc.merkle(leaves, cap_h)
    d_dig = gpu.alloc(gpu.merkle_digest_count(log_leaves, cap_h) * 32)
    d_cols = gpu.to_device(np.ascontiguousarray(leaves.T))
    cap = gpu.merkle_build_dev(d_cols, n, width, log_leaves, cap_h, d_dig)
    assert np.array_equal(cap, cap_want)
    assert np.array_equal(d_dig.download().reshape(-1, 4), dig_want)
    d_cols.free(); d_dig.free()
