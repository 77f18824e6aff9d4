// poseidon.hpp — Poseidon permutation over Goldilocks (width 12, x^7, 4+22+4 rounds), host + device.
//
// Replaces plonky2::hash::poseidon::Poseidon::poseidon for GoldilocksField, the permutation behind
// PoseidonHash / PoseidonGoldilocksConfig (reference common/src/circuit.rs:17). Round constants are
// upstream's ChaCha8Rng::seed_from_u64(0) stream (poseidon_constants.cpp derives them at start-up);
// the MDS matrix is circulant [17,15,41,16,2,28,13,13,39,18,34,20] + diag(8,0,..,0).
// The hashing kernels take the permutation as a template plug (the fork may back the
// same config with Poseidon2; that plug needs qp-poseidon-core's constants).
#pragma once
#include "gl64.hpp"

namespace poseidon {
using gl::u32;
using gl::u64;

constexpr int WIDTH = 12, RATE = 8, HALF_FULL = 4, PARTIAL = 22, ROUNDS = 30;

GL_HD u64 sbox7(u64 x) {
    u64 x2 = gl::sqr(x), x4 = gl::sqr(x2), x3 = gl::mul(x, x2);
    return gl::mul(x3, x4);
}

// s <- MDS * s. Works on the 32-bit halves with 64-bit accumulators (constants < 2^6, 13 terms),
// then one 96-bit fold per output.
GL_HD void mds_layer(u64 (&s)[WIDTH]) {
    constexpr u32 C[WIDTH] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    u32 lo[WIDTH], hi[WIDTH];
#pragma unroll
    for (int i = 0; i < WIDTH; i++) { lo[i] = (u32)s[i]; hi[i] = (u32)(s[i] >> 32); }
#pragma unroll
    for (int r = 0; r < WIDTH; r++) {
        u64 al = 0, ah = 0;
#pragma unroll
        for (int i = 0; i < WIDTH; i++) {
            const int j = (i + r) % WIDTH;
            al += (u64)lo[j] * C[i];
            ah += (u64)hi[j] * C[i];
        }
        if (r == 0) { al += (u64)lo[0] * 8u; ah += (u64)hi[0] * 8u; }
        // value = al + ah * 2^32, al, ah < 2^42
        u64 low = al + (ah << 32);
        u32 top = (u32)(ah >> 32) + (low < al ? 1u : 0u);
        s[r] = gl::reduce96(low, top);
    }
}

// rc: 360 canonical round constants
GL_HD void permute(u64 (&s)[WIDTH], const u64 *rc) {
    int r = 0;
    for (int k = 0; k < HALF_FULL; k++, r++) {
#pragma unroll
        for (int i = 0; i < WIDTH; i++) s[i] = sbox7(gl::add(s[i], rc[r * WIDTH + i]));
        mds_layer(s);
    }
    for (int k = 0; k < PARTIAL; k++, r++) {
#pragma unroll
        for (int i = 0; i < WIDTH; i++) s[i] = gl::add(s[i], rc[r * WIDTH + i]);
        s[0] = sbox7(s[0]);
        mds_layer(s);
    }
    for (int k = 0; k < HALF_FULL; k++, r++) {
#pragma unroll
        for (int i = 0; i < WIDTH; i++) s[i] = sbox7(gl::add(s[i], rc[r * WIDTH + i]));
        mds_layer(s);
    }
#pragma unroll
    for (int i = 0; i < WIDTH; i++) s[i] = gl::canon(s[i]);
}

// host: derive the 360 round constants (ChaCha8, seed 0, rand 0.8 gen_range(0..p))
void derive_round_constants(u64 *out360);
const u64 *host_round_constants();

}  // namespace poseidon
