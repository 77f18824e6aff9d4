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
// The S-boxes as permute() applies them: a whole layer, or lane 0 alone in a partial round. A unit that defines
// POSEIDON_GROUPED_SBOX (the throughput build of the hashing kernels, merkle_kernels_tp.hip) gets the products of a stage as one
// rare-fold group (gl::mul_group: 19 instead of 22 vector instructions per product, one scalar branch per stage).
#if defined(__HIP_DEVICE_COMPILE__) && defined(POSEIDON_GROUPED_SBOX)
template <int N>
__device__ __forceinline__ void sbox7_layer(u64 (&x)[N]) {
    u64 x2[N], x3[N], x4[N];
    gl::mul_group(x2, x, x);
    gl::mul_group(x4, x2, x2);
    gl::mul_group(x3, x, x2);
    gl::mul_group(x, x3, x4);
}
__device__ __forceinline__ u64 sbox7_lane(u64 x) {
    u64 a[1] = {x}, x2[1];
    gl::mul_group(x2, a, a);
    u64 l[2] = {x2[0], x}, r[2] = {x2[0], x2[0]}, q[2];      // x^4 and x^3 in one group
    gl::mul_group(q, l, r);
    u64 c[1] = {q[0]}, d[1] = {q[1]}, o[1];
    gl::mul_group(o, c, d);
    return o[0];
}
#else
template <int N>
GL_HD void sbox7_layer(u64 (&x)[N]) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int i = 0; i < N; i++) x[i] = sbox7(x[i]);
}
GL_HD u64 sbox7_lane(u64 x) { return sbox7(x); }
#endif

// Reference form of the MDS layer (circulant [17,15,41,16,2,28,13,13,39,18,34,20] + diag(8,0,...)):
// 32-bit halves, 64-bit accumulators, one 96-bit fold per output. Kept for the self-check in tests.
GL_HD void mds_layer_naive(u64 (&s)[WIDTH]) {
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
        u64 low = al + (ah << 32);
        u32 top = (u32)(ah >> 32) + (low < al ? 1u : 0u);
        s[r] = gl::reduce96(low, top);
    }
}

// The circulant part on one vector of 32-bit halves, without multiplications: reduce the length-12 cyclic
// convolution modulo x^3 - w for w in {1, -1, i} (4-point real FFT over the stride-3 subsequences), where the
// kernel's images are powers of two: 64*(1,2,1), 4*(-1,-8,2), 2*(2+i, -4-i, 16-i); three 3-term twisted
// convolutions with shifts only; inverse FFT. All in signed 64-bit integers (|values| < 2^42).
typedef long long i64;
// left shift of a possibly negative value, as two's-complement arithmetic (signed << of a negative is not defined before C++20)
GL_HD i64 shl(i64 x, int k) { return (i64)((u64)x << k); }
GL_HD void mds_circulant_half(const i64 (&x)[WIDTH], i64 (&o)[WIDTH]) {
    i64 U1[3], Um[3], F[3], H[3];
#pragma unroll
    for (int b = 0; b < 3; b++) {
        const i64 e = x[b] + x[b + 6], f = x[b] - x[b + 6], g = x[b + 3] + x[b + 9], h = x[b + 3] - x[b + 9];
        U1[b] = e + g; Um[b] = e - g; F[b] = f; H[b] = h;
    }
    const i64 T = U1[0] + U1[1] + U1[2];
    const i64 A[3] = {T + U1[2], T + U1[0], T + U1[1]};                       // V_1 / 64
    const i64 B[3] = {shl(Um[2], 3) - Um[0] - shl(Um[1], 1),                    // V_-1 / 4
                      -(shl(Um[0], 3) + Um[1] + shl(Um[2], 1)),
                      shl(Um[0], 1) - shl(Um[1], 3) - Um[2]};
    const i64 f0 = F[0], f1 = F[1], f2 = F[2], h0 = H[0], h1 = H[1], h2 = H[2];
    const i64 R[3] = {shl(f0, 1) - h0 + f1 - shl(h1, 4) + f2 + shl(h2, 2),       // Re(V_i / 2)
                      -shl(f0, 2) + h0 + shl(f1, 1) - h1 + f2 - shl(h2, 4),
                      shl(f0, 4) + h0 - shl(f1, 2) + h1 + shl(f2, 1) - h2};
    const i64 I[3] = {f0 + shl(h0, 1) + shl(f1, 4) + h1 - shl(f2, 2) + h2,       // Im(V_i / 2)
                      -f0 - shl(h0, 2) + f1 + shl(h1, 1) + shl(f2, 4) + h2,
                      -f0 + shl(h0, 4) - f1 - shl(h1, 2) + f2 + shl(h2, 1)};
#pragma unroll
    for (int b = 0; b < 3; b++) {
        const i64 a16 = shl(A[b], 4), p = a16 + B[b], q = a16 - B[b];
        o[b] = p + R[b]; o[b + 3] = q + I[b]; o[b + 6] = p - R[b]; o[b + 9] = q - I[b];
    }
}

// s <- MDS * s
GL_HD void mds_layer(u64 (&s)[WIDTH]) {
    i64 lo[WIDTH], hi[WIDTH], ol[WIDTH], oh[WIDTH];
#pragma unroll
    for (int i = 0; i < WIDTH; i++) { lo[i] = (i64)(u32)s[i]; hi[i] = (i64)(s[i] >> 32); }
    mds_circulant_half(lo, ol);
    mds_circulant_half(hi, oh);
    ol[0] += shl(lo[0], 3); oh[0] += shl(hi[0], 3);   // the diagonal term 8 * s[0]
#pragma unroll
    for (int r = 0; r < WIDTH; r++) {
        const u64 al = (u64)ol[r], ah = (u64)oh[r];   // both in [0, 2^42)
        const u64 low = al + (ah << 32);
        const u32 top = (u32)(ah >> 32) + (low < al ? 1u : 0u);
        s[r] = gl::reduce96(low, top);
    }
}

// ---- the 22 partial rounds in the circulant's spectral domain ----
// In a partial round only lane 0 goes through the S-box; lanes 1..11 see nothing but the MDS matrix, round after round. The
// matrix is circ(C) + 8 e0 e0^T, and mds_circulant_half evaluates circ(C) as: forward transform (the butterflies e, f, g, h,
// U1, Um), three twisted 3-term convolutions by shifts, inverse transform. Between two consecutive applications the inverse
// and the next forward transform cancel: the spectrum of circ(C) x is (64 A, 4 B, 2 R, 2 I) of the spectrum of x. So a run of
// partial rounds stays in the spectral domain (both 32-bit halves as signed 64-bit integers, exact): per round only lane 0's
// integer value is read out of the spectrum (x0 = (U1[0] + Um[0] + 2 F[0]) / 4), reduced, sent through the S-box, and the
// change of lane 0 — together with the rank-one term 8 x0' — goes back in as one addition to U1[0], Um[0] and F[0]. The
// integers grow by at most 2^8 per round (the matrix's row sum is 256): from 34 bits after the forward transform, three rounds
// fit signed 64-bit lanes, so the 22 rounds run as seven runs of three plus one ordinary round. Per round this drops the two
// transforms and eleven of the twelve recombine-and-reduce steps: about 300 instead of 437 VALU instructions.
struct Spectrum { i64 U1[3], Um[3], F[3], H[3]; };
GL_HD void spectrum_of(const i64 (&x)[WIDTH], Spectrum &s) {
#pragma unroll
    for (int b = 0; b < 3; b++) {
        const i64 e = x[b] + x[b + 6], g = x[b + 3] + x[b + 9];
        s.U1[b] = e + g; s.Um[b] = e - g; s.F[b] = x[b] - x[b + 6]; s.H[b] = x[b + 3] - x[b + 9];
    }
}
// spectrum of circ(C) x from the spectrum of x
GL_HD void spectrum_times_circulant(Spectrum &s) {
    const i64 T = s.U1[0] + s.U1[1] + s.U1[2];
    const i64 A0 = T + s.U1[2], A1 = T + s.U1[0], A2 = T + s.U1[1];
    const i64 B0 = shl(s.Um[2], 3) - s.Um[0] - shl(s.Um[1], 1);
    const i64 B1 = -(shl(s.Um[0], 3) + s.Um[1] + shl(s.Um[2], 1));
    const i64 B2 = shl(s.Um[0], 1) - shl(s.Um[1], 3) - s.Um[2];
    const i64 f0 = s.F[0], f1 = s.F[1], f2 = s.F[2], h0 = s.H[0], h1 = s.H[1], h2 = s.H[2];
    const i64 R0 = shl(f0, 1) - h0 + f1 - shl(h1, 4) + f2 + shl(h2, 2);
    const i64 R1 = -shl(f0, 2) + h0 + shl(f1, 1) - h1 + f2 - shl(h2, 4);
    const i64 R2 = shl(f0, 4) + h0 - shl(f1, 2) + h1 + shl(f2, 1) - h2;
    const i64 I0 = f0 + shl(h0, 1) + shl(f1, 4) + h1 - shl(f2, 2) + h2;
    const i64 I1 = -f0 - shl(h0, 2) + f1 + shl(h1, 1) + shl(f2, 4) + h2;
    const i64 I2 = -f0 + shl(h0, 4) - f1 - shl(h1, 2) + f2 + shl(h2, 1);
    s.U1[0] = shl(A0, 6); s.U1[1] = shl(A1, 6); s.U1[2] = shl(A2, 6);
    s.Um[0] = shl(B0, 2); s.Um[1] = shl(B1, 2); s.Um[2] = shl(B2, 2);
    s.F[0] = shl(R0, 1); s.F[1] = shl(R1, 1); s.F[2] = shl(R2, 1);
    s.H[0] = shl(I0, 1); s.H[1] = shl(I1, 1); s.H[2] = shl(I2, 1);
}
GL_HD i64 spectrum_lane0(const Spectrum &s) { return (s.U1[0] + s.Um[0] + shl(s.F[0], 1)) >> 2; }   // exact: the sum is 4 x[0]
GL_HD void vector_of(const Spectrum &s, i64 (&x)[WIDTH]) {
#pragma unroll
    for (int b = 0; b < 3; b++) {
        const i64 p = s.U1[b] + s.Um[b], q = s.U1[b] - s.Um[b], f2 = shl(s.F[b], 1), h2 = shl(s.H[b], 1);
        x[b] = (p + f2) >> 2; x[b + 6] = (p - f2) >> 2; x[b + 3] = (q + h2) >> 2; x[b + 9] = (q - h2) >> 2;
    }
}
// lo + 2^32 hi for non-negative 64-bit halves with hi < 2^58 -> loose field element
GL_HD u64 join_halves(i64 lo, i64 hi) {
    const u64 al = (u64)lo, ah = (u64)hi;
    const u64 low = al + (ah << 32);
    const u32 top = (u32)(ah >> 32) + (low < al ? 1u : 0u);
    return gl::reduce96(low, top);
}
// A coefficient of the low-half spectrum and the same coefficient of the high-half spectrum stand for lo + 2^32 hi. After three
// rounds both have grown to ~2^58; this brings them back below 2^35 WITHOUT leaving the spectral domain: what exceeds 32 bits of
// lo moves into hi (k 2^32 off lo, k onto hi), what exceeds 32 bits of hi is folded by 2^64 = 2^32 - 1 (m 2^32 off hi, m onto
// hi, m off lo: the pair changes by -m p). k and m are taken as multiples of 4: every lane of the vector the spectra stand for
// is (U1 +- Um +- 2F)/4 or (U1 -+ Um +- 2H)/4, and a change of one coefficient by a multiple of 4 changes four lanes by the same
// integer in both halves — the divisions stay exact and every lane keeps its value mod p.
GL_HD void renormalize_pair(i64 &lo, i64 &hi) {
    const i64 k = (lo >> 32) & ~(i64)3;
    lo -= shl(k, 32); hi += k;
    const i64 m = (hi >> 32) & ~(i64)3;
    hi -= shl(m, 32); hi += m; lo -= m;
}
GL_HD void renormalize(Spectrum &sl, Spectrum &sh) {
#pragma unroll
    for (int b = 0; b < 3; b++) {
        renormalize_pair(sl.U1[b], sh.U1[b]); renormalize_pair(sl.Um[b], sh.Um[b]);
        renormalize_pair(sl.F[b], sh.F[b]); renormalize_pair(sl.H[b], sh.H[b]);
    }
}
// lo + 2^32 hi for SIGNED halves (|lo|, |hi| < 2^59) -> loose field element. After a renormalisation the lanes the spectra stand
// for are signed (a lane is (U1 - Um + 2H)/4 of coefficients that were reduced one by one), and three rounds later they reach
// 2^58. A multiple of p that exceeds that in both halves is added first: 2^28 p = (2^60 + 2^28) + 2^32 (2^60 - 2^29).
GL_HD u64 join_signed_halves(i64 lo, i64 hi) { return join_halves(lo + (((i64)1 << 60) + ((i64)1 << 28)), hi + (((i64)1 << 60) - ((i64)1 << 29))); }

// All PARTIAL partial rounds in one spectral run: s <- (MDS . S-box on lane 0 . + rc)^PARTIAL s, rc0[k * WIDTH] = round k's lane-0
// constant. The coefficients are renormalised every third round (bound: below 2^35 after it, lanes below 2^35, growth at most
// 264 = 2^8.05 per round: below 2^60 before the next one).
GL_HD void partial_rounds_spectral(u64 (&s)[WIDTH], const u64 *rc0) {
    i64 lo[WIDTH], hi[WIDTH];
#pragma unroll
    for (int i = 0; i < WIDTH; i++) { lo[i] = (i64)(u32)s[i]; hi[i] = (i64)(s[i] >> 32); }
    Spectrum sl, sh;
    spectrum_of(lo, sl); spectrum_of(hi, sh);
    i64 x_lo = lo[0], x_hi = hi[0];      // lane 0 as the spectra hold it (integers), before the rank-one term of the previous round
    i64 d_lo = 0, d_hi = 0;              // 8 * lane 0 of the previous round's S-box output: the diagonal term, not yet in the spectra
    u64 x0 = s[0];
#pragma unroll 1
    for (int k0 = 0; k0 < PARTIAL; k0 += 3) {
#pragma unroll
        for (int kk = 0; kk < 3; kk++) {
            const int k = k0 + kk;
            if (k < PARTIAL) {
                if (k) x0 = join_signed_halves(x_lo + d_lo, x_hi + d_hi);
                const u64 y = sbox7_lane(gl::add_canonical(x0, rc0[k * WIDTH]));
                const i64 y_lo = (i64)(u32)y, y_hi = (i64)(y >> 32);
                // lane 0 becomes y: the spectra take (y - what they hold for lane 0); the pending diagonal term rides along
                const i64 in_lo = y_lo - x_lo, in_hi = y_hi - x_hi;
                sl.U1[0] += in_lo; sl.Um[0] += in_lo; sl.F[0] += in_lo;
                sh.U1[0] += in_hi; sh.Um[0] += in_hi; sh.F[0] += in_hi;
                spectrum_times_circulant(sl); spectrum_times_circulant(sh);
                d_lo = shl(y_lo, 3); d_hi = shl(y_hi, 3);
                if (kk == 2 && k + 1 < PARTIAL) renormalize(sl, sh);
                x_lo = spectrum_lane0(sl); x_hi = spectrum_lane0(sh);
            }
        }
    }
    vector_of(sl, lo); vector_of(sh, hi);
    lo[0] += d_lo; hi[0] += d_hi;
#pragma unroll
    for (int i = 0; i < WIDTH; i++) s[i] = join_signed_halves(lo[i], hi[i]);
}

// rc: the 360-entry table of host_hash_round_constants(): same layout as the round constants, with the partial rounds'
// constants pushed onto lane 0. In a partial round lanes 1..11 skip the S-box, so the lane-1..11 part of a round's
// constant vector commutes with it and can be carried through the MDS matrix into the next round: only the first
// partial round adds a full vector, the others add one scalar, and what is left over after the last partial round is
// folded into the constants of the following full round. Same function, 231 fewer modular additions.
GL_HD void permute(u64 (&s)[WIDTH], const u64 *rc) {
    int r = 0;
    for (int k = 0; k < HALF_FULL; k++, r++) {
#pragma unroll
        for (int i = 0; i < WIDTH; i++) s[i] = gl::add_canonical(s[i], rc[r * WIDTH + i]);
        sbox7_layer(s);
        mds_layer(s);
    }
#pragma unroll
    for (int i = 1; i < WIDTH; i++) s[i] = gl::add_canonical(s[i], rc[r * WIDTH + i]);
    partial_rounds_spectral(s, rc + r * WIDTH);
    r += PARTIAL;
    for (int k = 0; k < HALF_FULL; k++, r++) {
#pragma unroll
        for (int i = 0; i < WIDTH; i++) s[i] = gl::add_canonical(s[i], rc[r * WIDTH + i]);
        sbox7_layer(s);
        mds_layer(s);
    }
#pragma unroll
    for (int i = 0; i < WIDTH; i++) s[i] = gl::canon(s[i]);
}
// the same with every partial round as an ordinary MDS layer (the form rounds 1-2 shipped; kept as the cross-check)
GL_HD void permute_layerwise(u64 (&s)[WIDTH], const u64 *rc) {
    int r = 0;
    for (int k = 0; k < HALF_FULL; k++, r++) {
#pragma unroll
        for (int i = 0; i < WIDTH; i++) s[i] = sbox7(gl::add_canonical(s[i], rc[r * WIDTH + i]));
        mds_layer(s);
    }
#pragma unroll
    for (int i = 1; i < WIDTH; i++) s[i] = gl::add_canonical(s[i], rc[r * WIDTH + i]);
    for (int k = 0; k < PARTIAL; k++, r++) {
        s[0] = sbox7(gl::add_canonical(s[0], rc[r * WIDTH]));
        mds_layer(s);
    }
    for (int k = 0; k < HALF_FULL; k++, r++) {
#pragma unroll
        for (int i = 0; i < WIDTH; i++) s[i] = sbox7(gl::add_canonical(s[i], rc[r * WIDTH + i]));
        mds_layer(s);
    }
#pragma unroll
    for (int i = 0; i < WIDTH; i++) s[i] = gl::canon(s[i]);
}
// the textbook schedule on the plain round constants (cross-check of the table above)
GL_HD void permute_textbook(u64 (&s)[WIDTH], const u64 *rc) {
    int r = 0;
    for (int k = 0; k < ROUNDS; k++, r++) {
        const bool full = k < HALF_FULL || k >= HALF_FULL + PARTIAL;
        for (int i = 0; i < WIDTH; i++) s[i] = gl::add_canonical(s[i], rc[r * WIDTH + i]);
        for (int i = 0; i < (full ? WIDTH : 1); i++) s[i] = sbox7(s[i]);
        mds_layer_naive(s);
    }
    for (int i = 0; i < WIDTH; i++) s[i] = gl::canon(s[i]);
}

// ---- plonky2's fast partial rounds (the basis the PoseidonGate's partial-round wires live in) ----
// Flat table layout: FIRST[12] | RC[22] (last 0) | VS[22][11] | W_HATS[22][11] | INIT[11][11] (new[1+c] = sum_r INIT[c][r] s[1+r])
constexpr int FP_FIRST = 0, FP_RC = 12, FP_VS = 34, FP_WHATS = FP_VS + 22 * 11, FP_INIT = FP_WHATS + 22 * 11, FP_WORDS = FP_INIT + 121;
constexpr u64 MDS_00 = 25;   // MDS_MATRIX_CIRC[0] + MDS_MATRIX_DIAG[0]

// partial_first_constant_layer + mds_partial_layer_init
GL_HD void fast_partial_enter(u64 (&s)[WIDTH], const u64 *fp) {
#pragma unroll
    for (int i = 0; i < WIDTH; i++) s[i] = gl::add(s[i], fp[FP_FIRST + i]);
    u64 t[11];
    for (int c = 0; c < 11; c++) {
        u64 acc = 0;
        for (int r = 0; r < 11; r++) acc = gl::add(acc, gl::mul(fp[FP_INIT + c * 11 + r], s[1 + r]));
        t[c] = acc;
    }
    for (int c = 0; c < 11; c++) s[1 + c] = t[c];
}
// after the S-box on s[0]: add the round constant (0 for the last round) and apply the sparse layer of round r
GL_HD void fast_partial_linear(u64 (&s)[WIDTH], const u64 *fp, int r) {
    const u64 s0 = gl::add(s[0], fp[FP_RC + r]);
    u64 d = gl::mul(s0, MDS_00);
    for (int i = 0; i < 11; i++) d = gl::add(d, gl::mul(fp[FP_WHATS + r * 11 + i], s[1 + i]));
    for (int i = 0; i < 11; i++) s[1 + i] = gl::add(s[1 + i], gl::mul(s0, fp[FP_VS + r * 11 + i]));
    s[0] = d;
}
const u64 *host_fast_partial();   // FP_WORDS entries, derived at start-up (poseidon_constants.cpp)

// host: derive the 360 round constants (ChaCha8, seed 0, rand 0.8 gen_range(0..p))
void derive_round_constants(u64 *out360);
const u64 *host_round_constants();        // plonky2's ALL_ROUND_CONSTANTS (PoseidonGate, fast-partial derivation)
const u64 *host_hash_round_constants();   // the table permute() takes (see there)

}  // namespace poseidon

// ---- Poseidon2 (width 12, x^7, 4 + 22 + 4 rounds) as a parameter plug ----
// The qp fork ships a second hash family whose constants (qp-poseidon-core 3.1.0) are not available offline, and which
// permutation backs the proof-system hasher of the fork cannot be told from the reference: the
// Merkle / challenger permutation is therefore selectable, with the Poseidon2 parameters injected by the caller.
namespace poseidon2 {
using gl::u64;
struct Params {
    u64 rc_ext[8 * 12];   // external (full) round constants
    u64 rc_int[22];       // internal (partial) round constants, element 0 only
    u64 diag_m1[12];      // internal matrix = J + diag(diag_m1)
    u64 m4[16];           // 4x4 block of the external matrix circ(2 M4, M4, M4), row major
};
constexpr int PARAM_WORDS = 96 + 22 + 12 + 16;
const Params &qp_params();   // qp-poseidon-core 3.1.0's parameter set, pinned by the reference's known-answer vectors (poseidon_constants.cpp)

GL_HD void ext_layer(u64 (&s)[12], const Params &p) {
    u64 t[12];
#pragma unroll
    for (int b = 0; b < 3; b++)
#pragma unroll
        for (int i = 0; i < 4; i++) {
            u64 acc = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) acc = gl::add(acc, gl::mul(p.m4[4 * i + j], s[4 * b + j]));
            t[4 * b + i] = acc;
        }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const u64 sum = gl::add(gl::add(t[i], t[4 + i]), t[8 + i]);
#pragma unroll
        for (int b = 0; b < 3; b++) s[4 * b + i] = gl::add(t[4 * b + i], sum);
    }
}
// The same layer for qp-poseidon-core's block MDSMat4 = circ(2, 3, 1, 1), without multiplications: with t = a + b + c + d the
// four outputs of a block are t + a + 2b, t + b + 2c, t + c + 2d, t + d + 2a. The Poseidon2 GATE and the application sponge
// always run on that parameter set (poseidon2::qp_params, pinned by the reference's vectors; poseidon_constants.cpp asserts the
// block), so their kernels use this form; the proof-system hasher plug keeps the general one (caller-supplied blocks).
GL_HD void ext_layer_qp(u64 (&s)[12]) {
    u64 t[12];
#pragma unroll
    for (int b = 0; b < 3; b++) {
        const u64 a = s[4 * b], bb = s[4 * b + 1], c = s[4 * b + 2], d = s[4 * b + 3];
        const u64 sum = gl::add(gl::add(a, bb), gl::add(c, d));
        t[4 * b] = gl::add(gl::add(sum, a), gl::add(bb, bb));
        t[4 * b + 1] = gl::add(gl::add(sum, bb), gl::add(c, c));
        t[4 * b + 2] = gl::add(gl::add(sum, c), gl::add(d, d));
        t[4 * b + 3] = gl::add(gl::add(sum, d), gl::add(a, a));
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const u64 sum = gl::add(gl::add(t[i], t[4 + i]), t[8 + i]);
#pragma unroll
        for (int b = 0; b < 3; b++) s[4 * b + i] = gl::add(t[4 * b + i], sum);
    }
}
// internal (partial-round) linear layer: s <- (J + diag(diag_m1)) s
GL_HD void int_layer(u64 (&s)[12], const Params &p) {
    u64 sum = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) sum = gl::add(sum, s[i]);
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = gl::add(gl::mul(s[i], p.diag_m1[i]), sum);
}
// qp-poseidon-core's permutation: p must be qp_params() (only the round constants and the diagonal are read from it; they are
// canonical, so adding them needs one carry fold)
GL_HD void permute_qp(u64 (&s)[12], const Params &p) {
    ext_layer_qp(s);
    for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int i = 0; i < 12; i++) s[i] = gl::add_canonical(s[i], p.rc_ext[r * 12 + i]);
        poseidon::sbox7_layer(s);
        ext_layer_qp(s);
    }
    for (int r = 0; r < 22; r++) {
        s[0] = poseidon::sbox7_lane(gl::add_canonical(s[0], p.rc_int[r]));
        int_layer(s, p);
    }
    for (int r = 4; r < 8; r++) {
#pragma unroll
        for (int i = 0; i < 12; i++) s[i] = gl::add_canonical(s[i], p.rc_ext[r * 12 + i]);
        poseidon::sbox7_layer(s);
        ext_layer_qp(s);
    }
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = gl::canon(s[i]);
}
GL_HD void permute(u64 (&s)[12], const Params &p) {
    ext_layer(s, p);
    for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int i = 0; i < 12; i++) s[i] = gl::add(s[i], p.rc_ext[r * 12 + i]);
        poseidon::sbox7_layer(s);
        ext_layer(s, p);
    }
    for (int r = 0; r < 22; r++) {
        s[0] = poseidon::sbox7_lane(gl::add(s[0], p.rc_int[r]));
        int_layer(s, p);
    }
    for (int r = 4; r < 8; r++) {
#pragma unroll
        for (int i = 0; i < 12; i++) s[i] = gl::add(s[i], p.rc_ext[r * 12 + i]);
        poseidon::sbox7_layer(s);
        ext_layer(s, p);
    }
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = gl::canon(s[i]);
}
}  // namespace poseidon2

// ---- choice of the proof-system permutation (Merkle trees, challenger, public-input hash, PoW) ----
// A property of the context (qpgpu_ctx_set_hasher): two contexts of one process may prove under different hashers. The
// process-wide default below only seeds new contexts and the host-only helpers that have no context (synthetic circuit
// generator, qpgpu_challenger_*); qpgpu_set_hasher changes it and is kept for callers written against the first ABI.
namespace hasher {
enum Kind : int { POSEIDON = 0, POSEIDON2 = 1 };
struct Config {
    int kind = POSEIDON;
    poseidon2::Params p2{};
    void permute(gl::u64 (&s)[12]) const;     // the selected permutation on the host (challenger, small hashes)
};
const Config &process_default();
void set_process_default(int kind, const poseidon2::Params *p);
inline int kind() { return process_default().kind; }
inline void host_permute(gl::u64 (&s)[12]) { process_default().permute(s); }
}  // namespace hasher
