// poseidon_mfma.hpp — the 22 partial rounds of the Poseidon permutation (poseidon.hpp) as one constant matrix on the matrix pipe.
//
// Same function as poseidon::permute (plonky2::hash::poseidon::Poseidon::poseidon for GoldilocksField; reference
// common/src/circuit.rs:17), bit for bit; only the schedule of the partial rounds differs. In a partial round lane 0 alone goes
// through the S-box, so with N = M P (M the MDS matrix, P = "zero lane 0"), m0 = M e0 and y_k the S-box output of round k, the S-box
// input of every round and the state after the last one are affine in (s, y_0 .. y_{k-1}):
//     x_k = (e0^T N^k)(s + c') + sum_{j<k} (e0^T N^{k-1-j} m0) y_j + c_k         out = N^22 (s + c') + sum_j y_j N^{21-j} m0 + c''
// (c' the lanes-1..11 constants before the first partial round, c'' the constants of the full round that follows: both folded into
// the additive terms; and s = M t for the S-box outputs t of the fourth full round, so that round's MDS layer is in the matrix too). That is one matrix of dense 64-bit field constants with 34 columns; a 64-bit modular matrix product is an
// int8 GEMM on byte digits, and v_mfma_i32_32x32x32_i8 runs beside the vector ALU, which the rest of the permutation saturates:
//   * B operand = the inputs as 8 signed base-256 digits per element. The state of lane n is column n, an element's 8 digits are 8
//     consecutive k: the registers already hold the operand, up to one v_permlane32_swap per register pair (lanes 32..63 of a
//     32-column tile carry the second half of K).
//   * A operand: column (element e, digit b) of output o holds the 8 signed digits of the constant w[o][e] 256^b mod p (the
//     representative in the balanced range: w or w - p always fits 8 digits in [-128, 127]), one digit per output limb: 8 limbs per
//     output, 4 outputs per 32-row tile. Tiles live in LDS (60 KB), one ds_read_b128 per MFMA pair.
//   * C operand = 2^23 + the additive constant's bytes: every limb comes out in [0, 2^24) (|sum| < 2^22.2), so recombining 8 limbs
//     at 8-bit spacing is three byte concatenations (limbs l, l+3, l+6 do not overlap), one 96-bit add chain and one reduce96:
//     about 20 vector instructions per output where an MDS layer in the spectral form costs about 270.
//   * The K dimension only ever holds complete groups of four y; the y of the running group reach the rounds of that group through
//     three small integer coefficients (25, 5 017, 1 259 209) on the VALU.
// The accumulator layout gives a lane 16 rows of its own column and 16 of the partner lane's (lane +-32): 16 swaps give every lane
// the 32 rows of its own column. Every register of both accumulators is read by those swaps: nothing may be written into the
// destination of an MFMA that is still in flight (measured: a wave sharing its SIMD's matrix pipe with other waves sees its MFMAs
// finish later than the fixed distance the compiler's hazard table assumes when it reuses a dead destination register for an LDS
// load; wrong results on every wave but one per SIMD).
#pragma once
#include <string.h>
#include <utility>
#include "poseidon.hpp"

namespace pmf {
using gl::u32;
using gl::u64;
#if defined(__HIPCC__)
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
#endif

constexpr int LIMBS = 8, N_GROUP = 6, N_FIN = 3, FIN_STEPS = 9, N_ELEM = 36;   // 12 state elements + 22 y + 2 zero
constexpr int group_steps(int g) { return 3 + g; }                  // group g = rounds 4g .. 4g+3: elements 12 + 4g known
constexpr int group_base(int g) { int b = 0; for (int i = 0; i < g; i++) b += group_steps(i); return b; }
constexpr int N_GROUP_TILES = group_base(N_GROUP);                  // 33
constexpr int N_TILES = N_GROUP_TILES + N_FIN * FIN_STEPS;          // 60
constexpr int CINIT_OFF = N_TILES * 1024;
constexpr int G_OFF = CINIT_OFF + (N_GROUP + N_FIN) * 128;         // three 64-bit coefficients of the running group's y
constexpr int TABLE_BYTES = G_OFF + 32;
static_assert(N_GROUP_TILES == 33 && TABLE_BYTES % 16 == 0, "table layout");
constexpr u64 DIGIT_BIAS = 0x8080808080808080ull;

// coefficients of the running group's y: G[d] = e0^T N^d m0 for d = 0, 1, 2 (integers, no reduction)
constexpr u64 mds_entry(int r, int j) {
    constexpr u32 C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    return C[(j - r + 12) % 12] + ((r == 0 && j == 0) ? 8 : 0);
}
constexpr u64 small_g(int d) {
    u64 R[12] = {1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int it = 0; it < d; it++) {
        u64 T[12] = {};
        for (int j = 1; j < 12; j++)
            for (int i = 0; i < 12; i++) T[j] += R[i] * mds_entry(i, j);
        for (int j = 0; j < 12; j++) R[j] = T[j];
    }
    u64 g = 0;
    for (int i = 0; i < 12; i++) g += R[i] * mds_entry(i, 0);
    return g;
}
constexpr u32 G0 = (u32)small_g(0), G1 = (u32)small_g(1), G2 = (u32)small_g(2);
static_assert(small_g(0) == 25 && small_g(2) < (1ull << 32), "group coefficients fit one word");

// ---- pieces shared by the device code and the host emulation ----
// v -> eight signed digits d_b in [-128, 127] packed in a u64 with sum d_b 256^b = v (mod p): the integer v if v + BIAS does not
// wrap, else v - p (then v + BIAS + 2^32 - 1 wraps exactly once); digit = byte - 128, i.e. byte ^ 0x80 read as signed
GL_HD u64 to_digits(u64 v) {
    u64 t = v + DIGIT_BIAS;
    t += (t < v) ? 0xFFFFFFFFull : 0ull;
    return t ^ DIGIT_BIAS;
}
GL_HD u32 byte_perm(u32 s0, u32 s1, u32 sel) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_perm(s0, s1, sel);
#else
    const u64 pool = ((u64)s0 << 32) | s1;
    u32 r = 0;
    for (int i = 0; i < 4; i++) r |= (u32)((pool >> (8 * ((sel >> (8 * i)) & 7))) & 0xFF) << (8 * i);
    return r;
#endif
}
// sum_l z[l] 2^(8 l) mod p for 8 limbs in [0, 2^24): limbs l, l+3, l+6 do not overlap (three byte concatenations), then one
// 96-bit add chain and one reduce96 (the top word stays below 2^18)
GL_HD u64 recombine(const u32 (&z)[LIMBS]) {
    const u32 a0 = (z[3] << 24) | z[0], a1 = byte_perm(z[6], z[3], 0x05040201u), a2 = z[6] >> 16;
    const u32 b0 = z[1] << 8, b1 = (z[7] << 24) | z[4], b2 = z[7] >> 8;
    const u32 c0 = z[2] << 16, c1 = byte_perm(z[5], z[2], 0x06050402u);
    typedef unsigned __int128 u128;
    const u128 A = ((u128)a2 << 64) | ((u128)a1 << 32) | a0;
    const u128 B = ((u128)b2 << 64) | ((u128)b1 << 32) | b0;
    const u128 Cc = ((u128)c1 << 32) | c0;
    const u128 S = A + B + Cc;
    return gl::reduce96((u64)S, (u32)(S >> 64));
}
GL_HD u64 mul_small(u64 y, u32 g) {
    const u64 p0 = (u64)(u32)y * g;
    const u64 p1 = (u64)(u32)(y >> 32) * g + (p0 >> 32);
    return gl::reduce96((p1 << 32) | (u32)p0, (u32)(p1 >> 32));
}
// Tile rows. After the swaps a lane holds the 32 rows of its column in two register sets: X[i] = row (i&3) + 8 (i>>2), Y[i] = that
// row + 4. Outputs 0, 1 of a tile are X[0..7], X[8..15]; outputs 2, 3 are Y[0..7], Y[8..15]; limb = i & 7.
constexpr int row_of(int output, int limb) { const int i = 8 * (output & 1) + limb; return (i & 3) + 8 * (i >> 2) + 4 * (output >> 1); }

#if !defined(__HIP_DEVICE_COMPILE__)
// ---- host: table construction (once per process) and an integer emulation of the device schedule ----
namespace host {
typedef unsigned __int128 u128;
inline u64 fmul(u64 a, u64 b) { return (u64)((u128)a * b % gl::P); }
inline u64 fadd(u64 a, u64 b) { return (u64)(((u128)a + b) % gl::P); }
inline u64 fsub(u64 a, u64 b) { return (u64)(((u128)a + gl::P - b % gl::P) % gl::P); }
// eight signed digits in [-128, 127] of the representative of w mod p in the balanced range
inline bool const_digits(u64 w, signed char (&d)[LIMBS]) {
    w %= gl::P;
    __int128 rep = w <= 0x7F7F7F7F7F7F7F7Full ? (__int128)w : (__int128)w - (__int128)gl::P;
    for (int a = 0; a < LIMBS; a++) {
        int dg = (int)(rep & 255);
        if (dg >= 128) dg -= 256;
        d[a] = (signed char)dg;
        rep = (rep - dg) / 256;
    }
    return rep == 0;
}
struct Rows { u64 w[4][N_ELEM]; u64 addc[4]; };   // the four outputs of a tile
inline bool fill_tiles(const Rows &rw, int steps, unsigned char *tiles, unsigned char *cinit) {
    static signed char dig[4][N_ELEM][8][LIMBS];
    for (int o = 0; o < 4; o++)
        for (int e = 0; e < N_ELEM; e++) {
            u64 w = rw.w[o][e] % gl::P;
            for (int b = 0; b < 8; b++) { if (!const_digits(w, dig[o][e][b])) return false; w = fmul(w, 256); }
        }
    for (int q = 0; q < steps; q++)
        for (int o = 0; o < 4; o++)
            for (int limb = 0; limb < LIMBS; limb++)
                for (int k = 0; k < 32; k++) {
                    const int lane = row_of(o, limb) + 32 * (k >> 4), j = k & 15, e = 4 * q + (k >> 3), b = k & 7;
                    tiles[(size_t)q * 1024 + lane * 16 + j] = (unsigned char)dig[o][e][b][limb];
                }
    // accumulator start: 2^23 + byte l of (addc - BIAS), BIAS = sum_{l<LIMBS} 2^23 2^(8l); register i of lane half h is row
    // (i&3) + 8 (i>>2) + 4 h = row_of(2 h + (i >> 3), i & 7)
    u64 bias = 0, pw = 1u << 23;
    for (int l = 0; l < LIMBS; l++) { bias = fadd(bias, pw); pw = fmul(pw, 256); }
    for (int h = 0; h < 2; h++)
        for (int i = 0; i < 16; i++) {
            const u64 c = fsub(rw.addc[2 * h + (i >> 3)], bias);
            const u32 v = (1u << 23) + (u32)((c >> (8 * (i & 7))) & 0xFF);
            memcpy(cinit + h * 64 + i * 4, &v, 4);
        }
    return true;
}
// The partial-round block of a Poseidon-family permutation as the builder sees it: 22 rounds of `lane 0 <- (lane 0 + c_k)^7, state
// <- M state`, entered with `pre t + cvec` (t = the S-box outputs of the full round before: that round's linear layer is `pre`)
// and left with `+ after` (the constants of the full round behind).
struct Linear {
    u64 M[12][12], pre[12][12];
    u64 rc[22][12], after[12];      // rc[k]: the constants added to the state before round k's S-box (all twelve lanes)
};
inline bool build_tables(const Linear &L, unsigned char *tab) {
    static u64 R[23][12], G[22], Q[22][12], NP[12][12];
    const auto &M = L.M;
    for (int j = 0; j < 12; j++) R[0][j] = j == 0;
    for (int k = 0; k < 22; k++)
        for (int j = 0; j < 12; j++) {
            u64 a = 0;
            if (j >= 1) for (int i = 0; i < 12; i++) a = fadd(a, fmul(R[k][i], M[i][j]));
            R[k + 1][j] = a;
        }
    for (int d = 0; d < 22; d++) { u64 a = 0; for (int i = 0; i < 12; i++) a = fadd(a, fmul(R[d][i], M[i][0])); G[d] = a; }
    for (int i = 0; i < 12; i++) Q[0][i] = M[i][0] % gl::P;
    for (int d = 0; d + 1 < 22; d++)
        for (int i = 0; i < 12; i++) { u64 a = 0; for (int j = 1; j < 12; j++) a = fadd(a, fmul(M[i][j], Q[d][j])); Q[d + 1][i] = a; }
    for (int i = 0; i < 12; i++) for (int j = 0; j < 12; j++) NP[i][j] = i == j;
    for (int it = 0; it < 22; it++) {
        u64 T[12][12];
        for (int i = 0; i < 12; i++) for (int e = 0; e < 12; e++) { u64 a = 0; for (int j = 1; j < 12; j++) a = fadd(a, fmul(M[i][j], NP[j][e])); T[i][e] = a; }
        memcpy(NP, T, sizeof T);
    }
    // additive parts: acc_k = what the constants of rounds < k have put into the state before round k (acc_0 = 0,
    // acc_{k+1} = N (acc_k + rc_k)); round k's S-box input carries acc_k[0] + rc_k[0], the final state acc_22 + after
    static u64 accv[23][12];
    memset(accv, 0, sizeof accv);
    for (int k = 0; k < 22; k++)
        for (int i = 0; i < 12; i++) { u64 a = 0; for (int j = 1; j < 12; j++) a = fadd(a, fmul(M[i][j], fadd(accv[k][j], L.rc[k][j] % gl::P))); accv[k + 1][i] = a; }
    memset(tab, 0, TABLE_BYTES);
    for (int g = 0; g < N_GROUP; g++) {
        Rows rw{};
        for (int o = 0; o < 4; o++) {
            const int k = 4 * g + o;
            if (k >= 22) continue;                            // the last group has two rounds
            for (int e = 0; e < 12; e++) { u64 a = 0; for (int i = 0; i < 12; i++) a = fadd(a, fmul(R[k][i], L.pre[i][e])); rw.w[o][e] = a; }   // (e0^T N^k) pre
            for (int j = 0; j < 4 * g; j++) rw.w[o][12 + j] = G[k - 1 - j];
            rw.addc[o] = fadd(accv[k][0], L.rc[k][0] % gl::P);
        }
        if (!fill_tiles(rw, group_steps(g), tab + (size_t)group_base(g) * 1024, tab + CINIT_OFF + g * 128)) return false;
    }
    for (int f = 0; f < N_FIN; f++) {
        Rows rw{};
        for (int o = 0; o < 4; o++) {
            const int i = 4 * f + o;
            for (int e = 0; e < 12; e++) { u64 a = 0; for (int j = 0; j < 12; j++) a = fadd(a, fmul(NP[i][j], L.pre[j][e])); rw.w[o][e] = a; }   // N^22 pre
            for (int j = 0; j < 22; j++) rw.w[o][12 + j] = Q[21 - j][i];
            rw.addc[o] = fadd(accv[22][i], L.after[i] % gl::P);
        }
        if (!fill_tiles(rw, FIN_STEPS, tab + (size_t)(N_GROUP_TILES + f * FIN_STEPS) * 1024, tab + CINIT_OFF + (N_GROUP + f) * 128)) return false;
    }
    memcpy(tab + G_OFF, G, 3 * sizeof(u64));                  // the running group's coefficients (Poseidon: 25, 5 017, 1 259 209)
    return true;
}
// plonky2's Poseidon; rc: the 360-entry table poseidon::permute takes (host_hash_round_constants layout)
inline bool build_tables(const u64 *rc, unsigned char *tab) {
    static Linear L;
    memset(&L, 0, sizeof L);
    for (int r = 0; r < 12; r++) for (int j = 0; j < 12; j++) L.M[r][j] = L.pre[r][j] = mds_entry(r, j);
    for (int k = 0; k < 22; k++) L.rc[k][0] = rc[(4 + k) * 12];
    for (int i = 1; i < 12; i++) L.rc[0][i] = rc[4 * 12 + i];
    for (int i = 0; i < 12; i++) L.after[i] = rc[26 * 12 + i];
    if (!build_tables(L, tab)) return false;
    u64 g[3]; memcpy(g, tab + G_OFF, sizeof g);
    return g[0] == G0 && g[1] == G1 && g[2] == G2;            // the device code has them as immediates
}
// plonky2's PoseidonGate (quotient kernel): the same matrices with the PLAIN round constants (ALL_ROUND_CONSTANTS: twelve lanes in
// every partial round), as the gate's constraints are written against the textbook schedule; rcp: poseidon::host_round_constants()
inline bool build_tables_gate(const u64 *rcp, unsigned char *tab) {
    static Linear L;
    memset(&L, 0, sizeof L);
    for (int r = 0; r < 12; r++) for (int j = 0; j < 12; j++) L.M[r][j] = L.pre[r][j] = mds_entry(r, j);
    for (int k = 0; k < 22; k++) for (int i = 0; i < 12; i++) L.rc[k][i] = rcp[(4 + k) * 12 + i];
    for (int i = 0; i < 12; i++) L.after[i] = rcp[26 * 12 + i];
    return build_tables(L, tab);
}
// qp-poseidon-core's Poseidon2 (poseidon2::permute_qp): internal layer J + diag, the external layer before it folded in; both
// matrices are read off the layer functions themselves (columns = images of the unit vectors)
inline bool build_tables_p2(const poseidon2::Params &p, unsigned char *tab) {
    static Linear L;
    memset(&L, 0, sizeof L);
    for (int j = 0; j < 12; j++) {
        u64 e[12] = {}, f[12] = {};
        e[j] = f[j] = 1;
        poseidon2::int_layer(e, p); poseidon2::ext_layer_qp(f);
        for (int i = 0; i < 12; i++) { L.M[i][j] = gl::canon(e[i]); L.pre[i][j] = gl::canon(f[i]); }
    }
    for (int k = 0; k < 22; k++) L.rc[k][0] = p.rc_int[k];
    for (int i = 0; i < 12; i++) L.after[i] = p.rc_ext[4 * 12 + i];
    return build_tables(L, tab);
}
// what one MFMA chain computes for one column, from the table bytes (layout assumptions as the device code's)
inline bool emu_gemm(const unsigned char *tab, int tile0, int cidx, int steps, const u64 *D, u32 (&Z)[4][LIMBS]) {
    for (int o = 0; o < 4; o++)
        for (int limb = 0; limb < LIMBS; limb++) {
            const int row = row_of(o, limb);
            u32 c; memcpy(&c, tab + CINIT_OFF + cidx * 128 + (o >> 1) * 64 + (8 * (o & 1) + limb) * 4, 4);
            long long acc = c;
            for (int q = 0; q < steps; q++)
                for (int k = 0; k < 32; k++) {
                    const int lane = row + 32 * (k >> 4), j = k & 15;
                    const int a = (signed char)tab[(size_t)(tile0 + q) * 1024 + lane * 16 + j];
                    const int bdig = (signed char)(D[4 * q + (k >> 3)] >> (8 * (k & 7)));
                    acc += (long long)a * bdig;
                }
            if (acc < 0 || acc >= (1 << 24)) return false;
            Z[o][limb] = (u32)acc;
        }
    return true;
}
// the partial-round block as the device runs it: s = S-box outputs of the full round before -> the state the S-box layer of the
// full round behind applies to
inline bool emu_partial_rounds(u64 (&s)[12], const unsigned char *tab) {
    u64 D[N_ELEM] = {}, y[24], G[3];
    memcpy(G, tab + G_OFF, sizeof G);
    for (int e = 0; e < 12; e++) D[e] = to_digits(s[e]);
    for (int g = 0; g < N_GROUP; g++) {
        u32 Z[4][LIMBS];
        if (!emu_gemm(tab, group_base(g), g, group_steps(g), D, Z)) return false;
        for (int o = 0; o < 4 && 4 * g + o < 22; o++) {
            const int k = 4 * g + o;
            u64 x = recombine(Z[o]);
            for (int j = 4 * g; j < k; j++) x = gl::add(x, G[k - 1 - j] >> 32 ? gl::mul(y[j], G[k - 1 - j]) : mul_small(y[j], (u32)G[k - 1 - j]));
            y[k] = poseidon::sbox7_lane(x);
        }
        for (int o = 0; o < 4 && 4 * g + o < 22; o++) D[12 + 4 * g + o] = to_digits(y[4 * g + o]);
    }
    for (int f = 0; f < N_FIN; f++) {
        u32 Z[4][LIMBS];
        if (!emu_gemm(tab, N_GROUP_TILES + f * FIN_STEPS, N_GROUP + f, FIN_STEPS, D, Z)) return false;
        for (int o = 0; o < 4; o++) s[4 * f + o] = recombine(Z[o]);
    }
    return true;
}
// textbook permutation (plain constants) through the gate table: what the PoseidonGate kernel's matrix form computes when every
// S-box input equals its wire (a satisfied row)
inline bool emu_permute_textbook(u64 (&s)[12], const u64 *rcp, const unsigned char *tab) {
    using namespace poseidon;
    int r = 0;
    for (int k = 0; k < HALF_FULL; k++, r++) {
        for (int i = 0; i < WIDTH; i++) s[i] = sbox7(gl::add_canonical(s[i], rcp[r * WIDTH + i]));
        if (k + 1 < HALF_FULL) mds_layer(s);
    }
    if (!emu_partial_rounds(s, tab)) return false;
    r += 1 + PARTIAL;
    for (int i = 0; i < WIDTH; i++) s[i] = sbox7(s[i]);
    mds_layer(s);
    for (int k = 1; k < HALF_FULL; k++, r++) {
        for (int i = 0; i < WIDTH; i++) s[i] = sbox7(gl::add_canonical(s[i], rcp[r * WIDTH + i]));
        mds_layer(s);
    }
    for (int i = 0; i < WIDTH; i++) s[i] = gl::canon(s[i]);
    return true;
}
inline bool emu_permute(u64 (&s)[12], const u64 *rc, const unsigned char *tab) {
    using namespace poseidon;
    int r = 0;
    for (int k = 0; k < HALF_FULL; k++, r++) {
        for (int i = 0; i < WIDTH; i++) s[i] = gl::add_canonical(s[i], rc[r * WIDTH + i]);
        sbox7_layer(s);
        if (k + 1 < HALF_FULL) mds_layer(s);          // the fourth round's MDS layer is part of the matrix
    }
    if (!emu_partial_rounds(s, tab)) return false;
    r += 1 + PARTIAL;
    sbox7_layer(s); mds_layer(s);
    for (int k = 1; k < HALF_FULL; k++, r++) {
        for (int i = 0; i < WIDTH; i++) s[i] = gl::add_canonical(s[i], rc[r * WIDTH + i]);
        sbox7_layer(s);
        mds_layer(s);
    }
    for (int i = 0; i < WIDTH; i++) s[i] = gl::canon(s[i]);
    return true;
}
inline bool emu_permute_p2(u64 (&s)[12], const poseidon2::Params &p, const unsigned char *tab) {
    poseidon2::ext_layer_qp(s);
    for (int r = 0; r < 4; r++) {
        for (int i = 0; i < 12; i++) s[i] = gl::add(s[i], p.rc_ext[r * 12 + i]);
        poseidon::sbox7_layer(s);
        if (r < 3) poseidon2::ext_layer_qp(s);        // the fourth round's external layer is part of the matrix
    }
    if (!emu_partial_rounds(s, tab)) return false;
    poseidon::sbox7_layer(s); poseidon2::ext_layer_qp(s);
    for (int r = 5; r < 8; r++) {
        for (int i = 0; i < 12; i++) s[i] = gl::add(s[i], p.rc_ext[r * 12 + i]);
        poseidon::sbox7_layer(s);
        poseidon2::ext_layer_qp(s);
    }
    for (int i = 0; i < 12; i++) s[i] = gl::canon(s[i]);
    return true;
}
template <class F, class Gf>
inline bool selfcheck_with(F plain, Gf emu, int n) {
    u64 seed = 0x243F6A8885A308D3ull;
    for (int t = 0; t < n; t++) {
        u64 a[12], b[12];
        for (int i = 0; i < 12; i++) {
            seed = seed * 6364136223846793005ull + 1442695040888963407ull;
            u64 v = seed ^ (seed >> 29);
            if (t % 7 == 0) v = (i & 1) ? ~0ull - (v & 0xFF) : (v & 0xFF);            // extremes: near 0 and near 2^64
            if (t % 11 == 0) v = gl::P - 1 - (v & 3);
            a[i] = b[i] = v;
        }
        plain(a);
        if (!emu(b)) return false;
        for (int i = 0; i < 12; i++) if (a[i] != b[i]) return false;
    }
    return true;
}
inline bool selfcheck(const u64 *rc, const unsigned char *tab, int n = 2000) {
    return selfcheck_with([&](u64 (&s)[12]) { poseidon::permute(s, rc); }, [&](u64 (&s)[12]) { return emu_permute(s, rc, tab); }, n);
}
inline bool selfcheck_gate(const u64 *rcp, const unsigned char *tab, int n = 2000) {
    return selfcheck_with([&](u64 (&s)[12]) { for (int i = 0; i < 12; i++) s[i] = gl::canon(s[i]); poseidon::permute_textbook(s, rcp); },
                          [&](u64 (&s)[12]) { for (int i = 0; i < 12; i++) s[i] = gl::canon(s[i]); return emu_permute_textbook(s, rcp, tab); }, n);
}
inline bool selfcheck_p2(const poseidon2::Params &p, const unsigned char *tab, int n = 2000) {
    return selfcheck_with([&](u64 (&s)[12]) { poseidon2::permute_qp(s, p); }, [&](u64 (&s)[12]) { return emu_permute_p2(s, p, tab); }, n);
}
}  // namespace host
inline bool build_tables(const u64 *rc, unsigned char *tab) { return host::build_tables(rc, tab); }
inline bool host_selfcheck(const u64 *rc, const unsigned char *tab, int n = 2000) { return host::selfcheck(rc, tab, n); }
inline bool build_tables_gate(const u64 *rcp, unsigned char *tab) { return host::build_tables_gate(rcp, tab); }
inline bool host_selfcheck_gate(const u64 *rcp, const unsigned char *tab, int n = 2000) { return host::selfcheck_gate(rcp, tab, n); }
inline bool build_tables_p2(const poseidon2::Params &p, unsigned char *tab) { return host::build_tables_p2(p, tab); }
inline bool host_selfcheck_p2(const poseidon2::Params &p, const unsigned char *tab, int n = 2000) { return host::selfcheck_p2(p, tab, n); }
#else
bool build_tables(const u64 *, unsigned char *);          // host functions: declared only in the device pass
bool host_selfcheck(const u64 *, const unsigned char *, int n = 2000);
bool build_tables_p2(const poseidon2::Params &, unsigned char *);
bool build_tables_gate(const u64 *, unsigned char *);
bool host_selfcheck_gate(const u64 *, const unsigned char *, int n = 2000);
bool host_selfcheck_p2(const poseidon2::Params &, const unsigned char *, int n = 2000);
#endif

#if defined(__HIP_DEVICE_COMPILE__)
// ---- device ----
__device__ __forceinline__ void swap32(u32 &x, u32 &y) {     // lanes 32..63 of x <-> lanes 0..31 of y
    auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
    x = r[0]; y = r[1];
}
struct Run {
    u32 Dlo[N_ELEM], Dhi[N_ELEM];     // digits, natural layout until their K-step is formed
    v4i B0[FIN_STEPS], B1[FIN_STEPS];  // B operands of the columns of lanes 0..31 / 32..63
    u64 y[24];
    u64 Gd[3];                         // the running group's coefficients when they are not Poseidon's three small integers
    const unsigned char *lds;
};
__device__ __forceinline__ void form_step(Run &st, int q) {
    swap32(st.Dlo[4 * q], st.Dlo[4 * q + 2]); swap32(st.Dhi[4 * q], st.Dhi[4 * q + 2]);
    swap32(st.Dlo[4 * q + 1], st.Dlo[4 * q + 3]); swap32(st.Dhi[4 * q + 1], st.Dhi[4 * q + 3]);
    st.B0[q] = v4i{(int)st.Dlo[4 * q], (int)st.Dhi[4 * q], (int)st.Dlo[4 * q + 1], (int)st.Dhi[4 * q + 1]};
    st.B1[q] = v4i{(int)st.Dlo[4 * q + 2], (int)st.Dhi[4 * q + 2], (int)st.Dlo[4 * q + 3], (int)st.Dhi[4 * q + 3]};
}
// one tile: Z[o] = the 8 limbs of output o for this lane's column
template <int STEPS>
__device__ __forceinline__ void gemm(const Run &st, int tile0, int cidx, u32 (&Z)[4][LIMBS]) {
    const int lane = threadIdx.x & 63;
    const v4i *cp = (const v4i *)(st.lds + CINIT_OFF + cidx * 128 + (lane >> 5) * 64);
    v16i a0, a1;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const v4i c = cp[i];
        a0[4 * i] = c[0]; a0[4 * i + 1] = c[1]; a0[4 * i + 2] = c[2]; a0[4 * i + 3] = c[3];
    }
    a1 = a0;
    const v4i *ap = (const v4i *)(st.lds + (size_t)tile0 * 1024) + lane;
#pragma unroll
    for (int q = 0; q < STEPS; q++) {
        const v4i a = ap[q * 64];
        a0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, st.B0[q], a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, st.B1[q], a1, 0, 0, 0);
    }
#ifndef PMF_NO_KEEPALIVE   // (defined only by tools/mfma_hazard_check.py --first-layout, to show what the compiler does without it)
    asm volatile("" : "+v"(a0), "+v"(a1));    // both destinations stay allocated as a whole until the chain has been issued
#endif
#pragma unroll
    for (int i = 0; i < 16; i++) {
        u32 x = (u32)a0[i], yv = (u32)a1[i];
        swap32(x, yv);
        Z[i >> 3][i & 7] = x; Z[2 + (i >> 3)][i & 7] = yv;
    }
}
// what a round does with its S-box input: the permutation sends it through the S-box; a gate's quotient kernel compares it with
// the wire that holds it and sends the WIRE through (poseidon quotient kernels)
struct SboxOfInput { __device__ __forceinline__ u64 operator()(int, u64 x) const { return poseidon::sbox7_lane(x); } };
template <int G, bool SMALL_G, class F>
__device__ __forceinline__ void group_step(Run &st, F &f) {
    u32 Z[4][LIMBS];
    gemm<group_steps(G)>(st, group_base(G), G, Z);
    constexpr int n = 4 * G + 4 <= 22 ? 4 : 22 - 4 * G;
#pragma unroll
    for (int o = 0; o < n; o++) {
        const int k = 4 * G + o;
        u64 x = recombine(Z[o]);
        constexpr u32 Gc[3] = {G0, G1, G2};
#pragma unroll
        for (int j = 4 * G; j < k; j++) x = gl::add(x, SMALL_G ? mul_small(st.y[j], Gc[k - 1 - j]) : gl::mul(st.y[j], st.Gd[k - 1 - j]));
        st.y[k] = f(k, x);
    }
#pragma unroll
    for (int o = 0; o < n; o++) {
        const u64 t = to_digits(st.y[4 * G + o]);
        st.Dlo[12 + 4 * G + o] = (u32)t; st.Dhi[12 + 4 * G + o] = (u32)(t >> 32);
    }
    form_step(st, 3 + G);
}
template <bool SMALL_G, class F, int... G>
__device__ __forceinline__ void all_groups(Run &st, F &f, std::integer_sequence<int, G...>) { (group_step<G, SMALL_G>(st, f), ...); }

// s: the S-box outputs of the fourth full round (its MDS layer is part of the matrix) -> the state the S-box layer of the first
// closing full round applies to (that round's constants are already in)
template <bool SMALL_G, class F>
__device__ __forceinline__ void partial_rounds(u64 (&s)[12], const unsigned char *lds, F &f) {
    Run st;
    st.lds = lds;
    if (!SMALL_G) {
#pragma unroll
        for (int d = 0; d < 3; d++) {      // wave-uniform: scalar registers
            const u64 g = *(const u64 *)(lds + G_OFF + 8 * d);
            st.Gd[d] = ((u64)(u32)__builtin_amdgcn_readfirstlane((int)(g >> 32)) << 32) | (u32)__builtin_amdgcn_readfirstlane((int)(u32)g);
        }
    }
#pragma unroll
    for (int e = 0; e < N_ELEM; e++) {
        const u64 t = e < 12 ? to_digits(s[e]) : 0;
        st.Dlo[e] = (u32)t; st.Dhi[e] = (u32)(t >> 32);
    }
    form_step(st, 0); form_step(st, 1); form_step(st, 2);
    all_groups<SMALL_G>(st, f, std::make_integer_sequence<int, N_GROUP>{});
#pragma unroll
    for (int f = 0; f < N_FIN; f++) {
        u32 Z[4][LIMBS];
        gemm<FIN_STEPS>(st, N_GROUP_TILES + f * FIN_STEPS, N_GROUP + f, Z);
#pragma unroll
        for (int o = 0; o < 4; o++) s[4 * f + o] = recombine(Z[o]);
    }
}
// poseidon::permute with the partial rounds on the matrix pipe, in two halves so that a caller can place loads between them (the
// closing full rounds need the fewest registers). Every lane of the wave must be here (MFMA and the lane swaps ignore or need the
// whole wave); lds: the table of build_tables, 16-byte aligned.
__device__ __forceinline__ void permute_head(u64 (&s)[12], const u64 *rc, const unsigned char *lds) {
    using namespace poseidon;
    int r = 0;
    for (int k = 0; k < HALF_FULL; k++, r++) {
#pragma unroll
        for (int i = 0; i < WIDTH; i++) s[i] = gl::add_canonical(s[i], rc[r * WIDTH + i]);
        sbox7_layer(s);
        if (k + 1 < HALF_FULL) mds_layer(s);
    }
    SboxOfInput f;
    partial_rounds<true>(s, lds, f);
}
__device__ __forceinline__ void permute_tail(u64 (&s)[12], const u64 *rc) {
    using namespace poseidon;
    int r = HALF_FULL + 1 + PARTIAL;
    sbox7_layer(s);
    mds_layer(s);
    for (int k = 1; k < HALF_FULL; k++, r++) {
#pragma unroll
        for (int i = 0; i < WIDTH; i++) s[i] = gl::add_canonical(s[i], rc[r * WIDTH + i]);
        sbox7_layer(s);
        mds_layer(s);
    }
#pragma unroll
    for (int i = 0; i < WIDTH; i++) s[i] = gl::canon(s[i]);
}
__device__ __forceinline__ void permute(u64 (&s)[12], const u64 *rc, const unsigned char *lds) {
    permute_head(s, rc, lds);
    permute_tail(s, rc);
}
// poseidon2::permute_qp (qp-poseidon-core's parameter set) with the 22 internal rounds, the external layer before them and the
// constants of the external round behind them on the matrix pipe; lds: the table of build_tables_p2 for the same parameters
__device__ __forceinline__ void permute_p2qp(u64 (&s)[12], const poseidon2::Params &p, const unsigned char *lds) {
    poseidon2::ext_layer_qp(s);
    for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int i = 0; i < 12; i++) s[i] = gl::add_canonical(s[i], p.rc_ext[r * 12 + i]);   // the block's constants are canonical (parse_p2, qp_params)
        poseidon::sbox7_layer(s);
        if (r < 3) poseidon2::ext_layer_qp(s);
    }
    SboxOfInput f;
    partial_rounds<false>(s, lds, f);
    poseidon::sbox7_layer(s);
    poseidon2::ext_layer_qp(s);
    for (int r = 5; r < 8; r++) {
#pragma unroll
        for (int i = 0; i < 12; i++) s[i] = gl::add_canonical(s[i], p.rc_ext[r * 12 + i]);   // the block's constants are canonical (parse_p2, qp_params)
        poseidon::sbox7_layer(s);
        poseidon2::ext_layer_qp(s);
    }
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = gl::canon(s[i]);
}
#elif defined(__HIPCC__)
__device__ void permute(u64 (&s)[12], const u64 *rc, const unsigned char *lds);   // host pass of a .hip unit: names only
__device__ void permute_head(u64 (&s)[12], const u64 *rc, const unsigned char *lds);
__device__ void permute_tail(u64 (&s)[12], const u64 *rc);
__device__ void permute_p2qp(u64 (&s)[12], const poseidon2::Params &p, const unsigned char *lds);
template <bool SMALL_G, class F> __device__ void partial_rounds(u64 (&)[12], const unsigned char *, F &) {}   // body: device pass only
#endif

}  // namespace pmf
