// poseidon_constants.cpp — start-up derivation of the Poseidon round constants (host).
// Upstream plonky2 generated ALL_ROUND_CONSTANTS with ChaCha8Rng::seed_from_u64(0) and
// Rng::gen_range(0..ORDER) (rand 0.8: widening multiply, reject when the low word exceeds the zone).
#include <cstring>
#include <mutex>
#include <vector>
#include "poseidon.hpp"

namespace poseidon {
struct ChaCha8 {
    int double_rounds = 4;     // ChaCha8; 10 = ChaCha20
    uint32_t key[8];
    uint64_t counter = 0;
    uint32_t block[16];
    int pos = 16;
    static uint32_t rotl(uint32_t v, int n) { return (v << n) | (v >> (32 - n)); }
    static void quarter(uint32_t *x, int a, int b, int c, int d) {
        x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 16);
        x[c] += x[d]; x[b] = rotl(x[b] ^ x[c], 12);
        x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 8);
        x[c] += x[d]; x[b] = rotl(x[b] ^ x[c], 7);
    }
    explicit ChaCha8(uint64_t seed, int rounds = 8) : double_rounds(rounds / 2) {
        // rand_core::SeedableRng::seed_from_u64: PCG32 stream fills the 32-byte key
        uint64_t st = seed;
        for (int i = 0; i < 8; i++) {
            st = st * 6364136223846793005ULL + 11634580027462260723ULL;
            uint32_t xs = (uint32_t)(((st >> 18) ^ st) >> 27), rot = (uint32_t)(st >> 59);
            key[i] = (xs >> rot) | (xs << ((32 - rot) & 31));
        }
    }
    void refill() {
        uint32_t in[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u};
        std::memcpy(in + 4, key, sizeof key);
        in[12] = (uint32_t)counter; in[13] = (uint32_t)(counter >> 32); in[14] = 0; in[15] = 0;
        uint32_t x[16];
        std::memcpy(x, in, sizeof x);
        for (int dr = 0; dr < double_rounds; dr++) {
            quarter(x, 0, 4, 8, 12); quarter(x, 1, 5, 9, 13); quarter(x, 2, 6, 10, 14); quarter(x, 3, 7, 11, 15);
            quarter(x, 0, 5, 10, 15); quarter(x, 1, 6, 11, 12); quarter(x, 2, 7, 8, 13); quarter(x, 3, 4, 9, 14);
        }
        for (int i = 0; i < 16; i++) block[i] = x[i] + in[i];
        counter++; pos = 0;
    }
    uint32_t next32() { if (pos == 16) refill(); return block[pos++]; }
    uint64_t next64() { uint64_t lo = next32(); uint64_t hi = next32(); return (hi << 32) | lo; }
    uint64_t below(uint64_t range) {
        const uint64_t zone = (range << __builtin_clzll(range)) - 1;
        for (;;) {
            unsigned __int128 m = (unsigned __int128)next64() * range;
            if ((uint64_t)m <= zone) return (uint64_t)(m >> 64);
        }
    }
};
namespace {
u64 g_rc[ROUNDS * WIDTH];
std::once_flag g_once;
}  // namespace

void derive_round_constants(u64 *out360) {
    ChaCha8 rng(0);
    for (int i = 0; i < ROUNDS * WIDTH; i++) out360[i] = rng.below(gl::P);
}
const u64 *host_round_constants() {
    std::call_once(g_once, [] { derive_round_constants(g_rc); });
    return g_rc;
}

// ---- the hashing kernels' table: partial-round constants pushed onto lane 0 (poseidon.hpp: permute) ----
namespace {
u64 g_rc_hash[ROUNDS * WIDTH];
std::once_flag g_hash_once;
void derive_hash_constants() {
    const u64 *rc = host_round_constants();
    std::memcpy(g_rc_hash, rc, sizeof g_rc_hash);
    // x_r = s_r + c_r. With u_r = s_r + b_r - d_r (b_r = c_r without lane 0, d_r the carried vector, d_0 = 0):
    //   u_{r+1} = M sbox0(u_r + k_r e0) + alpha_{r+1} e0,   M d_r + b_{r+1} = alpha_{r+1} e0 + d_{r+1},   k_r = c_r[0] + alpha_r
    u64 d[WIDTH] = {0};
    u64 alpha = 0;
    for (int r = 0; r < PARTIAL; r++) {
        u64 *row = g_rc_hash + (HALF_FULL + r) * WIDTH;
        const u64 *c = rc + (HALF_FULL + r) * WIDTH;
        row[0] = gl::canon(gl::add(c[0], alpha));
        for (int i = 1; i < WIDTH; i++) row[i] = r == 0 ? c[i] : 0;
        // M d_r + b_{r+1}
        u64 md[WIDTH];
        for (int i = 0; i < WIDTH; i++) md[i] = d[i];
        mds_layer_naive(md);
        if (r + 1 < PARTIAL) {
            const u64 *cn = rc + (HALF_FULL + r + 1) * WIDTH;
            alpha = gl::canon(md[0]);
            d[0] = 0;
            for (int i = 1; i < WIDTH; i++) d[i] = gl::canon(gl::add(md[i], cn[i]));
        } else {
            // s after the last partial round = M sbox0(..) + M d_21: fold M d_21 into the next full round's constants
            u64 *nxt = g_rc_hash + (HALF_FULL + PARTIAL) * WIDTH;
            for (int i = 0; i < WIDTH; i++) nxt[i] = gl::canon(gl::add(nxt[i], gl::canon(md[i])));
        }
    }
}
}  // namespace
const u64 *host_hash_round_constants() {
    std::call_once(g_hash_once, derive_hash_constants);
    return g_rc_hash;
}

// ---- fast partial rounds: the published HADES optimisation, written for column vectors ----
namespace {
typedef std::vector<std::vector<u64>> Mat;
Mat mat_mul(const Mat &a, const Mat &b) {
    Mat c(a.size(), std::vector<u64>(b[0].size(), 0));
    for (size_t i = 0; i < a.size(); i++)
        for (size_t j = 0; j < b[0].size(); j++) {
            u64 acc = 0;
            for (size_t k = 0; k < b.size(); k++) acc = gl::add(acc, gl::mul(a[i][k], b[k][j]));
            c[i][j] = gl::canon(acc);
        }
    return c;
}
std::vector<u64> mat_vec(const Mat &a, const std::vector<u64> &x) {
    std::vector<u64> y(a.size());
    for (size_t i = 0; i < a.size(); i++) { u64 acc = 0; for (size_t k = 0; k < x.size(); k++) acc = gl::add(acc, gl::mul(a[i][k], x[k])); y[i] = gl::canon(acc); }
    return y;
}
Mat mat_transpose(const Mat &a) { Mat t(a[0].size(), std::vector<u64>(a.size())); for (size_t i = 0; i < a.size(); i++) for (size_t j = 0; j < a[0].size(); j++) t[j][i] = a[i][j]; return t; }
Mat mat_inverse(Mat a) {   // Gauss-Jordan over the field
    const size_t n = a.size();
    Mat inv(n, std::vector<u64>(n, 0));
    for (size_t i = 0; i < n; i++) inv[i][i] = 1;
    for (size_t c = 0; c < n; c++) {
        size_t piv = c;
        while (gl::canon(a[piv][c]) == 0) piv++;
        std::swap(a[c], a[piv]); std::swap(inv[c], inv[piv]);
        const u64 f = gl::inv(a[c][c]);
        for (size_t j = 0; j < n; j++) { a[c][j] = gl::canon(gl::mul(a[c][j], f)); inv[c][j] = gl::canon(gl::mul(inv[c][j], f)); }
        for (size_t r = 0; r < n; r++) {
            if (r == c || a[r][c] == 0) continue;
            const u64 g = a[r][c];
            for (size_t j = 0; j < n; j++) {
                a[r][j] = gl::canon(gl::sub(a[r][j], gl::mul(g, a[c][j])));
                inv[r][j] = gl::canon(gl::sub(inv[r][j], gl::mul(g, inv[c][j])));
            }
        }
    }
    return inv;
}
u64 g_fp[FP_WORDS];
std::once_flag g_fp_once;
void derive_fast_partial() {
    const u64 *rc = host_round_constants();
    const u64 CIRC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    Mat M(12, std::vector<u64>(12));
    for (int r = 0; r < 12; r++) for (int c = 0; c < 12; c++) M[r][c] = CIRC[(c - r + 12) % 12] + (r == 0 && c == 0 ? 8 : 0);
    // constants: move the constants of partial round k+1 up behind the S-box of round k
    const Mat Minv = mat_inverse(M);
    std::vector<std::vector<u64>> C(22);
    for (int k = 0; k < 22; k++) C[k].assign(rc + (4 + k) * 12, rc + (5 + k) * 12);
    u64 post[22] = {0};
    for (int k = 20; k >= 0; k--) {
        const std::vector<u64> w = mat_vec(Minv, C[k + 1]);
        for (int i = 1; i < 12; i++) C[k][i] = gl::canon(gl::add(C[k][i], w[i]));
        post[k] = w[0];
    }
    for (int i = 0; i < 12; i++) g_fp[FP_FIRST + i] = C[0][i];
    for (int k = 0; k < 22; k++) g_fp[FP_RC + k] = k < 21 ? post[k] : 0;
    // matrices: Mmul = M'' * M', M' = diag(1, B) moves in front of the round's S-box (into the previous linear layer)
    Mat Mmul = M, Mi;
    for (int i = 21; i >= 0; i--) {
        Mat B(11, std::vector<u64>(11));
        std::vector<u64> col(11), row(11);
        for (int r = 0; r < 11; r++) { col[r] = Mmul[r + 1][0]; row[r] = Mmul[0][r + 1]; for (int c = 0; c < 11; c++) B[r][c] = Mmul[r + 1][c + 1]; }
        const std::vector<u64> what = mat_vec(mat_inverse(mat_transpose(B)), row);
        for (int k = 0; k < 11; k++) { g_fp[FP_VS + i * 11 + k] = col[k]; g_fp[FP_WHATS + i * 11 + k] = what[k]; }
        Mi.assign(12, std::vector<u64>(12, 0));
        Mi[0][0] = 1;
        for (int r = 0; r < 11; r++) for (int c = 0; c < 11; c++) Mi[r + 1][c + 1] = B[r][c];
        Mmul = mat_mul(Mi, M);
    }
    for (int c = 0; c < 11; c++) for (int r = 0; r < 11; r++) g_fp[FP_INIT + c * 11 + r] = Mi[c + 1][r + 1];
}
}  // namespace
const u64 *host_fast_partial() {
    std::call_once(g_fp_once, derive_fast_partial);
    return g_fp;
}
}  // namespace poseidon

// ---- qp-poseidon-core 3.1.0's Poseidon2 parameters (the application hash of the Wormhole circuits) ----
// Not in the reference tree; established by search against the reference's seven known-answer vectors
// (tools/derivation/p2_search.py, tests/golden/poseidon2_kats.json): Plonky3's Poseidon2Goldilocks<12>::new_from_rng_128 on
// rand_chacha's ChaCha20Rng::seed_from_u64(0x3141592653589793): 8 x 12 external round constants first, then the 22 internal
// ones, every element drawn by rejection (next_u64 below p); external 4x4 block = Plonky3's MDSMat4 circ(2, 3, 1, 1);
// internal matrix J + diag(MATRIX_DIAG_12_GOLDILOCKS).
namespace poseidon2 {
const Params &qp_params() {
    static Params p;
    static std::once_flag once;
    std::call_once(once, [] {
        poseidon::ChaCha8 rng(0x3141592653589793ULL, 20);
        auto draw = [&]() { for (;;) { const gl::u64 v = rng.next64(); if (v < gl::P) return v; } };
        for (int i = 0; i < 96; i++) p.rc_ext[i] = draw();
        for (int i = 0; i < 22; i++) p.rc_int[i] = draw();
        static const gl::u64 diag[12] = {0xc3b6c08e23ba9300ULL, 0xd84b5de94a324fb6ULL, 0x0d0c371c5b35b84fULL, 0x7964f570e7188037ULL,
                                         0x5daf18bbd996604bULL, 0x6743bc47b9595257ULL, 0x5528b9362c59bb70ULL, 0xac45e25b7127b68bULL,
                                         0xa2077d7dfbb606b5ULL, 0xf3faac6faee378aeULL, 0x0c6388b51545e883ULL, 0xd27dbb6944917b60ULL};
        for (int i = 0; i < 12; i++) p.diag_m1[i] = diag[i];
        // circ(2, 3, 1, 1): poseidon2::ext_layer_qp (poseidon.hpp) is this block without multiplications
        static const gl::u64 m4[16] = {2, 3, 1, 1, 1, 2, 3, 1, 1, 1, 2, 3, 3, 1, 1, 2};
        for (int i = 0; i < 16; i++) p.m4[i] = m4[i];
    });
    return p;
}
}  // namespace poseidon2

// ---- hasher selection (see poseidon.hpp) ----
namespace hasher {
namespace {
Config g_default{};
}  // namespace
const Config &process_default() { return g_default; }
void set_process_default(int k, const poseidon2::Params *p) {
    if (p) g_default.p2 = *p;
    g_default.kind = k;
}
void Config::permute(gl::u64 (&s)[12]) const {
    if (kind == POSEIDON2) poseidon2::permute(s, p2);
    else poseidon::permute(s, poseidon::host_hash_round_constants());
}
}  // namespace hasher
