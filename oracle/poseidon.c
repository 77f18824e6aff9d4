/*
 * oracle/poseidon.c — CPU restatement of plonky2::hash::{poseidon, hashing, merkle_tree} and
 * plonky2::iop::challenger (qp-plonky2 1.5.5, un-vendored; the proof-system hasher of
 * PoseidonGoldilocksConfig, reference common/src/circuit.rs:17).
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/gl.h header).
 *
 * Constants: upstream plonky2's ALL_ROUND_CONSTANTS are the first 360 outputs of
 * ChaCha8Rng::seed_from_u64(0).gen_range(0..p) (upstream generate_constants tool). They are
 * re-derived here at start-up instead of being tabulated; the derivation is pinned by
 * tests/golden/poseidon_v1.json (first constants 0xb585f766f2144405, ... and the upstream
 * permutation test vectors for inputs 0^12 and 0..11).
 * MDS: circulant [17,15,41,16,2,28,13,13,39,18,34,20] plus diag [8,0,...,0];
 * rounds: 4 full, 22 partial, 4 full; S-box x^7.
 *
 * Sponge / Merkle / challenger conventions: SURVEY.md Appendix A.3.
 */
#include "gl.h"
#include "challenger.h"
#include <stdlib.h>
#include <string.h>

#define SPONGE_WIDTH 12
#define SPONGE_RATE 8
#define N_FULL_HALF 4
#define N_PARTIAL 22
#define N_ROUNDS (2 * N_FULL_HALF + N_PARTIAL)

static gl_t RC[N_ROUNDS * SPONGE_WIDTH];
static const gl_t MDS_CIRC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
static const gl_t MDS_DIAG[12] = {8, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
static int rc_ready = 0;

/* ---- ChaCha8 block function + rand_core/rand 0.8 sampling, restated ---- */
static inline uint32_t rotl32(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }
#define QR(a, b, c, d) \
    a += b; d = rotl32(d ^ a, 16); c += d; b = rotl32(b ^ c, 12); \
    a += b; d = rotl32(d ^ a, 8);  c += d; b = rotl32(b ^ c, 7);
static void chacha8_block(const uint32_t key[8], uint64_t counter, uint32_t out[16]) {
    uint32_t in[16] = {0x61707865, 0x3320646e, 0x79622d32, 0x6b206574,
                       key[0], key[1], key[2], key[3], key[4], key[5], key[6], key[7],
                       (uint32_t)counter, (uint32_t)(counter >> 32), 0, 0};
    uint32_t x[16];
    memcpy(x, in, sizeof x);
    for (int r = 0; r < 4; r++) {
        QR(x[0], x[4], x[8], x[12]) QR(x[1], x[5], x[9], x[13])
        QR(x[2], x[6], x[10], x[14]) QR(x[3], x[7], x[11], x[15])
        QR(x[0], x[5], x[10], x[15]) QR(x[1], x[6], x[11], x[12])
        QR(x[2], x[7], x[8], x[13]) QR(x[3], x[4], x[9], x[14])
    }
    for (int i = 0; i < 16; i++) out[i] = x[i] + in[i];
}
typedef struct { uint32_t key[8]; uint64_t ctr; uint32_t buf[16]; int idx; } chacha_rng;
static void rng_seed_from_u64(chacha_rng *r, uint64_t state) {
    const uint64_t MUL = 6364136223846793005ULL, INC = 11634580027462260723ULL;
    for (int i = 0; i < 8; i++) {     /* PCG32 expansion of the 64-bit seed */
        state = state * MUL + INC;
        uint32_t xs = (uint32_t)(((state >> 18) ^ state) >> 27);
        uint32_t rot = (uint32_t)(state >> 59);
        r->key[i] = (xs >> rot) | (xs << ((32 - rot) & 31));
    }
    r->ctr = 0; r->idx = 16;
}
static uint32_t rng_u32(chacha_rng *r) {
    if (r->idx == 16) { chacha8_block(r->key, r->ctr++, r->buf); r->idx = 0; }
    return r->buf[r->idx++];
}
static uint64_t rng_u64(chacha_rng *r) { uint64_t lo = rng_u32(r), hi = rng_u32(r); return (hi << 32) | lo; }
static uint64_t rng_range(chacha_rng *r, uint64_t range) { /* rand 0.8 UniformInt::sample_single */
    uint64_t zone = (range << __builtin_clzll(range)) - 1;
    for (;;) {
        u128 m = (u128)rng_u64(r) * range;
        if ((uint64_t)m <= zone) return (uint64_t)(m >> 64);
    }
}
static void init_rc(void) {
    if (rc_ready) return;
    chacha_rng r; rng_seed_from_u64(&r, 0);
    for (int i = 0; i < N_ROUNDS * SPONGE_WIDTH; i++) RC[i] = rng_range(&r, GL_P);
    rc_ready = 1;
}
void orc_poseidon_round_constants(gl_t *out) { init_rc(); memcpy(out, RC, sizeof RC); }

/* ---- permutation (naive HADES schedule; upstream's fast partial rounds compute the same map) ---- */
static inline gl_t sbox7(gl_t x) { gl_t x2 = gl_sqr(x), x4 = gl_sqr(x2), x3 = gl_mul(x, x2); return gl_mul(x3, x4); }
/* definition: row r of the MDS matrix is the circulant shifted by r, plus the diagonal */
static void mds_layer_naive(gl_t s[12]) {
    gl_t o[12];
    for (int r = 0; r < 12; r++) {
        u128 acc = 0;
        for (int i = 0; i < 12; i++) acc += (u128)s[(i + r) % 12] * MDS_CIRC[i];
        acc += (u128)s[r] * MDS_DIAG[r];
        o[r] = gl_reduce128(acc);
    }
    memcpy(s, o, sizeof o);
}
/* same map, the way CPU implementations organise it: entries are < 2^6, so the 32-bit halves of the state are
   combined in plain 64-bit accumulators (12 * 2^32 * 41 < 2^42) and each output is reduced once */
static void mds_layer(gl_t s[12]) {
    uint64_t lo[24], hi[24];
    for (int i = 0; i < 12; i++) { lo[i] = lo[i + 12] = (uint32_t)s[i]; hi[i] = hi[i + 12] = s[i] >> 32; }
    for (int r = 0; r < 12; r++) {
        uint64_t al = 0, ah = 0;
        for (int i = 0; i < 12; i++) { al += lo[i + r] * MDS_CIRC[i]; ah += hi[i + r] * MDS_CIRC[i]; }
        if (r == 0) { al += lo[0] * MDS_DIAG[0]; ah += hi[0] * MDS_DIAG[0]; }
        s[r] = gl_reduce128((u128)al + ((u128)ah << 32));
    }
}
void orc_poseidon_permute_naive(gl_t s[12]) {
    init_rc();
    int rc = 0;
    for (int r = 0; r < N_FULL_HALF; r++, rc++) {
        for (int i = 0; i < 12; i++) s[i] = sbox7(gl_add(s[i], RC[rc * 12 + i]));
        mds_layer_naive(s);
    }
    for (int r = 0; r < N_PARTIAL; r++, rc++) {
        for (int i = 0; i < 12; i++) s[i] = gl_add(s[i], RC[rc * 12 + i]);
        s[0] = sbox7(s[0]);
        mds_layer_naive(s);
    }
    for (int r = 0; r < N_FULL_HALF; r++, rc++) {
        for (int i = 0; i < 12; i++) s[i] = sbox7(gl_add(s[i], RC[rc * 12 + i]));
        mds_layer_naive(s);
    }
}
/* the proof-system permutation is a plug: when a Poseidon2 parameter block has been selected,
   every sponge, Merkle node, challenger duplexing and proof-of-work evaluation of the oracle uses it instead */
static const void *g_p2_plug = 0;
void orc_p2_permute(const void *p, gl_t s[12]);
void orc_select_hasher_p2(const void *params) { g_p2_plug = params; }   /* NULL: back to Poseidon */
void orc_poseidon_permute_fast(gl_t s[12]);     /* prove.c: the same map in plonky2's fast-partial-round organisation */
void orc_poseidon_permute(gl_t s[12]) {
    if (g_p2_plug) { orc_p2_permute(g_p2_plug, s); return; }
    orc_poseidon_permute_fast(s);
}
/* the textbook schedule with the 32-bit-halves MDS layer (kept: the tests hold the three forms against each other) */
void orc_poseidon_permute_textbook(gl_t s[12]) {
    init_rc();
    int rc = 0;
    for (int r = 0; r < N_FULL_HALF; r++, rc++) {
        for (int i = 0; i < 12; i++) s[i] = sbox7(gl_add(s[i], RC[rc * 12 + i]));
        mds_layer(s);
    }
    for (int r = 0; r < N_PARTIAL; r++, rc++) {
        for (int i = 0; i < 12; i++) s[i] = gl_add(s[i], RC[rc * 12 + i]);
        s[0] = sbox7(s[0]);
        mds_layer(s);
    }
    for (int r = 0; r < N_FULL_HALF; r++, rc++) {
        for (int i = 0; i < 12; i++) s[i] = sbox7(gl_add(s[i], RC[rc * 12 + i]));
        mds_layer(s);
    }
}

/* ---- hashing (overwrite-mode sponge, no padding) ---- */
void orc_hash_n_to_m_no_pad(const gl_t *in, size_t n, gl_t *out, size_t m) {
    gl_t st[12] = {0};
    for (size_t i = 0; i < n; i += SPONGE_RATE) {
        size_t len = n - i < SPONGE_RATE ? n - i : SPONGE_RATE;
        memcpy(st, in + i, len * sizeof(gl_t));
        orc_poseidon_permute(st);
    }
    size_t got = 0;
    for (;;) {
        for (int i = 0; i < SPONGE_RATE; i++) { out[got++] = st[i]; if (got == m) return; }
        orc_poseidon_permute(st);
    }
}
void orc_hash_no_pad(const gl_t *in, size_t n, gl_t out[4]) { orc_hash_n_to_m_no_pad(in, n, out, 4); }
void orc_hash_or_noop(const gl_t *in, size_t n, gl_t out[4]) {
    if (n <= 4) { memset(out, 0, 4 * sizeof(gl_t)); memcpy(out, in, n * sizeof(gl_t)); }
    else orc_hash_no_pad(in, n, out);
}
void orc_two_to_one(const gl_t l[4], const gl_t r[4], gl_t out[4]) {
    gl_t st[12] = {l[0], l[1], l[2], l[3], r[0], r[1], r[2], r[3], 0, 0, 0, 0};
    orc_poseidon_permute(st);
    memcpy(out, st, 4 * sizeof(gl_t));
}

/*
 * MerkleTree::new(leaves, cap_height). leaves: row-major [n_leaves][width].
 * digests_out: level-ordered array. Level 0 = leaf digests (n_leaves*4), level 1 = n_leaves/2 ... down to
 * the cap level (2^cap_height digests), concatenated. cap_out = last level. Returns total digests written.
 * (upstream stores digests in a different in-memory order; only cap + authentication paths are
 *  observable, and those are taken from this level structure by orc_merkle_path.)
 */
size_t orc_merkle_build(const gl_t *leaves, size_t n_leaves, size_t width, unsigned cap_height,
                        gl_t *digests_out, gl_t *cap_out) {
    size_t cap_n = (size_t)1 << cap_height;
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)n_leaves; i++) orc_hash_or_noop(leaves + (size_t)i * width, width, digests_out + (size_t)i * 4);
    gl_t *prev = digests_out;
    size_t cnt = n_leaves, total = n_leaves;
    while (cnt > cap_n) {
        gl_t *next = prev + cnt * 4;
        size_t nn = cnt / 2;
#pragma omp parallel for schedule(static)
        for (long i = 0; i < (long)nn; i++) orc_two_to_one(prev + (size_t)i * 8, prev + (size_t)i * 8 + 4, next + (size_t)i * 4);
        prev = next; cnt = nn; total += nn;
    }
    memcpy(cap_out, prev, cap_n * 4 * sizeof(gl_t));
    return total;
}
/* siblings from the leaf level up to (excluding) the cap; returns path length */
size_t orc_merkle_path(const gl_t *digests, size_t n_leaves, unsigned cap_height, size_t index, gl_t *path_out) {
    size_t cap_n = (size_t)1 << cap_height, cnt = n_leaves, len = 0;
    const gl_t *lvl = digests;
    while (cnt > cap_n) {
        memcpy(path_out + len * 4, lvl + (index ^ 1) * 4, 4 * sizeof(gl_t));
        len++; lvl += cnt * 4; cnt >>= 1; index >>= 1;
    }
    return len;
}

/* ---- Challenger (duplex sponge) ---- */
size_t orc_challenger_size(void) { return sizeof(orc_challenger); }
void orc_challenger_init(orc_challenger *c) { memset(c, 0, sizeof *c); }
static void duplex(orc_challenger *c) {
    for (int i = 0; i < c->n_in; i++) c->state[i] = c->in[i];
    c->n_in = 0;
    orc_poseidon_permute(c->state);
    memcpy(c->out, c->state, SPONGE_RATE * sizeof(gl_t));
    c->n_out = SPONGE_RATE;
}
void orc_challenger_observe(orc_challenger *c, const gl_t *x, size_t n) {
    for (size_t i = 0; i < n; i++) {
        c->n_out = 0;
        c->in[c->n_in++] = x[i];
        if (c->n_in == SPONGE_RATE) duplex(c);
    }
}
gl_t orc_challenger_get(orc_challenger *c) {
    if (c->n_in > 0 || c->n_out == 0) duplex(c);
    return c->out[--c->n_out];
}
void orc_challenger_get_n(orc_challenger *c, gl_t *out, size_t n) { for (size_t i = 0; i < n; i++) out[i] = orc_challenger_get(c); }
/* PoW check of SURVEY A.5: absorb the candidate at position n_in of a copy, permute, look at the last rate element */
gl_t orc_challenger_pow_response(const orc_challenger *c, gl_t nonce) {
    gl_t st[12];
    memcpy(st, c->state, sizeof st);
    for (int i = 0; i < c->n_in; i++) st[i] = c->in[i];
    st[c->n_in] = nonce;
    orc_poseidon_permute(st);
    return st[SPONGE_RATE - 1];
}
