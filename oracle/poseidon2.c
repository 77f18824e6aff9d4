/*
 * oracle/poseidon2.c — parametric Poseidon2 (width 12, rate 8, out 4) sponge and the byte<->felt codecs
 * of qp-poseidon-core 3.1.0 (un-vendored, Cargo.lock:929-932 of the reference), restated from the
 * reference's call sites:
 *   - padding rule `input || 1 || 0*` to a RATE multiple: wormhole/circuit/tests/heap_zeroization.rs:133-160
 *   - bytes_to_u64s (4 bytes/felt LE, 0x01 terminator):  common/src/serialization.rs:127-141
 *   - digest <-> 32 bytes (4 x LE u64):                   common/src/serialization.rs:228-247
 *   - double hash for the unspendable account:            wormhole/circuit/src/unspendable_account.rs:63-94
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/gl.h header).
 *
 * The permutation is a plug: callers inject (external round constants 8x12, internal round constants 22, internal diagonal
 * 12, 4x4 external block). qp-poseidon-core's own set is not in the reference tree; the one
 * orc_p2_qp_params derives was found by search and is PINNED by all seven known-answer vectors the reference holds
 * (tests/golden/poseidon2_kats.json; tests/test_oracle_poseidon.py).
 */
#include "gl.h"
#include <string.h>
#include <stdlib.h>

#include "poseidon2.h"

size_t orc_p2_params_size(void) { return sizeof(orc_p2_params); }

static inline gl_t sbox7(gl_t x) { gl_t x2 = gl_sqr(x), x4 = gl_sqr(x2), x3 = gl_mul(x, x2); return gl_mul(x3, x4); }
static void ext_layer(const orc_p2_params *p, gl_t s[12]) {
    gl_t t[12];
    for (int b = 0; b < 3; b++)
        for (int i = 0; i < 4; i++) {
            u128 acc = 0;
            for (int j = 0; j < 4; j++) acc += (u128)p->m4[i][j] * s[4 * b + j];
            t[4 * b + i] = gl_reduce128(acc);
        }
    for (int i = 0; i < 4; i++) {
        gl_t sum = gl_add(gl_add(t[i], t[4 + i]), t[8 + i]);
        for (int b = 0; b < 3; b++) s[4 * b + i] = gl_add(t[4 * b + i], sum);
    }
}
static void int_layer(const orc_p2_params *p, gl_t s[12]) {
    gl_t sum = 0;
    for (int i = 0; i < 12; i++) sum = gl_add(sum, s[i]);
    for (int i = 0; i < 12; i++) s[i] = gl_add(gl_mul(s[i], p->diag_m1[i]), sum);
}
void orc_p2_permute(const orc_p2_params *p, gl_t s[12]) {
    ext_layer(p, s);
    for (int r = 0; r < 4; r++) {
        for (int i = 0; i < 12; i++) s[i] = sbox7(gl_add(s[i], p->rc_ext[r][i]));
        ext_layer(p, s);
    }
    for (int r = 0; r < 22; r++) {
        s[0] = sbox7(gl_add(s[0], p->rc_int[r]));
        int_layer(p, s);
    }
    for (int r = 4; r < 8; r++) {
        for (int i = 0; i < 12; i++) s[i] = sbox7(gl_add(s[i], p->rc_ext[r][i]));
        ext_layer(p, s);
    }
}
/* ---- qp-poseidon-core 3.1.0's parameter set, re-derived here independently of the product ----
 * Established by tools/derivation/p2_search.py against the reference's seven known-answer vectors: Plonky3's
 * Poseidon2Goldilocks<12>::new_from_rng_128 on rand_chacha ChaCha20Rng::seed_from_u64(0x3141592653589793) (rand_core's PCG32
 * expansion of the 64-bit seed into the 256-bit key; 64-bit block counter, zero stream id): 8 x 12 external round constants,
 * then 22 internal ones, each a next_u64() accepted when below p; external block MDSMat4 = circ(2, 3, 1, 1); internal matrix
 * J + diag(MATRIX_DIAG_12_GOLDILOCKS); sponge: `|| 1 || 0*` padding to the rate 8 with ADDITIVE absorption (the 45-element
 * block-header vectors distinguish it from overwrite absorption). */
static uint32_t c20_rotl(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }
static void c20_block(const uint32_t key[8], uint64_t counter, uint32_t out[16]) {
    uint32_t in[16] = {0x61707865, 0x3320646e, 0x79622d32, 0x6b206574, key[0], key[1], key[2], key[3], key[4], key[5], key[6], key[7],
                       (uint32_t)counter, (uint32_t)(counter >> 32), 0, 0}, x[16];
    memcpy(x, in, sizeof x);
#define C20_QR(a, b, c, d) x[a] += x[b]; x[d] = c20_rotl(x[d] ^ x[a], 16); x[c] += x[d]; x[b] = c20_rotl(x[b] ^ x[c], 12); \
                           x[a] += x[b]; x[d] = c20_rotl(x[d] ^ x[a], 8);  x[c] += x[d]; x[b] = c20_rotl(x[b] ^ x[c], 7);
    for (int r = 0; r < 10; r++) {
        C20_QR(0, 4, 8, 12) C20_QR(1, 5, 9, 13) C20_QR(2, 6, 10, 14) C20_QR(3, 7, 11, 15)
        C20_QR(0, 5, 10, 15) C20_QR(1, 6, 11, 12) C20_QR(2, 7, 8, 13) C20_QR(3, 4, 9, 14)
    }
#undef C20_QR
    for (int i = 0; i < 16; i++) out[i] = x[i] + in[i];
}
void orc_p2_qp_params(orc_p2_params *p) {
    uint32_t key[8], buf[16];
    uint64_t st = 0x3141592653589793ULL, ctr = 0;
    for (int i = 0; i < 8; i++) {     /* rand_core::SeedableRng::seed_from_u64 */
        st = st * 6364136223846793005ULL + 11634580027462260723ULL;
        uint32_t xs = (uint32_t)(((st >> 18) ^ st) >> 27), rot = (uint32_t)(st >> 59);
        key[i] = (xs >> rot) | (xs << ((32 - rot) & 31));
    }
    int idx = 16, n = 0;
    gl_t flat[96 + 22];
    while (n < 96 + 22) {
        uint32_t w[2];
        for (int k = 0; k < 2; k++) { if (idx == 16) { c20_block(key, ctr++, buf); idx = 0; } w[k] = buf[idx++]; }
        const uint64_t v = ((uint64_t)w[1] << 32) | w[0];
        if (v < GL_P) flat[n++] = v;
    }
    memcpy(p->rc_ext, flat, sizeof p->rc_ext);
    memcpy(p->rc_int, flat + 96, sizeof p->rc_int);
    static const gl_t diag[12] = {0xc3b6c08e23ba9300ULL, 0xd84b5de94a324fb6ULL, 0x0d0c371c5b35b84fULL, 0x7964f570e7188037ULL,
                                  0x5daf18bbd996604bULL, 0x6743bc47b9595257ULL, 0x5528b9362c59bb70ULL, 0xac45e25b7127b68bULL,
                                  0xa2077d7dfbb606b5ULL, 0xf3faac6faee378aeULL, 0x0c6388b51545e883ULL, 0xd27dbb6944917b60ULL};
    memcpy(p->diag_m1, diag, sizeof diag);
    static const gl_t m4[4][4] = {{2, 3, 1, 1}, {1, 2, 3, 1}, {1, 1, 2, 3}, {3, 1, 1, 2}};
    memcpy(p->m4, m4, sizeof m4);
    p->absorb_add = 1;
}

/* Poseidon2Hash::hash_no_pad of the qp fork: pads `|| 1 || 0*` to a multiple of 8 */
void orc_p2_hash_pad10(const orc_p2_params *p, const gl_t *in, size_t n, gl_t out[4]) {
    size_t padded = ((n + 1 + 7) / 8) * 8;
    gl_t *buf = (gl_t *)calloc(padded, sizeof(gl_t));
    memcpy(buf, in, n * sizeof(gl_t));
    buf[n] = 1;
    gl_t st[12] = {0};
    for (size_t i = 0; i < padded; i += 8) {
        for (int j = 0; j < 8; j++) st[j] = p->absorb_add ? gl_add(st[j], buf[i + j]) : buf[i + j];
        orc_p2_permute(p, st);
    }
    memcpy(out, st, 4 * sizeof(gl_t));
    memset(buf, 0, padded * sizeof(gl_t));
    free(buf);
}

/* ---- codecs ---- */
/* returns number of felts written; out must hold len/4 + 1 */
size_t orc_bytes_to_u64s(const uint8_t *in, size_t len, uint64_t *out) {
    size_t total = len + 1, padded = (total + 3) / 4 * 4, n = padded / 4;
    for (size_t i = 0; i < n; i++) {
        uint32_t v = 0;
        for (int k = 0; k < 4; k++) {
            size_t idx = i * 4 + k;
            uint8_t b = idx < len ? in[idx] : (idx == len ? 0x01 : 0x00);
            v |= (uint32_t)b << (8 * k);
        }
        out[i] = v;
    }
    return n;
}
void orc_bytes_to_digest(const uint8_t in[32], gl_t out[4]) {
    for (int i = 0; i < 4; i++) {
        uint64_t v = 0;
        for (int k = 0; k < 8; k++) v |= (uint64_t)in[i * 8 + k] << (8 * k);
        out[i] = gl_from_u64(v);
    }
}
void orc_digest_to_bytes(const gl_t in[4], uint8_t out[32]) {
    for (int i = 0; i < 4; i++)
        for (int k = 0; k < 8; k++) out[i * 8 + k] = (uint8_t)(in[i] >> (8 * k));
}
void orc_u64_to_felts(uint64_t v, gl_t out[2]) { out[0] = v >> 32; out[1] = v & 0xFFFFFFFFULL; }
