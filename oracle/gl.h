/*
 * oracle/gl.h — Goldilocks field F = GF(2^64 - 2^32 + 1) and its quadratic extension F[x]/(x^2-7).
 *
 * TEST INFRASTRUCTURE ONLY. This directory is the CPU restatement ("oracle") of the
 * qp-plonky2 1.5.5 proving path that the HIP backend replaces. Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the product
 * library (libqpgpu.so) never links or calls anything in here.
 *
 * What it restates (reference call sites; the arithmetic itself lives in the un-vendored
 * crate qp-plonky2-field 1.5.5, Cargo.lock:839-927 of the reference):
 *   - field type F:            /root/reference common/src/circuit.rs:18 (GoldilocksField)
 *   - extension degree D = 2:  common/src/circuit.rs:16
 *   - canonical u64 encoding:  common/src/serialization.rs:35-43
 * Published algorithm restated: p = 2^64 - 2^32 + 1, 2^64 = 2^32 - 1 (mod p), 2^96 = -1 (mod p),
 * extension non-residue W = 7, multiplicative generator 7 -> coset shift g = 14293326489335486720,
 * two-adicity 32 with POWER_OF_TWO_GENERATOR = 7277203076849721926 (= g^((p-1)/2^32)).
 *
 * All values handled here are canonical (in [0, p)).
 */
#ifndef ORACLE_GL_H
#define ORACLE_GL_H
#include <stdint.h>
#include <stddef.h>

typedef uint64_t gl_t;
typedef unsigned __int128 u128;

#define GL_P 0xFFFFFFFF00000001ULL
#define GL_EPS 0xFFFFFFFFULL                 /* 2^32 - 1 = 2^64 mod p */
#define GL_MULT_GEN 14293326489335486720ULL  /* coset shift g (SURVEY.md §8 "Parameters") */
#define GL_ROOT_2_32 7277203076849721926ULL  /* primitive 2^32-th root of unity */
#define GL_EXT_W 7ULL                        /* x^2 = 7 */

/* written with masks rather than branches: the conditions are data dependent (about 50 % taken) */
static inline gl_t gl_canon(gl_t x) { return x - (GL_P & (gl_t)-(gl_t)(x >= GL_P)); }

static inline gl_t gl_add(gl_t a, gl_t b) {   /* canonical inputs: a + b < 2p */
    gl_t s = a + b;
    return s - (GL_P & (gl_t)-(gl_t)((s < a) | (s >= GL_P)));
}
static inline gl_t gl_sub(gl_t a, gl_t b) { gl_t d = a - b; return d + (GL_P & (gl_t)-(gl_t)(a < b)); }
static inline gl_t gl_neg(gl_t a) { return a ? GL_P - a : 0; }

/* 128-bit -> field reduction, the three-step form of SURVEY.md Appendix A.1. */
static inline gl_t gl_reduce128(u128 x) {
    uint64_t lo = (uint64_t)x, hi = (uint64_t)(x >> 64);
    uint64_t hi_hi = hi >> 32, hi_lo = hi & GL_EPS;
    uint64_t t0 = lo - hi_hi;
    t0 -= GL_EPS & (uint64_t)-(uint64_t)(lo < hi_hi);   /* borrow: subtract 2^64 mod p */
    uint64_t t1 = hi_lo * GL_EPS;
    uint64_t t2 = t0 + t1;
    t2 += GL_EPS & (uint64_t)-(uint64_t)(t2 < t1);      /* carry */
    return gl_canon(t2);
}
static inline gl_t gl_mul(gl_t a, gl_t b) { return gl_reduce128((u128)a * b); }
static inline gl_t gl_sqr(gl_t a) { return gl_mul(a, a); }

static inline gl_t gl_pow(gl_t b, uint64_t e) {
    gl_t r = 1;
    while (e) { if (e & 1) r = gl_mul(r, b); b = gl_sqr(b); e >>= 1; }
    return r;
}
static inline gl_t gl_inv(gl_t a) { return gl_pow(a, GL_P - 2); }
static inline gl_t gl_exp_pow2(gl_t a, unsigned k) { while (k--) a = gl_sqr(a); return a; }
/* primitive_root_of_unity(k) = ROOT_2_32 ^ (2^(32-k)) */
static inline gl_t gl_root_of_unity(unsigned log_n) { return gl_exp_pow2(GL_ROOT_2_32, 32 - log_n); }
static inline gl_t gl_from_u64(uint64_t x) { return x >= GL_P ? x - GL_P : x; }

/* ---- quadratic extension: a0 + a1*x, serialized [a0, a1] ---- */
typedef struct { gl_t c[2]; } gl2_t;
static inline gl2_t gl2_make(gl_t a, gl_t b) { gl2_t r = {{a, b}}; return r; }
static inline gl2_t gl2_from(gl_t a) { return gl2_make(a, 0); }
static inline gl2_t gl2_add(gl2_t a, gl2_t b) { return gl2_make(gl_add(a.c[0], b.c[0]), gl_add(a.c[1], b.c[1])); }
static inline gl2_t gl2_sub(gl2_t a, gl2_t b) { return gl2_make(gl_sub(a.c[0], b.c[0]), gl_sub(a.c[1], b.c[1])); }
static inline gl2_t gl2_neg(gl2_t a) { return gl2_make(gl_neg(a.c[0]), gl_neg(a.c[1])); }
static inline gl2_t gl2_mul(gl2_t a, gl2_t b) {
    gl_t c0 = gl_add(gl_mul(a.c[0], b.c[0]), gl_mul(GL_EXT_W, gl_mul(a.c[1], b.c[1])));
    gl_t c1 = gl_add(gl_mul(a.c[0], b.c[1]), gl_mul(a.c[1], b.c[0]));
    return gl2_make(c0, c1);
}
static inline gl2_t gl2_scale(gl2_t a, gl_t s) { return gl2_make(gl_mul(a.c[0], s), gl_mul(a.c[1], s)); }
static inline gl2_t gl2_inv(gl2_t a) {
    /* 1/(a0 + a1 x) = (a0 - a1 x) / (a0^2 - 7 a1^2) */
    gl_t n = gl_sub(gl_sqr(a.c[0]), gl_mul(GL_EXT_W, gl_sqr(a.c[1])));
    gl_t ni = gl_inv(n);
    return gl2_make(gl_mul(a.c[0], ni), gl_mul(gl_neg(a.c[1]), ni));
}
static inline gl2_t gl2_pow(gl2_t b, uint64_t e) {
    gl2_t r = gl2_from(1);
    while (e) { if (e & 1) r = gl2_mul(r, b); b = gl2_mul(b, b); e >>= 1; }
    return r;
}
static inline int gl2_eq(gl2_t a, gl2_t b) { return a.c[0] == b.c[0] && a.c[1] == b.c[1]; }

static inline uint32_t bitrev32(uint32_t x, unsigned bits) {
    uint32_t r = 0;
    for (unsigned i = 0; i < bits; i++) { r = (r << 1) | (x & 1); x >>= 1; }
    return r;
}
#endif
