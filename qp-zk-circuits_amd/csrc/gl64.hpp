// gl64.hpp — Goldilocks field arithmetic for gfx950 device code (and the host-side plan builder).
//
// Replaces plonky2::field::goldilocks_field (qp-plonky2-field 1.5.5; reference type alias
// common/src/circuit.rs:18). p = 2^64 - 2^32 + 1. The GPU has no 64x64 multiplier: a product is
// four 32x32->64 v_mad_u64_u32 plus the 2^64 = 2^32-1, 2^96 = -1 folding, all in registers.
//
// Representation: values in registers may be any u64 ("loose"); gl_canon() brings them to [0,p)
// before they are written where the reference would serialise them.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define GL_HD __host__ __device__ __forceinline__
#else
#define GL_HD inline
#endif

namespace gl {

typedef uint64_t u64;
typedef uint32_t u32;

constexpr u64 P = 0xFFFFFFFF00000001ULL;
constexpr u64 EPS = 0xFFFFFFFFULL;  // 2^64 mod p
constexpr u64 MULT_GEN = 14293326489335486720ULL;
constexpr u64 ROOT_2_32 = 7277203076849721926ULL;

GL_HD u64 canon(u64 x) { return x >= P ? x - P : x; }

// GL_MASK_OF_VCC(dst): dst = vcc ? 0xFFFFFFFF : 0 per lane, right behind the carry chain that wrote vcc. gfx940+ has a hazard there
// for the obvious instruction (LLVM GCNHazardRecognizer, "VALU writes SGPR -> VALU reads it as an operand: 2 wait states"): a
// v_cndmask whose other operands are both constants is VOP3-encoded and reads vcc as an explicit operand; the compiler pads its
// own code with s_nop, inline assembly must look after itself. The form used here selects against a register holding all ones
// (asm operand [ones]), which is VOP2-encoded: vcc is its implicit mask, no operand read, no padding. Measured against the padded
// select and against `x - x - borrow` (profiles/r03_rare_fold.txt): the 2^20 NTT passes run 4 % faster than with padding, 1 % faster
// than the unpadded VOP3 select of rounds 1-2; the hashing kernels do not notice; one more live VGPR, no spills.
#define GL_MASK_OF_VCC(dst) "v_cndmask_b32_e32 " dst ", 0, %[ones], vcc"
#define GL_ONES_OPERAND [ones] "v"(0xFFFFFFFFu)
#if defined(__HIP_DEVICE_COMPILE__)
// ---- gfx950 device primitives: 32-bit carry chains (v_add_co/v_addc_co), no 64-bit compares ----
// All take any u64 ("loose") and return loose values; exact for every input.
__device__ __forceinline__ u64 add(u64 a, u64 b) {
    // v_lshl_add_u64 + 64-bit compare measured faster than a v_add_co/v_addc_co chain (tools/valu_rates.hip)
    u64 s = a + b;
    u64 c = s < a ? EPS : 0;
    u64 t = s + c;
    t += (t < c) ? EPS : 0;
    return t;
}
__device__ __forceinline__ u64 sub(u64 a, u64 b) {
    u32 lo, hi, t;
    asm("v_sub_co_u32 %0, vcc, %3, %5\n\t"
        "v_subb_co_u32 %1, vcc, %4, %6, vcc\n\t"
        GL_MASK_OF_VCC("%2") "\n\t"      // borrow -> - (2^32-1)
        "v_sub_co_u32 %0, vcc, %0, %2\n\t"
        "v_subbrev_co_u32 %1, vcc, 0, %1, vcc\n\t"
        GL_MASK_OF_VCC("%2") "\n\t"      // rare second borrow
        "v_sub_co_u32 %0, vcc, %0, %2\n\t"
        "v_subbrev_co_u32 %1, vcc, 0, %1, vcc"
        : "=&v"(lo), "=&v"(hi), "=&v"(t)
        : "v"((u32)a), "v"((u32)(a >> 32)), "v"((u32)b), "v"((u32)(b >> 32)), GL_ONES_OPERAND
        : "vcc");
    return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ u64 neg(u64 a) { return sub(0, a); }

// (hi:lo) 128-bit -> loose u64:  lo - hi_hi + hi_lo*(2^32-1), with the borrow / carry folds.
// The multiply-add by 2^32-1 is one v_mad_u64_u32 (2.9 issue units) instead of a 4-instruction carry chain.
__device__ __forceinline__ u64 reduce128(u64 lo, u64 hi) {
    u32 t0, t1, m;
    asm("v_sub_co_u32 %0, vcc, %3, %5\n\t"          // t = lo - hi_hi
        "v_subbrev_co_u32 %1, vcc, 0, %4, vcc\n\t"
        GL_MASK_OF_VCC("%2") "\n\t"          // borrow -> - (2^32-1)
        "v_sub_co_u32 %0, vcc, %0, %2\n\t"
        "v_subbrev_co_u32 %1, vcc, 0, %1, vcc"
        : "=&v"(t0), "=&v"(t1), "=&v"(m)
        : "v"((u32)lo), "v"((u32)(lo >> 32)), "v"((u32)(hi >> 32)), GL_ONES_OPERAND
        : "vcc");
    const u64 t = ((u64)t1 << 32) | t0;
    u64 r;
    u32 c;
    asm("v_mad_u64_u32 %0, vcc, %2, -1, %3\n\t"     // r = hi_lo * (2^32-1) + t, carry in vcc
        GL_MASK_OF_VCC("%1")
        : "=&v"(r), "=&v"(c)
        : "v"((u32)hi), "v"(t), GL_ONES_OPERAND
        : "vcc");
    return r + (u64)c;   // carry -> + (2^32-1); cannot wrap (see host version)
}
// a * (2^32 - 1) for a 32-bit a, as two 32-bit ops (the compiler would pick an 8-cycle v_mad_u64_u32)
__device__ __forceinline__ u64 mul_eps(u32 a) {
    u32 r0, r1;
    asm("v_sub_co_u32 %0, vcc, 0, %2\n\t"
        "v_subbrev_co_u32 %1, vcc, 0, %2, vcc"
        : "=&v"(r0), "=&v"(r1) : "v"(a) : "vcc");
    return ((u64)r1 << 32) | r0;
}
// lo + hi*2^64 with hi < 2^32:  hi*(2^32-1) + lo as one v_mad_u64_u32, then the carry fold
__device__ __forceinline__ u64 reduce96(u64 lo, u32 hi) {
    u64 r;
    u32 c;
    asm("v_mad_u64_u32 %0, vcc, %2, -1, %3\n\t"
        GL_MASK_OF_VCC("%1")
        : "=&v"(r), "=&v"(c)
        : "v"(hi), "v"(lo), GL_ONES_OPERAND
        : "vcc");
    return r + (u64)c;   // wrapped r < hi*(2^32-1) <= (2^32-1)^2, so adding 2^32-1 cannot wrap
}
#else
// host versions (plan building, table generation)
inline u64 add(u64 a, u64 b) {
    u64 s = a + b;
    u64 c = s < a ? EPS : 0;
    u64 t = s + c;
    t += (t < c) ? EPS : 0;
    return t;
}
inline u64 sub(u64 a, u64 b) {
    u64 d = a - b;
    u64 c = a < b ? EPS : 0;
    u64 t = d - c;
    t -= (d < c) ? EPS : 0;
    return t;
}
inline u64 neg(u64 a) { return sub(0, a); }
inline u64 reduce128(u64 lo, u64 hi) {
    u64 hh = hi >> 32, hl = hi & EPS;
    u64 t0 = lo - hh;
    if (lo < hh) t0 -= EPS;
    u64 t1 = (hl << 32) - hl;
    u64 t2 = t0 + t1;
    if (t2 < t1) t2 += EPS;
    return t2;
}
inline u64 mul_eps(u32 a) { return ((u64)a << 32) - a; }
inline u64 reduce96(u64 lo, u32 hi) {
    u64 t1 = ((u64)hi << 32) - hi;
    u64 t2 = lo + t1;
    if (t2 < t1) t2 += EPS;
    return t2;
}
#endif

GL_HD void mul64wide(u64 a, u64 b, u64 &lo, u64 &hi) {
#if defined(__HIP_DEVICE_COMPILE__)
    u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    u64 p00 = (u64)a0 * b0;
    u64 p01 = (u64)a0 * b1 + (p00 >> 32);   // v_mad_u64_u32, no overflow
    u64 p10 = (u64)a1 * b0 + (u32)p01;      // no overflow
    u64 p11 = (u64)a1 * b1 + (p01 >> 32);   // no overflow: (2^32-1)^2 + 2^32-1 < 2^64
    lo = (p10 << 32) | (u32)p00;
    hi = p11 + (p10 >> 32);
#else
    unsigned __int128 m = (unsigned __int128)a * b;
    lo = (u64)m; hi = (u64)(m >> 64);
#endif
}
GL_HD u64 mul(u64 a, u64 b) {
    u64 lo, hi;
    mul64wide(a, b, lo, hi);
    return reduce128(lo, hi);
}
#if defined(__HIP_DEVICE_COMPILE__)
// ---- lazy products for throughput builds: the rare folds of a group of independent products behind ONE wave-uniform branch ----
// reduce128 begins with t = lo - hi_hi, whose borrow (lo < hi_hi < 2^32) random data sees about once in 2^32 products, and spends
// three of the product's 22 vector instructions on folding it. The lazy form returns t and the wave mask of the lanes whose fold
// is still due (a scalar register pair); mul_group ORs the masks of N independent products, tests the union once (GL_ANY_RARE: a
// scalar compare and branch, no vector issue slot; `unlikely` keeps the folds out of line) and finishes the products: 19 vector
// instructions each. A taken or not-taken scalar branch still costs its wave tens of cycles of latency, which only other resident
// waves cover — hence one branch per group, and hence the plain forms wherever a launch is small (qpgpu_tp_min_threads).
// Measured (profiles/r03_rare_fold.txt): Poseidon permutation 2.61 -> 2.81 G/s in registers at full occupancy. The same treatment
// of the modular sum and difference (5 + 5 instead of 7 + 8 instructions per NTT butterfly) was built and measured too: the
// radix-32 register block alone ran 1.13-1.26x faster, the NTT pass kernels did not move (+-3 %: four waves per SIMD do not hide
// the branches, and the passes are not bound by butterfly issue alone), so sums and differences keep their plain forms.
#define GL_ANY_RARE(mask) __builtin_expect((mask) != 0, 0)
// a product in two halves: t = lo - hi_hi of the 128-bit product (borrow -> rare) and the word still to be folded in
struct LazyProd { u32 t0, t1, hl; };
__device__ __forceinline__ LazyProd mul_lazy(u64 a, u64 b, u64 &rare) {
    u64 lo, hi;
    mul64wide(a, b, lo, hi);
    LazyProd r;
    asm("v_sub_co_u32 %0, vcc, %3, %5\n\t"
        "v_subbrev_co_u32 %1, vcc, 0, %4, vcc\n\t"     // VOP2: the carry-in is implicit (no operand-read hazard)
        "s_mov_b64 %2, vcc"
        : "=&v"(r.t0), "=&v"(r.t1), "=s"(rare)
        : "v"((u32)lo), "v"((u32)(lo >> 32)), "v"((u32)(hi >> 32))
        : "vcc");
    r.hl = (u32)hi;
    return r;
}
__device__ __forceinline__ void mul_fold(LazyProd &r, u64 rare) {
    u32 m;
    asm volatile("s_nop 1\n\t"
                 "v_cndmask_b32 %2, 0, -1, %3\n\t"          // borrow -> - (2^32-1)
                 "v_sub_co_u32 %0, vcc, %0, %2\n\t"
                 "v_subbrev_co_u32 %1, vcc, 0, %1, vcc"
                 : "+v"(r.t0), "+v"(r.t1), "=&v"(m) : "s"(rare) : "vcc");
}
__device__ __forceinline__ u64 mul_finish(const LazyProd &p) {
    const u64 t = ((u64)p.t1 << 32) | p.t0;
    u64 r;
    u32 c;
    asm("v_mad_u64_u32 %0, vcc, %2, -1, %3\n\t"     // r = hi_lo * (2^32-1) + t, carry in vcc
        GL_MASK_OF_VCC("%1")
        : "=&v"(r), "=&v"(c)
        : "v"(p.hl), "v"(t), GL_ONES_OPERAND
        : "vcc");
    return r + (u64)c;
}
// r[i] = a[i] * b[i] for N independent products, one rare-fold branch for the group
template <int N>
__device__ __forceinline__ void mul_group(u64 (&r)[N], const u64 (&a)[N], const u64 (&b)[N]) {
    LazyProd p[N];
    u64 rare[N], any = 0;
#pragma unroll
    for (int i = 0; i < N; i++) { p[i] = mul_lazy(a[i], b[i], rare[i]); any |= rare[i]; }
    if (GL_ANY_RARE(any)) {
#pragma unroll
        for (int i = 0; i < N; i++) mul_fold(p[i], rare[i]);
    }
#pragma unroll
    for (int i = 0; i < N; i++) r[i] = mul_finish(p[i]);
}
#elif defined(__HIPCC__)
// host pass of a .hip unit: device templates that name mul_group must still parse (never called)
template <int N>
inline void mul_group(u64 (&r)[N], const u64 (&a)[N], const u64 (&b)[N]) { for (int i = 0; i < N; i++) r[i] = mul(a[i], b[i]); }
#endif

// ---- sums of products with one reduction at the end: acc (lo + hi 2^64 + top 2^128) += a * b ----
// The quotient kernels weight up to a few hundred constraints per point with powers of alpha: reducing every product costs a
// reduce128 and a modular addition (about 16 instructions); accumulating the 128-bit products costs a five-instruction carry chain.
struct Acc192 { u64 lo, hi; u32 top; };
GL_HD Acc192 acc_zero() { Acc192 a; a.lo = 0; a.hi = 0; a.top = 0; return a; }
GL_HD void acc_mul(Acc192 &acc, u64 a, u64 b) {
    u64 l, h;
    mul64wide(a, b, l, h);
#if defined(__HIP_DEVICE_COMPILE__)
    u32 l0 = (u32)acc.lo, l1 = (u32)(acc.lo >> 32), h0 = (u32)acc.hi, h1 = (u32)(acc.hi >> 32), t = acc.top;
    asm("v_add_co_u32 %0, vcc, %0, %5\n\t"
        "v_addc_co_u32 %1, vcc, %1, %6, vcc\n\t"
        "v_addc_co_u32 %2, vcc, %2, %7, vcc\n\t"
        "v_addc_co_u32 %3, vcc, %3, %8, vcc\n\t"
        "v_addc_co_u32 %4, vcc, 0, %4, vcc"
        // early-clobber: the chain writes %0..%2 before it reads %6..%8, so no input may share a register with an accumulator limb
        : "+&v"(l0), "+&v"(l1), "+&v"(h0), "+&v"(h1), "+&v"(t)
        : "v"((u32)l), "v"((u32)(l >> 32)), "v"((u32)h), "v"((u32)(h >> 32))
        : "vcc");
    acc.lo = ((u64)l1 << 32) | l0; acc.hi = ((u64)h1 << 32) | h0; acc.top = t;
#else
    const u64 lo = acc.lo + l;
    const u64 c0 = lo < l ? 1 : 0;
    const u64 hi1 = acc.hi + h, c1 = hi1 < h ? 1 : 0;
    const u64 hi2 = hi1 + c0, c2 = hi2 < c0 ? 1 : 0;
    acc.lo = lo; acc.hi = hi2; acc.top += (u32)(c1 + c2);
#endif
}
// 2^128 = (2^64)^2 = (2^32 - 1)^2 = 2^64 - 2^33 + 1 = -2^32 (mod p): the carries out of the 128 bits are subtracted, shifted
GL_HD u64 acc_reduce(const Acc192 &acc) { return sub(reduce128(acc.lo, acc.hi), (u64)acc.top << 32); }

// a + b for a canonical b (< p): one carry fold is exact (a + b - 2^64 < b <= p - 1, so adding 2^32 - 1 cannot wrap)
GL_HD u64 add_canonical(u64 a, u64 b) {
    const u64 s = a + b;
    return s + (s < a ? EPS : 0);
}
GL_HD u64 sqr(u64 a) { return mul(a, a); }

// x * 2^S mod p for a compile-time S in [0, 192). 2^96 = -1, 2^192 = 1.
template <int S>
GL_HD u64 mul_pow2(u64 x) {
    static_assert(S >= 0 && S < 192, "shift out of range");
    if constexpr (S == 0) {
        return x;
    } else if constexpr (S >= 96) {
        return neg(mul_pow2<S - 96>(x));
    } else if constexpr (S < 32) {
        return reduce96(x << S, (u32)(x >> (64 - S)));
    } else if constexpr (S == 32) {
        return reduce96(x << 32, (u32)(x >> 32));
    } else if constexpr (S < 64) {
        return reduce128(x << S, x >> (64 - S));
    } else if constexpr (S == 64) {
        // x*W^2 = x0*(W-1) - x1   (W = 2^32, W^2 = W - 1, W^3 = -1)
        u32 x0 = (u32)x, x1 = (u32)(x >> 32);
        return sub(mul_eps(x0), (u64)x1);
    } else {
        // 64 < S < 96: (a0,a1,a2) = x << (S-64) as three 32-bit limbs; x*2^S = a0*(W-1) - (a1 + a2*W)
        constexpr int r = S - 64;
        u32 a0 = (u32)x << r;
        u64 top = x >> (32 - r);   // a1 + a2*W
        return sub(mul_eps(a0), top);
    }
}

GL_HD u64 pow(u64 b, u64 e) {
    u64 r = 1;
    while (e) { if (e & 1) r = mul(r, b); b = sqr(b); e >>= 1; }
    return canon(r);
}
GL_HD u64 sqr_n(u64 v, int n) { for (int i = 0; i < n; i++) v = sqr(v); return v; }
// a^(p-2) by an addition chain: p - 2 = (2^31 - 1) * 2^33 + (2^32 - 1); e_k = a^(2^k - 1). 64 squarings + 10 products
// (the binary method needs 63 + 63: the exponent has 63 one bits) — inversions sit on the critical path of the witness
// generators (EqualityGenerator, NonzeroTestGenerator, QuotientGeneratorExtension) and of the partial-product rows.
GL_HD u64 inv(u64 a) {
    const u64 e2 = mul(sqr(a), a);
    const u64 e4 = mul(sqr_n(e2, 2), e2);
    const u64 e8 = mul(sqr_n(e4, 4), e4);
    const u64 e16 = mul(sqr_n(e8, 8), e8);
    const u64 e24 = mul(sqr_n(e16, 8), e8);
    const u64 e28 = mul(sqr_n(e24, 4), e4);
    const u64 e30 = mul(sqr_n(e28, 2), e2);
    const u64 e31 = mul(sqr(e30), a);
    const u64 e32 = mul(sqr(e31), a);
    return canon(mul(sqr_n(e31, 33), e32));
}
GL_HD u64 root_of_unity(unsigned log_n) {
    u64 r = ROOT_2_32;
    for (unsigned i = log_n; i < 32; i++) r = sqr(r);
    return canon(r);
}


// ---- quadratic extension F[x]/(x^2 - 7): serialised [a, b] = a + b x ----
struct e2 { u64 a, b; };
GL_HD e2 e2_make(u64 a, u64 b) { e2 r; r.a = a; r.b = b; return r; }
GL_HD e2 e2_from(u64 a) { return e2_make(a, 0); }
GL_HD e2 e2_add(e2 x, e2 y) { return e2_make(add(x.a, y.a), add(x.b, y.b)); }
GL_HD e2 e2_sub(e2 x, e2 y) { return e2_make(sub(x.a, y.a), sub(x.b, y.b)); }
GL_HD u64 mul7(u64 x) { return sub(mul_pow2<3>(x), x); }
GL_HD e2 e2_mul(e2 x, e2 y) {
    // Karatsuba: 3 base products
    u64 t0 = mul(x.a, y.a), t1 = mul(x.b, y.b);
    u64 t2 = mul(add(x.a, x.b), add(y.a, y.b));
    return e2_make(add(t0, mul7(t1)), sub(sub(t2, t0), t1));
}
GL_HD e2 e2_scale(e2 x, u64 s) { return e2_make(mul(x.a, s), mul(x.b, s)); }
GL_HD e2 e2_canon(e2 x) { return e2_make(canon(x.a), canon(x.b)); }
GL_HD e2 e2_pow(e2 b, u64 e) {
    e2 r = e2_from(1);
    while (e) { if (e & 1) r = e2_mul(r, b); b = e2_mul(b, b); e >>= 1; }
    return e2_canon(r);
}
GL_HD e2 e2_inv(e2 x) {
    u64 nrm = sub(sqr(x.a), mul7(sqr(x.b)));
    u64 ni = inv(nrm);
    return e2_canon(e2_make(mul(x.a, ni), mul(neg(x.b), ni)));
}

}  // namespace gl
