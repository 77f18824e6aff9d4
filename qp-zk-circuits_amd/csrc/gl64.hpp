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

// loose add: inputs any u64, at least one of them < 2^64 - 2^32 (true for canonical or reduce output
// minus the top sliver); result loose. Two carry folds keep it exact for all inputs.
GL_HD u64 add(u64 a, u64 b) {
    u64 s = a + b;
    u64 c = s < a ? EPS : 0;
    u64 t = s + c;
    t += (t < c) ? EPS : 0;
    return t;
}
GL_HD u64 sub(u64 a, u64 b) {
    u64 d = a - b;
    u64 c = a < b ? EPS : 0;
    u64 t = d - c;
    t -= (d < c) ? EPS : 0;
    return t;
}
GL_HD u64 neg(u64 a) { return sub(0, a); }

// (hi:lo) 128-bit -> loose u64
GL_HD u64 reduce128(u64 lo, u64 hi) {
    u64 hh = hi >> 32, hl = hi & EPS;
    u64 t0 = lo - hh;
    if (lo < hh) t0 -= EPS;
    u64 t1 = (hl << 32) - hl;  // hl * (2^32-1)
    u64 t2 = t0 + t1;
    if (t2 < t1) t2 += EPS;
    return t2;
}
// lo + hi*2^64 with hi < 2^32
GL_HD u64 reduce96(u64 lo, u32 hi) {
    u64 t1 = ((u64)hi << 32) - hi;
    u64 t2 = lo + t1;
    if (t2 < t1) t2 += EPS;
    return t2;
}

GL_HD void mul64wide(u64 a, u64 b, u64 &lo, u64 &hi) {
#if defined(__HIP_DEVICE_COMPILE__)
    u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    u64 p00 = (u64)a0 * b0;
    u64 p01 = (u64)a0 * b1;
    u64 p10 = (u64)a1 * b0;
    u64 p11 = (u64)a1 * b1;
    u64 mid = p01 + (p00 >> 32);          // no overflow: < 2^64
    u64 mid2 = p10 + (u32)mid;            // no overflow
    lo = (mid2 << 32) | (u32)p00;
    hi = p11 + (mid >> 32) + (mid2 >> 32);
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
        // x * 2^64 = x * (2^32 - 1)
        u64 y = reduce96(x << 32, (u32)(x >> 32));
        return sub(y, x);
    } else {
        // 64 < S < 96: first x*2^(S-64) (< 2^96), then times 2^64 = 2^32 - 1
        u64 y = reduce96(x << (S - 64), (u32)(x >> (128 - S)));
        u64 z = reduce96(y << 32, (u32)(y >> 32));
        return sub(z, y);
    }
}

GL_HD u64 pow(u64 b, u64 e) {
    u64 r = 1;
    while (e) { if (e & 1) r = mul(r, b); b = sqr(b); e >>= 1; }
    return canon(r);
}
GL_HD u64 inv(u64 a) { return pow(a, P - 2); }
GL_HD u64 root_of_unity(unsigned log_n) {
    u64 r = ROOT_2_32;
    for (unsigned i = log_n; i < 32; i++) r = sqr(r);
    return canon(r);
}

}  // namespace gl
