// ntt_limb_microbench.hip — is a carry-free limb representation cheaper than 64-bit modular arithmetic for the in-register
// radix-32 block of the NTT passes ("implement the limb form and measure it instead of estimating")?
//
// The block (ntt_kernel_impl.hpp: dif_regs<5>) is five butterfly levels on 32 elements per thread whose twiddles are powers of
// two (plonky2's 64th root of unity is 8): 2^(6j), 2^(12j), 2^(24j), 2^(48j), 1. In Z[X]/(X^4 + 1) with X = 2^24 (2^96 = -1 mod
// p) a field element is four signed 32-bit limbs a0 + a1 X + a2 X^2 + a3 X^3, an addition is four independent v_add_u32, a
// multiplication by 2^24q is a limb rotation with sign flips (free), and only shifts by r = s mod 24 in {6, 12, 18} cost
// instructions (split every limb at bit 24 - r, carry the high part into the next limb). Limbs stay below 2^31 through all five
// levels, so nothing is renormalised inside the block — but the block's inputs and outputs are 64-bit field elements (HBM
// layout, the real 64 x 64 twiddle product at every round boundary, the LDS exchange), so every block pays a conversion in
// and a reduction out.
//
// Variants timed (each thread transforms 32 elements per iteration, values chained so nothing is hoisted):
//   u64      the library's dif_regs<5> on 64-bit modular arithmetic
//   limbs    conversion in, the five levels on 4 x 24-bit limbs, reduction out  (bit-exact with u64: same checksum)
//   limbs*   the five levels alone on limb inputs that stay limbs (no conversions: the unreachable best case)
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -w -I qp-zk-circuits_amd/csrc tools/ntt_limb_microbench.hip -o tools/scratch_bin/ntt_limb_microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include "ntt_kernel_impl.hpp"

typedef int i32;
struct L4 { i32 a[4]; };

__device__ __forceinline__ L4 l4_from(u64 x) {
    L4 r;
    r.a[0] = (i32)((u32)x & 0xFFFFFFu);
    r.a[1] = (i32)((u32)(x >> 24) & 0xFFFFFFu);
    r.a[2] = (i32)(u32)(x >> 48);
    r.a[3] = 0;
    return r;
}
// a0 + a1 2^24 + a2 2^48 + a3 2^72 with signed limbs below 2^31 in magnitude -> loose field element
__device__ __forceinline__ u64 l4_to(const L4 &v) {
    // signed 64-bit pieces: lo = a0 + a1 2^24 (|lo| < 2^56), hi = a2 + a3 2^24 (|hi| < 2^56), value = lo + 2^48 hi
    const long long lo = (long long)v.a[0] + ((long long)v.a[1] << 24);
    const long long hi = (long long)v.a[2] + ((long long)v.a[3] << 24);
    // as field elements: a negative piece is p - |piece| (|piece| < p)
    const u64 flo = lo < 0 ? gl::P - (u64)(-lo) : (u64)lo;
    const u64 fhi = hi < 0 ? gl::P - (u64)(-hi) : (u64)hi;
    return gl::add(flo, gl::mul_pow2<48>(fhi));
}
__device__ __forceinline__ L4 l4_add(const L4 &x, const L4 &y) { L4 r; for (int i = 0; i < 4; i++) r.a[i] = x.a[i] + y.a[i]; return r; }
__device__ __forceinline__ L4 l4_sub(const L4 &x, const L4 &y) { L4 r; for (int i = 0; i < 4; i++) r.a[i] = x.a[i] - y.a[i]; return r; }
// x * 2^S, 0 <= S < 192 a compile-time constant
template <int S>
__device__ __forceinline__ L4 l4_shift(const L4 &x) {
    if constexpr (S >= 96) {
        L4 t = l4_shift<S - 96>(x);
        for (int i = 0; i < 4; i++) t.a[i] = -t.a[i];
        return t;
    } else {
        constexpr int q = S / 24, r = S % 24;
        L4 b = x;
        if constexpr (r > 0) {
            i32 hi[4];
#pragma unroll
            for (int i = 0; i < 4; i++) { hi[i] = x.a[i] >> (24 - r); b.a[i] = (x.a[i] & ((1 << (24 - r)) - 1)) << r; }
            b.a[0] -= hi[3]; b.a[1] += hi[0]; b.a[2] += hi[1]; b.a[3] += hi[2];
        }
        L4 c;
#pragma unroll
        for (int i = 0; i < 4; i++) { const int d = i + q; if (d < 4) c.a[d] = b.a[i]; else c.a[d - 4] = -b.a[i]; }
        return c;
    }
}
template <int LEN, int J>
__device__ __forceinline__ void l4_column(L4 (&x)[32]) {      // butterfly j of every block of this level
    constexpr int half = LEN >> 1, step = 192 / LEN;
    if constexpr (J < half) {
#pragma unroll
        for (int b = 0; b < 32; b += LEN) {
            const L4 u = x[b + J], v = x[b + J + half];
            x[b + J] = l4_add(u, v);
            x[b + J + half] = l4_shift<J * step>(l4_sub(u, v));
        }
        l4_column<LEN, J + 1>(x);
    }
}
template <int LEN>
__device__ __forceinline__ void l4_level(L4 (&x)[32]) {
    if constexpr (LEN >= 2) {
        l4_column<LEN, 0>(x);
        l4_level<(LEN >> 1)>(x);
    }
}

template <int MODE> __global__ __launch_bounds__(256) void k(u64 *out, int iters) {
    const u64 t = threadIdx.x + blockIdx.x * (u64)blockDim.x;
    u64 x[32];
#pragma unroll
    for (int i = 0; i < 32; i++) x[i] = (t * 0x9E3779B97F4A7C15ull + i * 0xD1B54A32D192ED03ull) % gl::P;
    if (MODE == 2) {
        L4 y[32];
#pragma unroll
        for (int i = 0; i < 32; i++) y[i] = l4_from(x[i]);
        for (int it = 0; it < iters; it++) {
            l4_level<32>(y);
#pragma unroll
            for (int i = 0; i < 32; i++) for (int k2 = 0; k2 < 4; k2++) y[i].a[k2] &= 0xFFFFFF;      // keep the limbs bounded between iterations (4 ands per element)
        }
#pragma unroll
        for (int i = 0; i < 32; i++) x[i] = l4_to(y[i]);
    } else {
        for (int it = 0; it < iters; it++) {
            if (MODE == 0) dif_regs<5, false>(x);
            else {
                L4 y[32];
#pragma unroll
                for (int i = 0; i < 32; i++) y[i] = l4_from(gl::canon(x[i]));
                l4_level<32>(y);
#pragma unroll
                for (int i = 0; i < 32; i++) x[i] = l4_to(y[i]);
            }
        }
    }
    u64 acc = 0;
#pragma unroll
    for (int i = 0; i < 32; i++) acc ^= gl::canon(x[i]) * (2 * i + 1);
    out[t] = acc;
}
template <int MODE> void run(const char *name, int iters) {
    u64 *out; const int blocks = 256 * 8, threads = 256; hipMalloc(&out, (size_t)blocks * threads * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(threads), 0, 0, out, 2);
    float best = 1e9; u64 chk = 0;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(threads), 0, 0, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    hipMemcpy(&chk, out + 4321, 8, hipMemcpyDeviceToHost);
    printf("%-44s %8.3f ms  %8.2f G elements/s (one 5-level block each)  chk %016llx\n", name, best, (double)blocks * threads * 32 * iters / best / 1e6, (unsigned long long)chk);
    hipFree(out);
}
int main() {
    run<0>("radix-32 block, u64 modular (library)", 200);
    run<1>("radix-32 block, 4x24-bit limbs + conversions", 200);
    run<2>("radix-32 block, limbs only (no conversions)", 200);
    return 0;
}
