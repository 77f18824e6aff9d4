// poseidon_microbench.hip — what bounds the Poseidon permutation on gfx950 (evidence for DESIGN.md section 4: "at the floor").
//   * the MDS layer alone, in registers: the library's form (two 32-bit halves in signed 64-bit arithmetic, v_lshl_add_u64)
//     against a three-limb form (22/21/21 bits) whose whole convolution fits 32-bit adds — fewer "expensive" instructions
//     on paper, measured slower; and the 64-bit form with its first butterfly stage on the 32-bit halves (carry pairs
//     instead of zero-extended register pairs: 331 instead of 364 VALU instructions, 68 instead of 86 VGPRs, not faster);
//   * the permutation in registers, with the round constants behind a kernel argument, the __constant__ symbol, and two
//     laundered pointers (batched vs one-at-a-time scalar loads: no difference);
//   * a leaf-hash-shaped kernel (135 columns, 17 permutations per leaf), plain and with the next chunk's loads issued early.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -w -I qp-zk-circuits_amd/csrc tools/poseidon_microbench.hip -o tools/scratch_bin/poseidon_microbench
// Results of the round-2 run: profiles/r02_poseidon_microbench.txt
#include <hip/hip_runtime.h>
#include <cstdio>
#include "poseidon.hpp"
namespace poseidon {
// circulant part on one vector of limbs, modulo 2^32 (exact when the true outputs lie in [0, 2^32))
GL_HD void mds_circulant_u32(const u32 (&x)[WIDTH], u32 (&o)[WIDTH]) {
    u32 U1[3], Um[3], F[3], H[3];
#pragma unroll
    for (int b = 0; b < 3; b++) {
        const u32 e = x[b] + x[b + 6], f = x[b] - x[b + 6], g = x[b + 3] + x[b + 9], h = x[b + 3] - x[b + 9];
        U1[b] = e + g; Um[b] = e - g; F[b] = f; H[b] = h;
    }
    const u32 T = U1[0] + U1[1] + U1[2];
    const u32 A[3] = {T + U1[2], T + U1[0], T + U1[1]};
    const u32 B[3] = {(Um[2] << 3) - (Um[0] + (Um[1] << 1)),
                      0u - ((Um[0] << 3) + Um[1] + (Um[2] << 1)),
                      (Um[0] << 1) - ((Um[1] << 3) + Um[2])};
    const u32 f0 = F[0], f1 = F[1], f2 = F[2], h0 = H[0], h1 = H[1], h2 = H[2];
    const u32 R[3] = {((f0 << 1) + f1 + f2 + (h2 << 2)) - (h0 + (h1 << 4)),
                      ((f1 << 1) + h0 + f2) - ((f0 << 2) + h1 + (h2 << 4)),
                      ((f0 << 4) + h0 + h1 + (f2 << 1)) - ((f1 << 2) + h2)};
    const u32 I[3] = {(f0 + (h0 << 1) + (f1 << 4) + h1 + h2) - (f2 << 2),
                      (f1 + (h1 << 1) + (f2 << 4) + h2) - (f0 + (h0 << 2)),
                      ((h0 << 4) + f2 + (h2 << 1)) - (f0 + f1 + (h1 << 2))};
#pragma unroll
    for (int b = 0; b < 3; b++) {
        const u32 a16 = A[b] << 4, p = a16 + B[b], q = a16 - B[b];
        o[b] = p + R[b]; o[b + 3] = q + I[b]; o[b + 6] = p - R[b]; o[b + 9] = q - I[b];
    }
}
template <class RC> GL_HD void permute_as(u64 (&s)[WIDTH], RC rc) {
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

// first butterfly stage on the 32-bit halves themselves (33-bit results built from the carry / borrow), so no
// zero-extended register pairs have to be materialised for the inputs
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ i64 add33(u32 a, u32 b) {
    u32 lo, hi;
    asm("v_add_co_u32 %0, vcc, %2, %3\n\tv_addc_co_u32 %1, vcc, 0, 0, vcc" : "=&v"(lo), "=&v"(hi) : "v"(a), "v"(b) : "vcc");
    return (i64)(((u64)hi << 32) | lo);
}
#else
inline i64 add33(u32 a, u32 b) { return (i64)((u64)a + b); }
#endif
GL_HD i64 sub33(u32 a, u32 b) { return (i64)a - (i64)b; }
GL_HD void mds_circulant_half_v2(const u32 (&x)[WIDTH], i64 (&o)[WIDTH]) {
    i64 U1[3], Um[3], F[3], H[3];
#pragma unroll
    for (int b = 0; b < 3; b++) {
        const i64 e = add33(x[b], x[b + 6]), f = sub33(x[b], x[b + 6]), g = add33(x[b + 3], x[b + 9]), h = sub33(x[b + 3], x[b + 9]);
        U1[b] = e + g; Um[b] = e - g; F[b] = f; H[b] = h;
    }
    const i64 T = U1[0] + U1[1] + U1[2];
    const i64 A[3] = {T + U1[2], T + U1[0], T + U1[1]};
    const i64 B[3] = {shl(Um[2], 3) - Um[0] - shl(Um[1], 1), -(shl(Um[0], 3) + Um[1] + shl(Um[2], 1)), shl(Um[0], 1) - shl(Um[1], 3) - Um[2]};
    const i64 f0 = F[0], f1 = F[1], f2 = F[2], h0 = H[0], h1 = H[1], h2 = H[2];
    const i64 R[3] = {shl(f0, 1) - h0 + f1 - shl(h1, 4) + f2 + shl(h2, 2), -shl(f0, 2) + h0 + shl(f1, 1) - h1 + f2 - shl(h2, 4), shl(f0, 4) + h0 - shl(f1, 2) + h1 + shl(f2, 1) - h2};
    const i64 I[3] = {f0 + shl(h0, 1) + shl(f1, 4) + h1 - shl(f2, 2) + h2, -f0 - shl(h0, 2) + f1 + shl(h1, 1) + shl(f2, 4) + h2, -f0 + shl(h0, 4) - f1 - shl(h1, 2) + f2 + shl(h2, 1)};
#pragma unroll
    for (int b = 0; b < 3; b++) {
        const i64 a16 = shl(A[b], 4), p = a16 + B[b], q = a16 - B[b];
        o[b] = p + R[b]; o[b + 3] = q + I[b]; o[b + 6] = p - R[b]; o[b + 9] = q - I[b];
    }
}
GL_HD void mds_layer_v2(u64 (&s)[WIDTH]) {
    u32 lo[WIDTH], hi[WIDTH]; i64 ol[WIDTH], oh[WIDTH];
#pragma unroll
    for (int i = 0; i < WIDTH; i++) { lo[i] = (u32)s[i]; hi[i] = (u32)(s[i] >> 32); }
    mds_circulant_half_v2(lo, ol);
    mds_circulant_half_v2(hi, oh);
    ol[0] += (i64)((u64)lo[0] << 3); oh[0] += (i64)((u64)hi[0] << 3);
#pragma unroll
    for (int r = 0; r < WIDTH; r++) {
        const u64 al = (u64)ol[r], ah = (u64)oh[r];
        const u64 low = al + (ah << 32);
        const u32 top = (u32)(ah >> 32) + (low < al ? 1u : 0u);
        s[r] = gl::reduce96(low, top);
    }
}
GL_HD void mds_layer3(u64 (&s)[WIDTH]) {
    u32 l0[WIDTH], l1[WIDTH], l2[WIDTH], o0[WIDTH], o1[WIDTH], o2[WIDTH];
#pragma unroll
    for (int i = 0; i < WIDTH; i++) {
        l0[i] = (u32)s[i] & 0x3FFFFFu; l1[i] = (u32)(s[i] >> 22) & 0x1FFFFFu; l2[i] = (u32)(s[i] >> 43);
    }
    mds_circulant_u32(l0, o0); mds_circulant_u32(l1, o1); mds_circulant_u32(l2, o2);
    o0[0] += l0[0] << 3; o1[0] += l1[0] << 3; o2[0] += l2[0] << 3;
#pragma unroll
    for (int r = 0; r < WIDTH; r++) {
        const u64 a = (u64)o0[r] + ((u64)o1[r] << 22);
        const u64 low = a + ((u64)o2[r] << 43);
        const u32 top = (o2[r] >> 21) + (low < a ? 1u : 0u);
        s[r] = gl::reduce96(low, top);
    }
}
}
using gl::u64;
template <int V> __device__ __forceinline__ void mds(u64 (&s)[12]) { if (V == 0) poseidon::mds_layer(s); else if (V == 1) poseidon::mds_layer3(s); else poseidon::mds_layer_v2(s); }
template <int V> __device__ void perm(u64 (&s)[12], const u64 *rc) {
    using namespace poseidon;
    int r = 0;
    for (int k = 0; k < HALF_FULL; k++, r++) {
#pragma unroll
        for (int i = 0; i < WIDTH; i++) s[i] = sbox7(gl::add_canonical(s[i], rc[r * WIDTH + i]));
        mds<V>(s);
    }
#pragma unroll
    for (int i = 1; i < WIDTH; i++) s[i] = gl::add_canonical(s[i], rc[r * WIDTH + i]);
    for (int k = 0; k < PARTIAL; k++, r++) { s[0] = sbox7(gl::add_canonical(s[0], rc[r * WIDTH])); mds<V>(s); }
    for (int k = 0; k < HALF_FULL; k++, r++) {
#pragma unroll
        for (int i = 0; i < WIDTH; i++) s[i] = sbox7(gl::add_canonical(s[i], rc[r * WIDTH + i]));
        mds<V>(s);
    }
#pragma unroll
    for (int i = 0; i < WIDTH; i++) s[i] = gl::canon(s[i]);
}
__constant__ u64 c_rc[360];
typedef const __attribute__((address_space(4))) u64 *crc_t;
template <int RCM> __global__ __launch_bounds__(256) void kc(u64 *out, const u64 *rc_arg, int iters) {
    u64 s[12];
    const u64 t = threadIdx.x + blockIdx.x * (u64)blockDim.x;
    for (int i = 0; i < 12; i++) s[i] = t * 0x9E3779B97F4A7C15ull + i;
    for (int it = 0; it < iters; it++) {
        if (RCM == 1) perm<0>(s, c_rc);
        if (RCM == 2) { const u64 *rc = c_rc; asm volatile("" : "+s"(rc)); perm<0>(s, rc); }
        if (RCM == 3) { crc_t rc = (crc_t)c_rc; asm volatile("" : "+s"(rc)); poseidon::permute_as(s, rc); }
    }
    u64 x = 0; for (int i = 0; i < 12; i++) x ^= s[i];
    out[t] = x;
}
template <int RCM> void runc(const char *name, int iters, const u64 *rc) {
    u64 *out; int blocks = 256 * 16, threads = 256; hipMalloc(&out, (size_t)blocks * threads * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((kc<RCM>), dim3(blocks), dim3(threads), 0, 0, out, rc, 2);
    float best = 1e9; u64 chk = 0;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((kc<RCM>), dim3(blocks), dim3(threads), 0, 0, out, rc, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    hipMemcpy(&chk, out + 12345, 8, hipMemcpyDeviceToHost);
    printf("%-24s %8.3f ms  %8.3f G/s  chk %016llx\n", name, best, (double)blocks * threads * iters / best / 1e6, (unsigned long long)chk);
    hipFree(out);
}
template <int V, int MODE> __global__ __launch_bounds__(256) void k(u64 *out, const u64 *rc, int iters) {
    u64 s[12];
    const u64 t = threadIdx.x + blockIdx.x * (u64)blockDim.x;
    for (int i = 0; i < 12; i++) s[i] = t * 0x9E3779B97F4A7C15ull + i;
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) mds<V>(s);
        else if (MODE == 1) perm<V>(s, rc);
        else if (MODE == 2) poseidon::permute(s, rc);              // the library's permutation: partial rounds in the spectral domain (round 3)
        else poseidon::permute_layerwise(s, rc);                   // the library's permutation of rounds 1-2: one MDS layer per round
    }
    u64 x = 0; for (int i = 0; i < 12; i++) x ^= s[i];
    out[t] = x;
}
template <int W, int LIB = 0> __global__ __launch_bounds__(256) void leafk(const u64 *src, u64 stride, u64 n, u64 *dig, const u64 *rc) {
    const u64 j = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (j >= n) return;
    u64 s[12];
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = 0;
    for (int c = 0; c < W; c += 8) {
#pragma unroll
        for (int i = 0; i < 8; i++) if (c + i < W) s[i] = src[(u64)(c + i) * stride + j];
        if (LIB) poseidon::permute(s, rc); else perm<0>(s, rc);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) dig[j * 4 + i] = s[i];
}
// same, with the next chunk's loads issued before the permutation of the current one
template <int W> __global__ __launch_bounds__(256) void leafk_pf(const u64 *src, u64 stride, u64 n, u64 *dig, const u64 *rc) {
    const u64 j = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (j >= n) return;
    u64 s[12], nx[8];
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) nx[i] = src[(u64)i * stride + j];
    for (int c = 0; c < W; c += 8) {
#pragma unroll
        for (int i = 0; i < 8; i++) if (c + i < W) s[i] = nx[i];
        if (c + 8 < W) {
#pragma unroll
            for (int i = 0; i < 8; i++) if (c + 8 + i < W) nx[i] = src[(u64)(c + 8 + i) * stride + j];
        }
        perm<0>(s, rc);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) dig[j * 4 + i] = s[i];
}
template <int W, int PF> void run_leaf(const char *name, u64 n, const u64 *rc) {
    u64 *src, *dig; hipMalloc(&src, n * W * 8); hipMalloc(&dig, n * 32); hipMemset(src, 0x5a, n * W * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 4; rep++) {
        hipEventRecord(e0);
        if (PF == 1) hipLaunchKernelGGL((leafk_pf<W>), dim3((n + 255) / 256), dim3(256), 0, 0, src, n, n, dig, rc);
        else if (PF == 2) hipLaunchKernelGGL((leafk<W, 1>), dim3((n + 255) / 256), dim3(256), 0, 0, src, n, n, dig, rc);
        else hipLaunchKernelGGL((leafk<W, 0>), dim3((n + 255) / 256), dim3(256), 0, 0, src, n, n, dig, rc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    u64 chk; hipMemcpy(&chk, dig + 4 * 777, 8, hipMemcpyDeviceToHost);
    printf("%-24s n=%llu %8.3f ms  %8.3f Gperm/s chk %016llx\n", name, (unsigned long long)n, best, (double)n * ((W + 7) / 8) / best / 1e6, (unsigned long long)chk);
    hipFree(src); hipFree(dig);
}
template <int V, int MODE> void run(const char *name, int iters, const u64 *rc) {
    u64 *out; int blocks = 256 * 16, threads = 256; hipMalloc(&out, (size_t)blocks * threads * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<V, MODE>), dim3(blocks), dim3(threads), 0, 0, out, rc, 2);
    float best = 1e9; u64 chk = 0;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<V, MODE>), dim3(blocks), dim3(threads), 0, 0, out, rc, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    hipMemcpy(&chk, out + 12345, 8, hipMemcpyDeviceToHost);
    printf("%-24s %8.3f ms  %8.3f G/s  chk %016llx\n", name, best, (double)blocks * threads * iters / best / 1e6, (unsigned long long)chk);
    hipFree(out);
}
int main() {
    u64 h[360]; for (int i = 0; i < 360; i++) h[i] = (0x123456789ABCDEFull * (i + 1)) % 0xFFFFFFFF00000001ull;
    u64 *rc; hipMalloc(&rc, sizeof h); hipMemcpy(rc, h, sizeof h, hipMemcpyHostToDevice);
    run<0, 0>("mds i64", 2000, rc); run<1, 0>("mds 3x u32", 2000, rc);
    run<2, 0>("mds i64, 33-bit stage 1", 2000, rc);
    run<0, 1>("permute i64", 64, rc); run<2, 1>("permute, 33-bit stage 1", 64, rc); run<1, 1>("permute 3x u32", 64, rc);
    run<0, 3>("permute, lib layerwise", 64, rc); run<0, 2>("permute, lib spectral", 64, rc);
    hipMemcpyToSymbol(HIP_SYMBOL(c_rc), h, sizeof h);
    runc<1>("permute, __constant__", 64, rc); runc<2>("permute, laundered flat", 64, rc); runc<3>("permute, laundered as4", 64, rc);
    run_leaf<135, 0>("leaf W=135", 1ull << 21, rc); run_leaf<135, 1>("leaf W=135 prefetch", 1ull << 21, rc);
    run_leaf<135, 2>("leaf W=135 spectral", 1ull << 21, rc); run_leaf<8, 2>("nodes W=8 spectral", 1ull << 20, rc); run_leaf<135, 2>("leaf W=135 spectral", 1ull << 16, rc);
    run_leaf<135, 0>("leaf W=135", 1ull << 16, rc); run_leaf<135, 1>("leaf W=135 prefetch", 1ull << 16, rc);
    run_leaf<8, 0>("nodes W=8", 1ull << 20, rc);
    return 0;
}
