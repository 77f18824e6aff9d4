// ntt_mfma_block_microbench.hip — the in-register radix-32 block of the NTT passes (ntt_kernel_impl.hpp: dif_regs<5>) as a constant
// 32 x 32 matrix on the matrix pipe, with the digit-GEMM machinery of poseidon_mfma.hpp (a measured lead for a later round, not
// library code). The block's twiddles are powers of two, so the vector-ALU form is shifts and modular sums (about 67 vector
// instructions per element); as a matrix of field constants it is 8 tiles x 8 K-steps of v_mfma_i32_32x32x32_i8 per wave (32
// inputs x 8 digits along K, 32 outputs x 8 limbs along M), plus digits (7), lane swaps and recombination (about 17) per element.
// The matrix is read off the library's own block (its images of the unit vectors), so both variants compute the same function:
// equal checksums are required.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -w -I qp-zk-circuits_amd/csrc tools/ntt_mfma_block_microbench.hip -o tools/scratch_bin/ntt_mfma_block_microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "ntt_kernel_impl.hpp"
#include "poseidon_mfma.hpp"

using gl::u32;
using gl::u64;
constexpr int TILES = 8, STEPS = 8, TABLE = TILES * STEPS * 1024;

__global__ void columns_kernel(u64 *m) {      // m[pos * 32 + j] = output pos of the block on the unit vector e_j
    const int j = threadIdx.x;
    u64 x[32];
#pragma unroll
    for (int i = 0; i < 32; i++) x[i] = i == j ? 1 : 0;
    dif_regs<5, false>(x);
#pragma unroll
    for (int i = 0; i < 32; i++) m[i * 32 + j] = gl::canon(x[i]);
}

struct Cinit { u32 c[8]; };
// eight signed base-256 digits of the representative of w mod p in the balanced range (as pmf::host::const_digits)
static bool balanced_digits(u64 w, signed char (&d)[8]) {
    w %= gl::P;
    __int128 rep = w <= 0x7F7F7F7F7F7F7F7Full ? (__int128)w : (__int128)w - (__int128)gl::P;
    for (int a = 0; a < 8; a++) {
        int dg = (int)(rep & 255);
        if (dg >= 128) dg -= 256;
        d[a] = (signed char)dg;
        rep = (rep - dg) / 256;
    }
    return rep == 0;
}

template <int MODE>
__global__ __launch_bounds__(256) void block_kernel(u64 *out, const uint4 *table, Cinit ci, int iters) {
    extern __shared__ uint4 lds4[];
    if (MODE == 1) {
        for (int i = threadIdx.x; i < TABLE / 16; i += 256) lds4[i] = table[i];
        __syncthreads();
    }
    const unsigned char *lds = (const unsigned char *)lds4;
    const u64 t = threadIdx.x + blockIdx.x * (u64)blockDim.x;
    u64 x[32];
#pragma unroll
    for (int i = 0; i < 32; i++) x[i] = (t * 0x9E3779B97F4A7C15ull + i * 0xD1B54A32D192ED03ull) % gl::P;
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) { dif_regs<5, false>(x); continue; }
#if defined(__HIP_DEVICE_COMPILE__)
        u32 Dlo[32], Dhi[32];
#pragma unroll
        for (int e = 0; e < 32; e++) { const u64 d = pmf::to_digits(x[e]); Dlo[e] = (u32)d; Dhi[e] = (u32)(d >> 32); }
        pmf::v4i B0[STEPS], B1[STEPS];
#pragma unroll
        for (int q = 0; q < STEPS; q++) {
            pmf::swap32(Dlo[4 * q], Dlo[4 * q + 2]); pmf::swap32(Dhi[4 * q], Dhi[4 * q + 2]);
            pmf::swap32(Dlo[4 * q + 1], Dlo[4 * q + 3]); pmf::swap32(Dhi[4 * q + 1], Dhi[4 * q + 3]);
            B0[q] = pmf::v4i{(int)Dlo[4 * q], (int)Dhi[4 * q], (int)Dlo[4 * q + 1], (int)Dhi[4 * q + 1]};
            B1[q] = pmf::v4i{(int)Dlo[4 * q + 2], (int)Dhi[4 * q + 2], (int)Dlo[4 * q + 3], (int)Dhi[4 * q + 3]};
        }
        const int lane = threadIdx.x & 63;
#pragma unroll
        for (int T = 0; T < TILES; T++) {
            pmf::v16i a0, a1;
#pragma unroll
            for (int i = 0; i < 16; i++) a0[i] = (int)ci.c[i & 7];
            a1 = a0;
            const pmf::v4i *ap = (const pmf::v4i *)(lds + (size_t)T * STEPS * 1024) + lane;
#pragma unroll
            for (int q = 0; q < STEPS; q++) {
                const pmf::v4i a = ap[q * 64];
                a0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, B0[q], a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, B1[q], a1, 0, 0, 0);
            }
            asm volatile("" : "+v"(a0), "+v"(a1));
            u32 Z[4][8];
#pragma unroll
            for (int i = 0; i < 16; i++) {
                u32 xx = (u32)a0[i], yy = (u32)a1[i];
                pmf::swap32(xx, yy);
                Z[i >> 3][i & 7] = xx; Z[2 + (i >> 3)][i & 7] = yy;
            }
#pragma unroll
            for (int o = 0; o < 4; o++) x[4 * T + o] = pmf::recombine(Z[o]);
        }
#endif
    }
    u64 acc = 0;
#pragma unroll
    for (int i = 0; i < 32; i++) acc ^= gl::canon(x[i]) * (2 * i + 1);
    out[t] = acc;
}

template <int MODE>
static void run(const char *name, int iters, const uint4 *table, Cinit ci) {
    u64 *out; const int blocks = 256 * 8, threads = 256; hipMalloc(&out, (size_t)blocks * threads * 8);
    const size_t shm = MODE == 1 ? TABLE : 0;
    if (shm) hipFuncSetAttribute((const void *)block_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((block_kernel<MODE>), dim3(blocks), dim3(threads), shm, 0, out, table, ci, 2);
    if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed\n", name); return; }
    float best = 1e9; u64 chk = 0;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((block_kernel<MODE>), dim3(blocks), dim3(threads), shm, 0, out, table, ci, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    hipMemcpy(&chk, out + 4321, 8, hipMemcpyDeviceToHost);
    printf("%-52s %8.3f ms  %8.2f G elements/s (one 5-level block each)  chk %016llx\n", name, best, (double)blocks * threads * 32 * iters / best / 1e6, (unsigned long long)chk);
    hipFree(out);
}

int main() {
    // the block's matrix, from the block itself
    u64 *dm; hipMalloc(&dm, 1024 * 8);
    hipLaunchKernelGGL(columns_kernel, dim3(1), dim3(32), 0, 0, dm);
    std::vector<u64> F(1024);
    if (hipMemcpy(F.data(), dm, 1024 * 8, hipMemcpyDeviceToHost) != hipSuccess) { printf("matrix read failed\n"); return 1; }
    std::vector<unsigned char> tab(TABLE, 0);
    typedef unsigned __int128 u128;
    for (int T = 0; T < TILES; T++)
        for (int q = 0; q < STEPS; q++)
            for (int o = 0; o < 4; o++)
                for (int k = 0; k < 32; k++) {
                    const int e = 4 * q + (k >> 3), b = k & 7;
                    u64 w = F[(4 * T + o) * 32 + e];
                    for (int s = 0; s < b; s++) w = (u64)((u128)w * 256 % gl::P);
                    signed char d[8];
                    if (!balanced_digits(w, d)) { printf("digits failed\n"); return 2; }
                    for (int limb = 0; limb < 8; limb++) {
                        const int ln = pmf::row_of(o, limb) + 32 * (k >> 4), j = k & 15;
                        tab[(size_t)(T * STEPS + q) * 1024 + ln * 16 + j] = (unsigned char)d[limb];
                    }
                }
    // accumulator start: 2^23 + byte l of (-BIAS), BIAS = sum_l 2^23 2^(8l)
    u64 bias = 0, pw = 1u << 23;
    for (int l = 0; l < 8; l++) { bias = (u64)(((u128)bias + pw) % gl::P); pw = (u64)((u128)pw * 256 % gl::P); }
    const u64 c = (gl::P - bias) % gl::P;
    Cinit ci;
    for (int l = 0; l < 8; l++) ci.c[l] = (1u << 23) + (u32)((c >> (8 * l)) & 0xFF);
    uint4 *table; hipMalloc(&table, TABLE); hipMemcpy(table, tab.data(), TABLE, hipMemcpyHostToDevice);
    run<0>("radix-32 block, u64 modular (library)", 200, table, ci);
    run<1>("radix-32 block, constant matrix on MFMA (64 KB LDS)", 200, table, ci);
    return 0;
}
