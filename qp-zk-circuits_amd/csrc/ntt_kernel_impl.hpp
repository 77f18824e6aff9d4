// ntt_kernel_impl.hpp — Goldilocks radix-2^k NTT pass kernel template for gfx950 (CDNA4); instantiated per
// (KA, KB) group in ntt_inst_*.hip so the groups compile in parallel.
//
// Replaces plonky2::field::fft::{fft, ifft, coset_fft} as used by PolynomialBatch::from_values /
// from_coeffs inside `prove` (reference call site wormhole/prover/src/lib.rs:171-175; stage s2 of
// SURVEY.md §8a). Output values are the same canonical field elements the CPU transform produces.
//
// Structure (MI355X-first, not a translation of the CPU butterfly loop):
//   * An N-point transform is 1..3 "passes"; a pass does a 2^(KA+KB)-point DIF transform on every
//     lane of a tile of T lanes, entirely on chip: round A (2^KA points per thread, in registers),
//     one padded LDS exchange, round B (2^KB points per thread, in registers).
//   * The register rounds are twiddle-free: plonky2's primitive 64th root of unity is 8, so every
//     twiddle inside a <=64-point block is a power of two and a multiplication is a shift plus the
//     2^64 = 2^32-1 / 2^96 = -1 folding. Real 64x64 multiplications happen once per element per
//     round boundary only.
//   * HBM access is tile-shaped: T adjacent lanes (T*8 B segments) on strided passes, full rows on
//     the contiguous pass; 64-bit loads/stores, no atomics, no inter-workgroup communication.
#pragma once
#include <hip/hip_runtime.h>
#include "gl64.hpp"
#include "ntt_pass.hpp"

using gl::u32;
using gl::u64;

namespace {

__device__ __forceinline__ u64 mul_pow2_dyn(u64 x, int s) {
    // s is a compile-time constant after unrolling; the chain folds to one arm.
    switch (s >> 5) {
        default:
        case 0:
            if (s == 0) return x;
            return gl::reduce96(x << s, (u32)(x >> (64 - s)));
        case 1:
            if (s == 32) return gl::reduce96(x << 32, (u32)(x >> 32));
            return gl::reduce128(x << s, x >> (64 - s));
        case 2: {
            if (s == 64) {
                u32 x0 = (u32)x, x1 = (u32)(x >> 32);
                return gl::sub(gl::mul_eps(x0), (u64)x1);
            }
            const int r = s - 64;
            u32 a0 = (u32)x << r;
            u64 top = x >> (32 - r);
            return gl::sub(gl::mul_eps(a0), top);
        }
    }
}

__device__ __forceinline__ constexpr int brev(int x, int bits) {
    int r = 0;
    for (int i = 0; i < bits; i++) r |= ((x >> i) & 1) << (bits - 1 - i);
    return r;
}

// In-register decimation-in-frequency transform of 2^K points. On exit x[j] holds X[bitrev_K(j)].
// One butterfly level per template instance (LEN = 2^K, 2^(K-1), .., 2), so every shift amount is a constant.
template <int K, bool INV, int LEN>
__device__ __forceinline__ void dif_level(u64 (&x)[1 << K]) {
    if constexpr (LEN >= 2) {
        constexpr int N = 1 << K, half = LEN >> 1;
        constexpr int step = 192 / LEN;  // w_LEN = 2^(192/LEN), LEN <= 64
#pragma unroll
        for (int b = 0; b < N; b += LEN) {
#pragma unroll
            for (int j = 0; j < half; j++) {
                u64 u = x[b + j], v = x[b + j + half];
                const int s = (INV ? ((LEN - j) % LEN) : j) * step;  // in [0,192)
                x[b + j] = gl::add(u, v);
                if (s >= 96) x[b + j + half] = mul_pow2_dyn(gl::sub(v, u), s - 96);
                else x[b + j + half] = mul_pow2_dyn(gl::sub(u, v), s);
            }
        }
        dif_level<K, INV, half>(x);
    }
}

template <int K, bool INV>
__device__ __forceinline__ void dif_regs(u64 (&x)[1 << K]) {
#ifndef NTT_EXPERIMENT_NO_BUTTERFLIES    // timing experiments only: what a pass costs without its register transforms
    dif_level<K, INV, (1 << K)>(x);
#endif
}

// The same transform when only the first V = 2^LV of the 2^K inputs are non-zero (the first pass of a zero-padded LDE: a
// coefficient vector of length n in a transform of length 8 n). With k = k1 + (2^K / V) k2: X[k] = DFT_V(x[v] w^(k1 v))[k2], so
// the first K - LV butterfly levels collapse into one shift-multiply per element and only LV levels of butterflies remain.
template <int K, bool INV, int LV>
__device__ __forceinline__ void dif_sparse(u64 (&x)[1 << K]) {
    constexpr int N = 1 << K, V = 1 << LV, G = N / V, UNIT = 192 / N;
    u64 in[V];
#pragma unroll
    for (int v = 0; v < V; v++) in[v] = x[v];
#pragma unroll
    for (int jh = 0; jh < G; jh++) {
        const int k1 = brev(jh, K - LV);
        if constexpr (LV == 1) {
            int e = (UNIT * k1) % 192;
            if (INV) e = (192 - e) % 192;
            const u64 t = mul_pow2_dyn(in[1], e >= 96 ? e - 96 : e);       // a factor 2^96 = -1 swaps the two outputs
            x[2 * jh] = e >= 96 ? gl::sub(in[0], t) : gl::add(in[0], t);
            x[2 * jh + 1] = e >= 96 ? gl::add(in[0], t) : gl::sub(in[0], t);
        } else {
            u64 t[V];
#pragma unroll
            for (int v = 0; v < V; v++) {
                int e = (UNIT * k1 * v) % 192;
                if (INV) e = (192 - e) % 192;
                t[v] = e >= 96 ? gl::neg(mul_pow2_dyn(in[v], e - 96)) : mul_pow2_dyn(in[v], e);
            }
            dif_regs<LV, INV>(t);
#pragma unroll
            for (int v = 0; v < V; v++) x[jh * V + v] = t[v];
        }
    }
}

// Tile geometry shared by the two rounds of a pass.
struct PassGeom {
    int T, tid, RP;
    u64 lane0;
    const u64 *in;
    u64 *out;
};

// Round A of a pass: loads 2^KA points per thread, coset scale, register transform, inner twiddles. With KB == 0 it also
// finishes the pass. On return x holds the values to exchange and wr_base the LDS word of x[0] (x[q] goes NB + 1 words
// further per q); TO_LDS stores them as whole 8-byte words right away. Returns whether this thread took part.
template <int KA, int KB, bool INV, bool TO_LDS, int SPARSE>
__device__ __forceinline__ bool ntt_round_a(const NttPassArgs &a, const PassGeom &g, u64 *lds, u64 (&x)[1 << KA], int &wr_base) {
    constexpr int NA = 1 << KA, NB = 1 << KB;
    const int T = g.T, tid = g.tid;
    int m, l;
    if (a.load_lane_fast) { l = tid & (T - 1); m = tid >> a.log_t; }
    else { m = tid & (NB - 1); l = tid >> KB; }
    const bool active = (tid < (NB << a.log_t));
    const u64 lane = g.lane0 + l;
    const bool lane_ok = lane < a.lanes_total;
    wr_base = l * g.RP + m;
    if (!active) return false;
    const u64 *in = g.in;
    // lane -> (outer row, inner lane) for strided passes: lane = row * M + mm
    // offsets inside one column fit 32 bits (N <= 2^30): 32-bit index arithmetic, one 64-bit add per access
    const u32 row = (u32)(lane >> a.log_m), mm = (u32)lane & ((1u << a.log_m) - 1);
    const u32 ips = (u32)a.in_p_stride;
    const u32 base = row * (u32)a.in_row_stride + mm * (u32)a.in_l_stride + (u32)m * ips;
    const u32 istep = (u32)NB * ips;
#pragma unroll
    for (int i = 0; i < NA; i++) {
        const u32 p = (u32)(i * NB + m);
        u64 v = 0;
        if (lane_ok && p < a.p_valid) v = in[base + (u32)i * istep];
        x[i] = v;
    }
    if (a.in_scale_a) {  // coset: x[p, lane] *= A[p] * B[mm]
        const u64 sb = a.in_scale_b[mm];
#pragma unroll
        for (int i = 0; i < NA; i++) {
            const u32 p = (u32)(i * NB + m);
            if (p < a.p_valid) x[i] = gl::mul(x[i], gl::mul(a.in_scale_a[p], sb));
        }
    }
    // SPARSE > 0 (the LDE kernels): inputs i >= 2^SPARSE are zero for every thread of the pass (p_valid <= 2^SPARSE * NB)
    if constexpr (SPARSE > 0) dif_sparse<KA, INV, SPARSE>(x);
    else dif_regs<KA, INV>(x);
    if constexpr (KB > 0) {
#pragma unroll
        for (int j = 0; j < NA; j++) {
            const int ka = brev(j, KA);
            if (ka != 0 && m != 0) x[j] = gl::mul(x[j], a.tw_inner[(u32)(m * ka)]);
            if constexpr (TO_LDS) lds[wr_base + j * (NB + 1)] = x[j];
        }
    } else {
        // single-round pass: finish here
        u64 *out = g.out;
#pragma unroll
        for (int j = 0; j < NA; j++) {
            const u32 k = (u32)brev(j, KA);
            u64 v = x[j];
            if (a.tw_lo) {
                const u32 e = mm * k;
                if (e) v = gl::mul(v, gl::mul(a.tw_hi[e >> a.tw_lo_bits], a.tw_lo[e & ((1u << a.tw_lo_bits) - 1)]));
            }
            if (a.has_out_scale) v = gl::mul(v, a.out_scale);
            v = gl::canon(v);
            const u32 pos = a.out_bitrev ? (u32)j : k;
            if (lane_ok) out[row * (u32)a.out_row_stride + mm * (u32)a.out_l_stride + pos * (u32)a.out_p_stride] = v;
        }
    }
    return true;
}

// Round B after the exchange: register transform of the 2^KB points in y, inter-pass twiddle, output scale, store.
template <int KA, int KB, bool INV>
__device__ __forceinline__ void ntt_round_b(const NttPassArgs &a, const PassGeom &g, u64 (&y)[1 << KB], int j, int l) {
    constexpr int NB = 1 << KB;
    u64 *out = g.out;
    const u64 lane = g.lane0 + l;
    const bool lane_ok = lane < a.lanes_total;
    dif_regs<KB, INV>(y);
    const u32 row = (u32)(lane >> a.log_m), mm = (u32)lane & ((1u << a.log_m) - 1);
    const u32 ops = (u32)a.out_p_stride;
    const u32 obase = row * (u32)a.out_row_stride + mm * (u32)a.out_l_stride;
    const u32 ka = (u32)brev(j, KA);
    if (a.tw_lo && a.tw_mode == 1) {
        // inter-pass twiddle w^(mm*k), k = ka + (r << KA) for output r = brev(jb): w^(mm*ka) * (w^(mm << KA))^r. Two
        // table lookups per thread and a running product instead of two scattered table reads per element.
        const u32 lo_mask = (1u << a.tw_lo_bits) - 1;
        const u32 e0 = mm * ka, es = mm << KA;
        u64 t = gl::mul(a.tw_hi[e0 >> a.tw_lo_bits], a.tw_lo[e0 & lo_mask]);
        if (a.tw_scale) t = gl::mul(t, a.tw_scale);   // an inverse transform's 1/N rides on the first factor: the last pass multiplies nothing
        const u64 step = gl::mul(a.tw_hi[es >> a.tw_lo_bits], a.tw_lo[es & lo_mask]);
#pragma unroll
        for (int r = 0; r < NB; r++) {
            const int jb = brev(r, KB);
            y[jb] = gl::mul(y[jb], t);
            if (r + 1 < NB) t = gl::mul(t, step);
        }
    }
    if (a.tw_lo && a.tw_mode == 3) {
        // the full table: one load and one product per element instead of two products (the running one and the element's)
        const u64 *T = a.tw_full + mm;
#pragma unroll
        for (int jb = 0; jb < NB; jb++) {
            const u32 k = ka + ((u32)brev(jb, KB) << KA);
            y[jb] = gl::mul(y[jb], T[(u64)k << a.log_m]);
        }
    }
#pragma unroll
    for (int jb = 0; jb < NB; jb++) {
        const u32 k = ka + ((u32)brev(jb, KB) << KA);
        u64 v = y[jb];
        if (a.tw_lo && a.tw_mode == 0) {
            const u32 e = mm * k;
            if (e) v = gl::mul(v, gl::mul(a.tw_hi[e >> a.tw_lo_bits], a.tw_lo[e & ((1u << a.tw_lo_bits) - 1)]));
        }
        if (a.has_out_scale) v = gl::mul(v, a.out_scale);
        if (!a.out_loose) v = gl::canon(v);           // an intermediate pass may hand over any representative
        const u32 pos = a.out_bitrev ? (u32)(j * NB + jb) : k;
        if (lane_ok) out[obase + pos * ops] = v;
    }
}

// One pass of one tile. Element s = j * NB + m of lane l crosses the rounds through LDS word l * RP + s + (s >> KB).
// Whole-word mode: one 8-byte exchange, one barrier. SPLIT: the low and the high 32-bit halves go through the same 4-byte
// slots one after the other — half the LDS per workgroup, so twice as many wavefronts share a CU, for two more barriers.
template <int KA, int KB, bool INV, bool SPLIT, int SPARSE = 0>
__device__ __forceinline__ void ntt_pass_body(const NttPassArgs &a) {
    constexpr int NA = 1 << KA, NB = 1 << KB;
    extern __shared__ __align__(16) u64 lds[];
    PassGeom g;
    g.T = 1 << a.log_t;
    g.tid = threadIdx.x;
    g.RP = (int)a.row_pitch;   // R + NA + a pad chosen for the exchange width and the tile shape (ntt_pass_row_pitch)
    g.lane0 = (u64)blockIdx.x << a.log_t;          // first lane of this tile (global lane index)
    const u64 col = blockIdx.y;
    g.in = a.in + col * a.in_col_stride + (u64)blockIdx.z * a.in_proof_stride;
    g.out = a.out + col * a.out_col_stride + (u64)blockIdx.z * a.out_proof_stride;

    if constexpr (!SPLIT) {
        {
            u64 x[NA];
            int wr_base;
            ntt_round_a<KA, KB, INV, true, SPARSE>(a, g, lds, x, wr_base);
        }
        if constexpr (KB == 0) return;
        __syncthreads();
        int j, l;   // round B thread mapping
        if (a.store_lane_fast) { l = g.tid & (g.T - 1); j = g.tid >> a.log_t; }
        else { j = g.tid & (NA - 1); l = g.tid >> KA; }
        const bool active_b = (g.tid < (NA << a.log_t));
        const int rd_base = l * g.RP + j * (NB + 1);
        if (active_b) {
            u64 y[NB];
#pragma unroll
            for (int i = 0; i < NB; i++) y[i] = lds[rd_base + i];
            ntt_round_b<KA, KB, INV>(a, g, y, j, l);
        }
    } else {
        static_assert(KB > 0, "split exchange needs two rounds");
        u32 *lds32 = reinterpret_cast<u32 *>(lds);
        u64 x[NA];
        int wr_base;
        const bool active_a = ntt_round_a<KA, KB, INV, false, SPARSE>(a, g, lds, x, wr_base);
        int j, l;   // round B thread mapping
        if (a.store_lane_fast) { l = g.tid & (g.T - 1); j = g.tid >> a.log_t; }
        else { j = g.tid & (NA - 1); l = g.tid >> KA; }
        const bool active_b = (g.tid < (NA << a.log_t));
        const int rd_base = l * g.RP + j * (NB + 1);
        u32 ylo[NB];
        u64 y[NB];
        if (active_a) {
#pragma unroll
            for (int q = 0; q < NA; q++) lds32[wr_base + q * (NB + 1)] = (u32)x[q];
        }
        __syncthreads();
        if (active_b) {
#pragma unroll
            for (int i = 0; i < NB; i++) ylo[i] = lds32[rd_base + i];
        }
        __syncthreads();
        if (active_a) {
#pragma unroll
            for (int q = 0; q < NA; q++) lds32[wr_base + q * (NB + 1)] = (u32)(x[q] >> 32);
        }
        __syncthreads();
        if (active_b) {
#pragma unroll
            for (int i = 0; i < NB; i++) y[i] = ((u64)lds32[rd_base + i] << 32) | ylo[i];
            ntt_round_b<KA, KB, INV>(a, g, y, j, l);
        }
    }
}

// ROWS only names the launch kind (contiguous-row pass vs strided / single pass) so profilers list them separately.
template <int KA, int KB, bool INV, bool ROWS>
__global__ void __launch_bounds__(512) ntt_pass_kernel(const NttPassArgs a) { ntt_pass_body<KA, KB, INV, false>(a); }

// Split-exchange variant for the 2^9- and 2^10-point passes: half the LDS per workgroup lets four 256-thread workgroups share
// a CU, so the register budget is pinned to four wavefronts per SIMD (128 VGPRs) to match.
template <int KA, int KB, bool INV, bool ROWS>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) ntt_pass_split_kernel(const NttPassArgs a) {
    ntt_pass_body<KA, KB, INV, true>(a);
}

template <int KA, int KB>
constexpr bool has_split_variant() { return KB > 0 && KA + KB >= 9; }

// First pass of a zero-padded LDE (forward, strided): only the first 2 (KA = 4: a 2^13-coefficient column in a 2^16-point
// transform) or 4 (KA = 5) inputs of every round-A thread are non-zero, so round A is dif_sparse.
template <int KA, int KB>
constexpr int lde_sparse_lv() { return (KB > 0 && KA == 4) ? 1 : (KB > 0 && KA == 5) ? 2 : 0; }
template <int KA, int KB>
__global__ void __launch_bounds__(512) ntt_pass_lde_kernel(const NttPassArgs a) {
    ntt_pass_body<KA, KB, false, false, lde_sparse_lv<KA, KB>()>(a);
}
template <int KA, int KB>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) ntt_pass_lde_split_kernel(const NttPassArgs a) {
    ntt_pass_body<KA, KB, false, true, lde_sparse_lv<KA, KB>()>(a);
}

template <int KA, int KB>
hipError_t launch_dir(const NttPassArgs &a, dim3 grid, dim3 block, size_t lds, hipStream_t st) {
    const bool rows = KB > 0 && a.log_m == 0 && a.load_lane_fast == 0 && a.in_p_stride == 1 && a.in_row_stride > 1 && a.in_col_stride != 0;
    if constexpr (lde_sparse_lv<KA, KB>() > 0) {
        if (a.sparse_lv == lde_sparse_lv<KA, KB>() && !a.inverse) {
            if constexpr (has_split_variant<KA, KB>()) {
                if (a.split_lds) { hipLaunchKernelGGL((ntt_pass_lde_split_kernel<KA, KB>), grid, block, lds, st, a); return hipGetLastError(); }
            }
            hipLaunchKernelGGL((ntt_pass_lde_kernel<KA, KB>), grid, block, lds, st, a);
            return hipGetLastError();
        }
    }
    if constexpr (has_split_variant<KA, KB>()) {
        if (a.split_lds) {
            if (rows) {
                if (a.inverse) hipLaunchKernelGGL((ntt_pass_split_kernel<KA, KB, true, true>), grid, block, lds, st, a);
                else hipLaunchKernelGGL((ntt_pass_split_kernel<KA, KB, false, true>), grid, block, lds, st, a);
            } else {
                if (a.inverse) hipLaunchKernelGGL((ntt_pass_split_kernel<KA, KB, true, false>), grid, block, lds, st, a);
                else hipLaunchKernelGGL((ntt_pass_split_kernel<KA, KB, false, false>), grid, block, lds, st, a);
            }
            return hipGetLastError();
        }
    }
    if constexpr (KB > 0) {
        if (rows) {
            if (a.inverse) hipLaunchKernelGGL((ntt_pass_kernel<KA, KB, true, true>), grid, block, lds, st, a);
            else hipLaunchKernelGGL((ntt_pass_kernel<KA, KB, false, true>), grid, block, lds, st, a);
            return hipGetLastError();
        }
    }
    if (a.inverse) hipLaunchKernelGGL((ntt_pass_kernel<KA, KB, true, false>), grid, block, lds, st, a);
    else hipLaunchKernelGGL((ntt_pass_kernel<KA, KB, false, false>), grid, block, lds, st, a);
    return hipGetLastError();
}

template <int KA, int KB>
hipError_t set_lds_attr() {
    const void *fns[4] = {(const void *)ntt_pass_kernel<KA, KB, false, false>, (const void *)ntt_pass_kernel<KA, KB, true, false>,
                          (const void *)ntt_pass_kernel<KA, KB, false, true>, (const void *)ntt_pass_kernel<KA, KB, true, true>};
    for (const void *f : fns) {
        hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
    }
    if constexpr (lde_sparse_lv<KA, KB>() > 0) {
        hipError_t e = hipFuncSetAttribute((const void *)ntt_pass_lde_kernel<KA, KB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        if constexpr (has_split_variant<KA, KB>()) {
            e = hipFuncSetAttribute((const void *)ntt_pass_lde_split_kernel<KA, KB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
        }
    }
    if constexpr (has_split_variant<KA, KB>()) {
        const void *sp[4] = {(const void *)ntt_pass_split_kernel<KA, KB, false, false>, (const void *)ntt_pass_split_kernel<KA, KB, true, false>,
                             (const void *)ntt_pass_split_kernel<KA, KB, false, true>, (const void *)ntt_pass_split_kernel<KA, KB, true, true>};
        for (const void *f : sp) {
            hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
        }
    }
    return hipSuccess;
}

}  // namespace

#define NTT_DEFINE_CASE(A, B) \
    hipError_t ntt_launch_##A##_##B(const NttPassArgs &a, dim3 grid, dim3 block, size_t lds, hipStream_t st) { return launch_dir<A, B>(a, grid, block, lds, st); } \
    hipError_t ntt_attr_##A##_##B() { return set_lds_attr<A, B>(); }
