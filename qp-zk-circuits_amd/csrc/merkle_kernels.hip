// merkle_kernels.hip — Poseidon leaf hashing and Merkle reduction for gfx950.
//
// Replaces plonky2::hash::merkle_tree::MerkleTree::new and hash::hashing::{hash_n_to_hash_no_pad,
// hash_or_noop, two_to_one} (stage s3 of SURVEY.md §8a; reached from PolynomialBatch::from_coeffs inside
// prove, reference call site wormhole/prover/src/lib.rs:171-175).
//
// The thread-per-hash kernels live in merkle_hash_impl.hpp and are built twice: here (latency build) and in
// merkle_kernels_tp.hip (throughput build, taken by launches of qpgpu_tp_min_threads() threads or more). Leaf sponges and tree
// levels of that size under the Poseidon hasher go to a third build, merkle_kernels_mx.hip: partial rounds on the matrix pipe.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "merkle.hpp"
#include "poseidon.hpp"
#include "prover_kernels.hpp"

using gl::u32;
using gl::u64;

__constant__ u64 c_poseidon_rc[poseidon::ROUNDS * poseidon::WIDTH];

// throughput build (merkle_kernels_tp.hip)
hipError_t merkle_tp_upload_constants(const u64 *rc360);
hipError_t merkle_tp_leaves(const MerkleLeafArgs &a, u64 total, const HasherDev &h, hipStream_t st);
hipError_t merkle_tp_rows(const u64 *rows, u64 n_leaves, u32 width, u64 *digests, u32 batch, u64 ps_rows, u64 ps_digests, const HasherDev &h, hipStream_t st);
hipError_t merkle_tp_nodes(const u64 *in, u64 *out, u64 n_out, u32 batch, u64 ps, const HasherDev &h, hipStream_t st);
hipError_t merkle_tp_pow(const PowArgs &a, dim3 g, const HasherDev &h, hipStream_t st);
// matrix-pipe build (merkle_kernels_mx.hip)
hipError_t merkle_mx_upload_constants(const u64 *rc360);
hipError_t merkle_mx_leaves(const MerkleLeafArgs &a, u64 total, const HasherDev &h, hipStream_t st);
hipError_t merkle_mx_nodes(const u64 *in, u64 *out, u64 n_out, u32 batch, u64 ps, const HasherDev &h, hipStream_t st);
hipError_t merkle_mx_pow(const PowArgs &a, const HasherDev &h, hipStream_t st);
// the matrix build serves plonky2's Poseidon and Poseidon2 with qp-poseidon-core's parameters
static bool mx_serves(const HasherDev &h) { return h.kind != hasher::POSEIDON2 || h.qp; }

// Launches of at least this many threads (four resident waves per SIMD on 256 CUs) take the throughput build of a hashing kernel,
// smaller ones the latency build. QPGPU_TP_MIN_THREADS overrides (0: always, a huge value: never).
uint64_t qpgpu_tp_min_threads() {
    static const uint64_t v = [] { const char *e = getenv("QPGPU_TP_MIN_THREADS"); return e && *e ? strtoull(e, nullptr, 10) : (uint64_t)1 << 18; }();
    return v;
}

// QPGPU_MX=0 keeps large launches on the throughput build (A/B runs, and the way back should the matrix form misbehave)
static bool mx_enabled() {
    static const bool v = [] { const char *e = getenv("QPGPU_MX"); return !(e && *e == '0'); }();
    return v;
}
// below this many independent hashes a level is latency-bound and the lane-cooperative form wins. The cooperative form
// does about 3x the work, so the crossover depends on how many proofs share the GPU: QPGPU_COOP_MAX overrides it.
static u64 coop_max_init() {
    const char *e = getenv("QPGPU_COOP_MAX");
    if (e && *e) { const long long v = atoll(e); if (v >= 0) return (u64)v; }
    return 16384;
}
static const u64 COOP_MAX = coop_max_init();
// Lockstep batches of QPGPU_TPUT_BATCH trees or more (default 8) are throughput work: the device is shared by several such batches
// and what counts is the work a level costs, not how soon it ends. They skip the lane-cooperative kernels (about 3x the work per
// hash) and the fused tree top, and run every level on the large-launch builds (thread per hash; matrix-pipe build where it serves
// the hasher). Measured with six workers x 32 proofs: + 1 % on 2^13- and 2^12-row circuits (profiles/r03_poseidon_mfma.txt
// item 11). Single proofs and small batches keep the latency-oriented routing.
static u32 tput_batch() {
    static const u32 v = [] { const char *e = getenv("QPGPU_TPUT_BATCH"); return e && *e ? (u32)strtoul(e, nullptr, 10) : 8u; }();
    return v;
}
static inline bool tput(u32 batch) { return batch >= tput_batch(); }
static inline u64 coop_max_for(u32 batch) { return tput(batch) ? 0 : COOP_MAX; }
static inline u64 tp_min_for(u32 batch) { return tput(batch) ? 0 : qpgpu_tp_min_threads(); }

bool merkle_mx_in_use() { return mx_enabled(); }
hipError_t merkle_upload_constants(const u64 *rc360) {
    hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(c_poseidon_rc), rc360, sizeof(u64) * poseidon::ROUNDS * poseidon::WIDTH);
    if (e == hipSuccess) e = merkle_tp_upload_constants(rc360);
    if (e != hipSuccess || !mx_enabled()) return e;       // QPGPU_MX=0: no matrix build, no table, no device self-test
    return merkle_mx_upload_constants(rc360);
}

#include "merkle_hash_impl.hpp"

namespace {

// ---- lane-cooperative permutation: one state spread over 16 lanes (element g in lane g, 12 used), four states
// per wave. The S-box layer runs on all elements at once and the MDS layer gathers the other eleven elements with
// wave shuffles (ds_bpermute), so one permutation has ~1/5 of the single-thread latency. Total work is ~3x higher
// (partial rounds keep 11 lanes idle), so it is used only where a tree level is latency-bound: few nodes.
__device__ __forceinline__ u64 coop_permute(u64 s, const int g, const int lane_base) {
    constexpr u32 C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    int src[12];
#pragma unroll
    for (int i = 0; i < 12; i++) { int e = g + i; e -= e >= 12 ? 12 : 0; src[i] = lane_base + (g < 12 ? e : i); }
#pragma unroll 1
    for (int r = 0; r < poseidon::ROUNDS; r++) {
        const u64 rc = g < 12 ? c_poseidon_rc[r * 12 + g] : 0;
        s = gl::add(s, rc);
        const bool full = r < poseidon::HALF_FULL || r >= poseidon::HALF_FULL + poseidon::PARTIAL;
        const u64 sb = poseidon::sbox7(s);
        s = (full || g == 0) ? sb : s;
        const u32 lo = (u32)s, hi = (u32)(s >> 32);
        u64 al = 0, ah = 0;
#pragma unroll
        for (int i = 0; i < 12; i++) {
            const u32 l = (u32)__shfl((int)lo, src[i], 64), h = (u32)__shfl((int)hi, src[i], 64);
            al += (u64)l * C[i];
            ah += (u64)h * C[i];
        }
        if (g == 0) { al += (u64)lo * 8u; ah += (u64)hi * 8u; }
        const u64 low = al + (ah << 32);
        const u32 top = (u32)(ah >> 32) + (low < al ? 1u : 0u);
        s = gl::reduce96(low, top);
    }
    return gl::canon(s);
}

// one 16-lane group per node: out[i] = two_to_one(in[2i], in[2i+1])
__global__ void __launch_bounds__(256) node_coop_kernel(const u64 *in, u64 *out, u64 n_out, u32 batch, u64 ps) {
    const u64 ggrp = (blockIdx.x * (u64)blockDim.x + threadIdx.x) >> 4;
    const int lane = threadIdx.x & 63, g = lane & 15, base = lane & 48;
    const bool live = ggrp < n_out * batch;
    const u64 pr = live ? ggrp >> ilog2_64(n_out) : 0, grp = ggrp & (n_out - 1);
    in += pr * ps; out += pr * ps;
    u64 s = (live && g < 8) ? in[grp * 8 + g] : 0;
    s = coop_permute(s, g, base);
    if (live && g < 4) out[grp * 4 + g] = s;
}

// The top of a tree in one launch: workgroup b owns the subtree under cap entry b (m <= MAXM digests of the input level),
// keeps the current level in LDS and walks up to its root; every level is also written to the digest array (Merkle paths read
// it). A level with more nodes than the workgroup has 16-lane groups runs one node per thread (full work efficiency, one
// permutation deep); smaller levels run lane-cooperative permutations (16 lanes per node, about a fifth of the latency). Replaces up to ten launches whose
// levels are too small to fill the chip.
template <int MAXM>
__global__ void __launch_bounds__(MAXM >= 512 ? 512 : 256) tree_top_kernel(u64 *levels, u64 cnt, u32 m, u64 ps) {
    __shared__ u64 buf[2][MAXM * 4];
    const u32 t = threadIdx.x, b = blockIdx.x, T = blockDim.x;
    levels += (u64)blockIdx.y * ps;
    const int lane = t & 63, g = lane & 15, base = lane & 48;
    const u32 grp = t >> 4;
    for (u32 i = t; i < m * 4; i += T) buf[0][i] = levels[(u64)b * m * 4 + i];
    __syncthreads();
    u64 *out = levels + cnt * 4;          // next level in the digest array
    u64 level_cnt = cnt / 2;
    int cur = 0;
    for (u32 nodes = m / 2; nodes >= 1; nodes >>= 1) {
        if (nodes > T / 16) {                         // more nodes than 16-lane groups in the workgroup
            for (u32 i = t; i < nodes; i += T) {      // one node per thread
                u64 st[12];
#pragma unroll
                for (int k = 0; k < 8; k++) st[k] = buf[cur][i * 8 + k];
                st[8] = st[9] = st[10] = st[11] = 0;
                poseidon::permute(st, c_poseidon_rc);
#pragma unroll
                for (int k = 0; k < 4; k++) { buf[cur ^ 1][i * 4 + k] = st[k]; out[((u64)b * nodes + i) * 4 + k] = st[k]; }
            }
        } else if (((t >> 6) << 2) < nodes) {         // waves that hold a live 16-lane group
            const bool live = grp < nodes;
            u64 s = (live && g < 8) ? buf[cur][grp * 8 + g] : 0;
            s = coop_permute(s, g, base);
            if (live && g < 4) { buf[cur ^ 1][grp * 4 + g] = s; out[((u64)b * nodes + grp) * 4 + g] = s; }
        }
        __syncthreads();
        cur ^= 1;
        out += level_cnt * 4; level_cnt >>= 1;
    }
}

// one 16-lane group per row-major leaf
__global__ void __launch_bounds__(256) leaf_rows_coop_kernel(const u64 *rows, u64 n_leaves, u32 width, u64 *digests, u32 batch, u64 ps_rows, u64 ps_digests) {
    const u64 ggrp = (blockIdx.x * (u64)blockDim.x + threadIdx.x) >> 4;
    const int lane = threadIdx.x & 63, g = lane & 15, base = lane & 48;
    const bool live = ggrp < n_leaves * batch;
    const u64 pr = live ? ggrp >> ilog2_64(n_leaves) : 0, grp = ggrp & (n_leaves - 1);
    rows += pr * ps_rows; digests += pr * ps_digests;
    const u64 *row = rows + (live ? grp : 0) * width;
    if (width <= 4) {
        if (live && g < 4) digests[grp * 4 + g] = g < (int)width ? gl::canon(row[g]) : 0;
        return;
    }
    u64 s = 0;
    for (u32 c = 0; c < width; c += 8) {
        if (g < 8 && c + g < width) s = row[c + g];
        s = coop_permute(s, g, base);
    }
    if (live && g < 4) digests[grp * 4 + g] = s;
}

// one 16-lane group per column-major leaf (small LDEs)
__global__ void __launch_bounds__(256) leaf_cols_coop_kernel(MerkleLeafArgs a) {
    const u64 ggrp = (blockIdx.x * (u64)blockDim.x + threadIdx.x) >> 4;
    const int lane = threadIdx.x & 63, g = lane & 15, base = lane & 48;
    const bool live = ggrp < a.n_leaves * a.batch;
    const u64 pr = live ? ggrp >> ilog2_64(a.n_leaves) : 0, j = ggrp & (a.n_leaves - 1);
    a.src0 += pr * a.ps_src0; a.src1 += pr * a.ps_src1; a.digests += pr * a.ps_digests;
    const u32 W = a.ncols0 + a.ncols1;
    auto elem = [&](u32 c) -> u64 {
        return c < a.ncols0 ? a.src0[(u64)c * a.stride0 + j] : a.src1[(u64)(c - a.ncols0) * a.stride1 + j];
    };
    if (W <= 4) {
        if (live && g < 4) a.digests[j * 4 + g] = g < (int)W ? gl::canon(elem(g)) : 0;
        return;
    }
    u64 s = 0;
    for (u32 c = 0; c < W; c += 8) {
        if (g < 8 && c + g < W) s = elem(c + g);
        s = coop_permute(s, g, base);
    }
    if (live && g < 4) a.digests[j * 4 + g] = s;
}

// Poseidon2Hash::hash_no_pad of the qp fork (the application hash inside the Wormhole circuits; reference call sites
// wormhole/circuit/src/unspendable_account.rs:87-88, nullifier.rs:119-120, block_header/header.rs:140): preimage i = `len`
// elements at in + i * len, padded `|| 1 || 0*` to a multiple of the rate 8 (wormhole/circuit/tests/heap_zeroization.rs:133-160),
// every block ADDED into the rate part of the state, 4 outputs. One thread per preimage.
template <bool QP>   // QP: p2 is qp-poseidon-core's set (multiplication-free external layers); otherwise a caller's block
__global__ void __launch_bounds__(256) p2_pad10_sponge_kernel(const u64 *in, u64 len, u64 count, u64 *out, const poseidon2::Params *p2) {
    const u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (i >= count) return;
    const u64 *src = in + i * len;
    u64 s[12];
#pragma unroll
    for (int k = 0; k < 12; k++) s[k] = 0;
    for (u64 c = 0; c <= len; c += 8) {          // the block that holds the terminator is the last one
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const u64 idx = c + k;
            const u64 v = idx < len ? gl::canon(src[idx]) : (idx == len ? 1 : 0);
            s[k] = gl::add_canonical(s[k], v);
        }
        if constexpr (QP) poseidon2::permute_qp(s, *p2); else poseidon2::permute(s, *p2);
    }
#pragma unroll
    for (int k = 0; k < 4; k++) out[i * 4 + k] = s[k];
}

template <class Perm>
__global__ void permute_kernel(u64 *states, u64 n, const poseidon2::Params *p2) {
    const u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (i >= n) return;
    u64 s[12];
#pragma unroll
    for (int k = 0; k < 12; k++) s[k] = gl::canon(states[i * 12 + k]);
    Perm::permute(s, p2);
#pragma unroll
    for (int k = 0; k < 12; k++) states[i * 12 + k] = s[k];
}

}  // namespace

hipError_t pk_pow(const PowArgs &a, const HasherDev &h, hipStream_t st) {
    if (a.count == 0 || a.batch == 0) return hipSuccess;
    const u64 chunks = (a.count + 255) / 256;
    if (chunks * a.batch > 0x7FFFFFFFull) return hipErrorInvalidValue;
    dim3 g((unsigned)(chunks * a.batch));
    if (chunks * a.batch * 256 < tp_min_for(a.batch)) return hash_launch_pow(a, g, h, st);
    return (mx_serves(h) && mx_enabled()) ? merkle_mx_pow(a, h, st) : merkle_tp_pow(a, g, h, st);
}

hipError_t merkle_leaf_hash(const MerkleLeafArgs &a0, const HasherDev &h, hipStream_t st) {
    MerkleLeafArgs a = a0;
    if (a.batch == 0) a.batch = 1;
    if (a.n_leaves == 0) return hipSuccess;
    if (a.n_leaves & (a.n_leaves - 1)) return hipErrorInvalidValue;
    const u64 total = a.n_leaves * a.batch;
    const bool p2 = h.kind == hasher::POSEIDON2;   // the lane-cooperative kernels exist for Poseidon only
    if (!p2 && total <= coop_max_for(a.batch) / 2) {
        dim3 block(256), grid((unsigned)((total * 16 + 255) / 256));
        hipLaunchKernelGGL(leaf_cols_coop_kernel, grid, block, 0, st, a);
        return hipGetLastError();
    }
    if (total < tp_min_for(a.batch)) return hash_launch_leaves(a, total, h, st);
    return (mx_serves(h) && a.ncols0 + a.ncols1 > 4 && mx_enabled()) ? merkle_mx_leaves(a, total, h, st) : merkle_tp_leaves(a, total, h, st);
}
hipError_t merkle_leaf_hash_rows(const u64 *rows, u64 n_leaves, u32 width, u64 *digests, u32 batch, u64 ps_rows, u64 ps_digests, const HasherDev &h, hipStream_t st) {
    if (n_leaves == 0 || batch == 0) return hipSuccess;
    if (n_leaves & (n_leaves - 1)) return hipErrorInvalidValue;
    const u64 total = n_leaves * batch;
    const bool p2 = h.kind == hasher::POSEIDON2;
    if (!p2 && total <= coop_max_for(batch)) {
        dim3 block(256), grid((unsigned)((total * 16 + 255) / 256));
        hipLaunchKernelGGL(leaf_rows_coop_kernel, grid, block, 0, st, rows, n_leaves, width, digests, batch, ps_rows, ps_digests);
        return hipGetLastError();
    }
    return total >= tp_min_for(batch) ? merkle_tp_rows(rows, n_leaves, width, digests, batch, ps_rows, ps_digests, h, st)
                                           : hash_launch_rows(rows, n_leaves, width, digests, batch, ps_rows, ps_digests, h, st);
}
static hipError_t merkle_reduce_level(const u64 *in, u64 *out, u64 n_out, u32 batch, u64 ps, const HasherDev &h, hipStream_t st) {
    if (n_out == 0) return hipSuccess;
    const u64 total = n_out * batch;
    const bool p2 = h.kind == hasher::POSEIDON2;
    if (!p2 && total <= coop_max_for(batch)) {
        dim3 block(256), grid((unsigned)((total * 16 + 255) / 256));
        hipLaunchKernelGGL(node_coop_kernel, grid, block, 0, st, in, out, n_out, batch, ps);
        return hipGetLastError();
    }
    if (total < tp_min_for(batch)) return hash_launch_nodes(in, out, n_out, batch, ps, h, st);
    return (mx_serves(h) && mx_enabled()) ? merkle_mx_nodes(in, out, n_out, batch, ps, h, st) : merkle_tp_nodes(in, out, n_out, batch, ps, h, st);
}
// every level from `cnt` digests (at `levels`, the following levels stored behind it) down to the cap
hipError_t merkle_reduce_to_cap(u64 *levels, u64 cnt, u64 cap_n, u32 batch, u64 ps, const HasherDev &h, hipStream_t st) {
    if (batch == 0) return hipSuccess;
    if ((cnt & (cnt - 1)) || (cap_n & (cap_n - 1))) return hipErrorInvalidValue;
    u64 *lvl = levels;
    while (cnt > cap_n) {
        const u64 m = cnt / cap_n;
        // The levels above this point have too few nodes to fill the chip, one launch each: fuse them. QPGPU_TREE_TOP = the
        // largest subtree (digests per cap entry) handed to the fused kernel: 0 off, 32 (default: the last five levels, all
        // lane-cooperative), up to 512 (larger settings measured within noise of 32 with lockstep batches of 16 and 32).
        static const u32 top_m = [] { const char *e = getenv("QPGPU_TREE_TOP"); const int v = e ? atoi(e) : 32; return (u32)(v < 0 ? 0 : v > 512 ? 512 : v); }();
        if (top_m >= 2 && !tput(batch) && h.kind != hasher::POSEIDON2 && m <= top_m && cap_n <= 65535 && batch <= 65535) {
            if (m <= 32) hipLaunchKernelGGL((tree_top_kernel<32>), dim3((unsigned)cap_n, batch), dim3(256), 0, st, lvl, cnt, (u32)m, ps);
            else if (m <= 256) hipLaunchKernelGGL((tree_top_kernel<256>), dim3((unsigned)cap_n, batch), dim3(256), 0, st, lvl, cnt, (u32)m, ps);
            else hipLaunchKernelGGL((tree_top_kernel<512>), dim3((unsigned)cap_n, batch), dim3(512), 0, st, lvl, cnt, (u32)m, ps);
            return hipGetLastError();
        }
        hipError_t e = merkle_reduce_level(lvl, lvl + cnt * 4, cnt / 2, batch, ps, h, st);
        if (e != hipSuccess) return e;
        lvl += cnt * 4; cnt >>= 1;
    }
    return hipSuccess;
}
hipError_t poseidon2_hash_pad10_batch(const u64 *in, u64 len, u64 count, u64 *out, const poseidon2::Params *p2, bool qp_set, hipStream_t st) {
    if (count == 0) return hipSuccess;
    const dim3 grid((unsigned)((count + 255) / 256)), block(256);
    if (qp_set) hipLaunchKernelGGL((p2_pad10_sponge_kernel<true>), grid, block, 0, st, in, len, count, out, p2);
    else hipLaunchKernelGGL((p2_pad10_sponge_kernel<false>), grid, block, 0, st, in, len, count, out, p2);
    return hipGetLastError();
}
hipError_t poseidon_permute_batch(u64 *states, u64 n, const HasherDev &h, hipStream_t st) {
    if (n == 0) return hipSuccess;
    dim3 block(256), grid((unsigned)((n + 255) / 256));
    if (h.kind == hasher::POSEIDON2) hipLaunchKernelGGL((permute_kernel<Poseidon2P>), grid, block, 0, st, states, n, h.p2);
    else hipLaunchKernelGGL((permute_kernel<PoseidonV1>), grid, block, 0, st, states, n, h.p2);
    return hipGetLastError();
}
