// merkle_kernels_mx.hip — matrix-pipe build of the thread-per-hash kernels of stages s3 / s10 (column-major leaf sponges, tree
// levels, proof of work):
// plonky2's Poseidon — and qp-poseidon-core's Poseidon2 when that is the context's hasher — with the 22 partial (internal) rounds
// as one int8 GEMM on v_mfma_i32_32x32x32_i8 (poseidon_mfma.hpp), the full rounds on the vector ALU with the S-box products as
// rare-fold groups (gl64.hpp). Same digests as merkle_hash_impl.hpp, bit for bit
// (tests/test_merkle_gpu.py runs every build on the same inputs). 3.64 against 2.81 G permutations/s in registers
// (profiles/r03_poseidon_mfma.txt).
//
// What the matrix form asks of a kernel: the 61 KB constant table in LDS (workgroups of 512 threads, two per CU: four waves per
// SIMD at 128 VGPRs), every lane of a wave inside the permutation (MFMA and the half-wave swaps work on whole waves: a thread
// without work hashes a clamped index and does not store), and enough hashes per workgroup to pay for the table load (a thread of
// the tree-level kernel takes `per_thread` nodes). merkle_kernels.hip routes launches of qpgpu_tp_min_threads() hashes or more of
// the Poseidon hasher here; smaller launches and the Poseidon2 plug keep the other builds.
#define POSEIDON_GROUPED_SBOX 1
#include <hip/hip_runtime.h>
#include <vector>
#include "merkle.hpp"
#include "poseidon_mfma.hpp"
#include "prover_kernels.hpp"

using gl::u32;
using gl::u64;

namespace mx {
constexpr int WG = 512;
__constant__ u64 c_poseidon_rc[poseidon::ROUNDS * poseidon::WIDTH];
__device__ uint4 g_table[pmf::TABLE_BYTES / 16];        // plonky2's Poseidon
__device__ uint4 g_table_p2[pmf::TABLE_BYTES / 16];     // qp-poseidon-core's Poseidon2

#define MX_KERNEL __global__ void __launch_bounds__(WG) __attribute__((amdgpu_waves_per_eu(4, 4)))

// the two permutations of the build: the matrix table they need and the permutation itself
struct PoseidonV1 {
    static __device__ __forceinline__ const uint4 *table() { return g_table; }
    static __device__ __forceinline__ void permute(u64 (&s)[12], const poseidon2::Params *, const unsigned char *lds) { pmf::permute(s, c_poseidon_rc, lds); }
};
struct Poseidon2QP {
    static __device__ __forceinline__ const uint4 *table() { return g_table_p2; }
    static __device__ __forceinline__ void permute(u64 (&s)[12], const poseidon2::Params *p2, const unsigned char *lds) { pmf::permute_p2qp(s, *p2, lds); }
};
template <class Perm>
__device__ __forceinline__ const unsigned char *table_to_lds() {
    extern __shared__ uint4 mx_lds[];
    const uint4 *t = Perm::table();
    for (int i = threadIdx.x; i < pmf::TABLE_BYTES / 16; i += WG) mx_lds[i] = t[i];
    __syncthreads();
    return (const unsigned char *)mx_lds;
}
__device__ __forceinline__ u32 ilog2_64(u64 x) { return 63u - (u32)__clzll((long long)x); }

// leaf j = [src0 cols..., src1 cols...] at slot j, W > 4 (the launcher keeps hash_or_noop's copy case on the other builds)
template <class Perm>
MX_KERNEL leaf_hash_kernel(MerkleLeafArgs a, const poseidon2::Params *p2) {
    const unsigned char *lds = table_to_lds<Perm>();
    const u64 total = a.n_leaves * a.batch;
    u64 gj = blockIdx.x * (u64)WG + threadIdx.x;
    const bool live = gj < total;
    gj = live ? gj : total - 1;
    const u64 pr = gj >> ilog2_64(a.n_leaves), j = gj & (a.n_leaves - 1);
    a.src0 += pr * a.ps_src0; a.src1 += pr * a.ps_src1; a.digests += pr * a.ps_digests;
    const u32 W = a.ncols0 + a.ncols1;
    auto elem = [&](u32 c) -> u64 {
        return c < a.ncols0 ? a.src0[(u64)c * a.stride0 + j] : a.src1[(u64)(c - a.ncols0) * a.stride1 + j];
    };
    // (requesting the next block's columns between the two halves of the permutation was measured: 16 more live registers at the
    // 128-register cap spill, 3.27 instead of 3.47 G permutations/s on the wires-shaped tree, gpurun_out/mx/leaf_time.txt)
    u64 s[12];
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = 0;
    for (u32 c = 0; c < W; c += 8) {
#pragma unroll
        for (int i = 0; i < 8; i++)
            if (c + i < W) s[i] = elem(c + i);
        Perm::permute(s, p2, lds);
    }
    if (live) {
        u64 *out = a.digests + j * 4;
#pragma unroll
        for (int i = 0; i < 4; i++) out[i] = s[i];
    }
}

// one level: out[i] = two_to_one(in[2i], in[2i+1]); thread t of workgroup b takes nodes (b * per_thread + k) * WG + t
template <class Perm>
MX_KERNEL node_kernel(const u64 *in, u64 *out, u64 n_out, u32 batch, u64 ps, u32 per_thread, const poseidon2::Params *p2) {
    const unsigned char *lds = table_to_lds<Perm>();
    const u64 total = n_out * batch;
    const u32 sh = ilog2_64(n_out);
    for (u32 k = 0; k < per_thread; k++) {
        u64 gi = ((u64)blockIdx.x * per_thread + k) * WG + threadIdx.x;
        const bool live = gi < total;
        gi = live ? gi : total - 1;
        const u64 pr = gi >> sh, i = gi & (n_out - 1);
        const ulonglong2 *p = reinterpret_cast<const ulonglong2 *>(in + pr * ps + i * 8);
        const ulonglong2 a = p[0], b = p[1], c = p[2], d = p[3];
        u64 s[12];
        s[0] = a.x; s[1] = a.y; s[2] = b.x; s[3] = b.y; s[4] = c.x; s[5] = c.y; s[6] = d.x; s[7] = d.y;
        s[8] = s[9] = s[10] = s[11] = 0;
        Perm::permute(s, p2, lds);
        if (live) {
            ulonglong2 *o = reinterpret_cast<ulonglong2 *>(out + pr * ps + i * 4);
            o[0] = make_ulonglong2(s[0], s[1]);
            o[1] = make_ulonglong2(s[2], s[3]);
        }
    }
}

// s10 fri_proof_of_work, same contract as pow_kernel of merkle_hash_impl.hpp (minimum accepted nonce per proof by atomicMin;
// workgroups chunk-major over the proofs, the proof rotating with the chunk). A workgroup leaves BEFORE it loads the table when its
// proof has its nonce already or when all of its 512 candidates lie above a nonce found in this launch.
template <class Perm>
MX_KERNEL pow_kernel(PowArgs a, const poseidon2::Params *p2) {
    const u32 chunk = blockIdx.x / a.batch, pr = (blockIdx.x % a.batch + chunk) % a.batch;
    const u64 first = (u64)chunk * WG;
    if (first >= a.count) return;
    const u64 base = a.bases[pr];
    if (base == ~0ull) return;
    // the exit is decided ONCE per workgroup (one wave's load, broadcast through LDS): waves that each looked at results[pr]
    // could decide differently while another workgroup lowers it, and the survivors would wait at the table load's barrier for
    // waves that have left
    __shared__ u32 s_leave;
    if (threadIdx.x == 0) s_leave = __hip_atomic_load(&a.results[pr], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < base + first ? 1u : 0u;
    __syncthreads();
    if (s_leave) return;
    const unsigned char *lds = table_to_lds<Perm>();
    const u64 idx = first + threadIdx.x, nonce = base + idx;
    const u64 *st = a.states + 12 * (u64)pr;
    u64 s[12];
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = (i == (int)a.pos) ? nonce : st[i];
    Perm::permute(s, p2, lds);
    if (idx < a.count && (s[7] >> (64 - a.pow_bits)) == 0) atomicMin((unsigned long long *)&a.results[pr], (unsigned long long)nonce);
}

// device self-test (below): one permutation per thread
template <class Perm>
MX_KERNEL selftest_kernel(u64 *states, const poseidon2::Params *p2) {
    const unsigned char *lds = table_to_lds<Perm>();
    const u64 t = blockIdx.x * (u64)WG + threadIdx.x;
    u64 s[12];
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = states[t * 12 + i];
    Perm::permute(s, p2, lds);
#pragma unroll
    for (int i = 0; i < 12; i++) states[t * 12 + i] = s[i];
}
}  // namespace mx

template <class Perm>
static hipError_t allow_table_in_lds() {
    hipError_t e = hipFuncSetAttribute((const void *)mx::leaf_hash_kernel<Perm>, hipFuncAttributeMaxDynamicSharedMemorySize, pmf::TABLE_BYTES);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void *)mx::node_kernel<Perm>, hipFuncAttributeMaxDynamicSharedMemorySize, pmf::TABLE_BYTES);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void *)mx::pow_kernel<Perm>, hipFuncAttributeMaxDynamicSharedMemorySize, pmf::TABLE_BYTES);
    return e;
}
// Once per device, right after a table has been uploaded: four workgroups of the matrix-build permutation (eight waves each, two
// per SIMD: the occupancy at which a mis-scheduled MFMA destination shows, poseidon_mfma.hpp) on 2 048 fixed states incl. extremes,
// against the plain permutation on the host, then the production kernels themselves (see below). A mismatch fails the context's set-up with hipErrorAssert instead of letting a
// platform on which the matrix form misbehaves hash anything (QPGPU_MX=0 runs without the matrix build and without this test).
template <class Perm, class HostPerm>
static hipError_t device_selftest(HostPerm host_perm, const poseidon2::Params *host_p2) {
    constexpr int N = 4 * mx::WG;
    std::vector<u64> in((size_t)N * 12), out((size_t)N * 12);
    u64 seed = 0x9E3779B97F4A7C15ull;
    for (int t = 0; t < N; t++)
        for (int i = 0; i < 12; i++) {
            seed = seed * 6364136223846793005ull + 1442695040888963407ull;
            u64 v = (seed ^ (seed >> 31)) % gl::P;
            if (t % 9 == 0) v = (i & 1) ? gl::P - 1 - (v & 3) : (v & 7);
            in[(size_t)t * 12 + i] = v;
        }
    u64 *d = nullptr; poseidon2::Params *dp = nullptr; hipStream_t st = nullptr;
    hipError_t e = hipMalloc((void **)&d, in.size() * 8);
    if (e == hipSuccess && host_p2) { e = hipMalloc((void **)&dp, sizeof *host_p2); if (e == hipSuccess) e = hipMemcpy(dp, host_p2, sizeof *host_p2, hipMemcpyHostToDevice); }
    if (e == hipSuccess) e = hipMemcpy(d, in.data(), in.size() * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void *)mx::selftest_kernel<Perm>, hipFuncAttributeMaxDynamicSharedMemorySize, pmf::TABLE_BYTES);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(mx::selftest_kernel<Perm>, dim3(N / mx::WG), dim3(mx::WG), pmf::TABLE_BYTES, st, d, dp);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(st);
    }
    if (e == hipSuccess) e = hipMemcpy(out.data(), d, out.size() * 8, hipMemcpyDeviceToHost);
    if (e == hipSuccess)
        for (int t = 0; t < N && e == hipSuccess; t++) {
            u64 s[12];
            for (int i = 0; i < 12; i++) s[i] = in[(size_t)t * 12 + i];
            host_perm(s);
            for (int i = 0; i < 12; i++) if (s[i] != out[(size_t)t * 12 + i]) e = hipErrorAssert;
        }
    // The hazard the test guards against depends on each kernel's own register allocation, so the kernels that SHIP run too,
    // at the occupancy they ship at (1 024 workgroups: two per CU, four waves per SIMD): one leaf build (2^19 sponges of 5
    // columns), one tree level over its digests (2^18 nodes) and one proof-of-work launch, each held against the host — for
    // the two hashing launches one lane of every wave (a mis-scheduled MFMA destination corrupts whole waves), for the proof of
    // work the minimum nonce itself.
    constexpr u64 NL = 1ull << 19, NN = NL / 2, POW_COUNT = 1ull << 18;
    constexpr u32 WCOLS = 5, POW_BITS = 11, POW_BATCH = 2;
    u64 *d_cols = nullptr, *d_dig = nullptr, *d_pow = nullptr;
    std::vector<u64> cols((size_t)WCOLS * NL), dig(4 * (NL + NN)), powbuf(12 * POW_BATCH + 2 * POW_BATCH);
    if (e == hipSuccess) {
        for (size_t i = 0; i < cols.size(); i++) { seed = seed * 6364136223846793005ull + 1442695040888963407ull; cols[i] = (seed ^ (seed >> 29)) % gl::P; }
        for (u32 b = 0; b < POW_BATCH; b++) {
            for (int i = 0; i < 12; i++) { seed = seed * 6364136223846793005ull + 1442695040888963407ull; powbuf[12 * b + i] = (seed ^ (seed >> 29)) % gl::P; }
            powbuf[12 * POW_BATCH + b] = 1000 * (b + 1);            // bases
            powbuf[12 * POW_BATCH + POW_BATCH + b] = ~0ull;         // results
        }
        e = hipMalloc((void **)&d_cols, cols.size() * 8);
        if (e == hipSuccess) e = hipMalloc((void **)&d_dig, dig.size() * 8);
        if (e == hipSuccess) e = hipMalloc((void **)&d_pow, powbuf.size() * 8);
        if (e == hipSuccess) e = hipMemcpy(d_cols, cols.data(), cols.size() * 8, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(d_pow, powbuf.data(), powbuf.size() * 8, hipMemcpyHostToDevice);
    }
    if (e == hipSuccess) {
        MerkleLeafArgs la{};
        la.src0 = d_cols; la.stride0 = NL; la.ncols0 = WCOLS; la.n_leaves = NL; la.digests = d_dig; la.batch = 1;
        hipLaunchKernelGGL(mx::leaf_hash_kernel<Perm>, dim3((unsigned)(NL / mx::WG)), dim3(mx::WG), pmf::TABLE_BYTES, st, la, dp);
        hipLaunchKernelGGL(mx::node_kernel<Perm>, dim3((unsigned)(NN / mx::WG)), dim3(mx::WG), pmf::TABLE_BYTES, st, d_dig, d_dig + 4 * NL, NN, 1u, (u64)0, 1u, dp);
        PowArgs pa{};
        pa.states = d_pow; pa.bases = d_pow + 12 * POW_BATCH; pa.results = d_pow + 12 * POW_BATCH + POW_BATCH;
        pa.pos = 3; pa.pow_bits = POW_BITS; pa.batch = POW_BATCH; pa.count = POW_COUNT;
        hipLaunchKernelGGL(mx::pow_kernel<Perm>, dim3((unsigned)(POW_COUNT / mx::WG * POW_BATCH)), dim3(mx::WG), pmf::TABLE_BYTES, st, pa, dp);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e == hipSuccess) e = hipMemcpy(dig.data(), d_dig, dig.size() * 8, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(powbuf.data(), d_pow, powbuf.size() * 8, hipMemcpyDeviceToHost);
    }
    if (e == hipSuccess) {
        for (u64 w = 0; w < NL / 64 && e == hipSuccess; w++) {           // one lane of every wave of the leaf build
            const u64 j = w * 64 + (w * 7) % 64;
            u64 s[12] = {0};
            for (u32 c = 0; c < WCOLS; c++) s[c] = cols[(size_t)c * NL + j];
            host_perm(s);
            for (int i = 0; i < 4; i++) if (s[i] != dig[4 * j + i]) e = hipErrorAssert;
        }
        for (u64 w = 0; w < NN / 64 && e == hipSuccess; w++) {           // one lane of every wave of the tree level
            const u64 i0 = w * 64 + (w * 11) % 64;
            u64 s[12] = {0};
            for (int i = 0; i < 8; i++) s[i] = dig[8 * i0 + i];
            host_perm(s);
            for (int i = 0; i < 4; i++) if (s[i] != dig[4 * NL + 4 * i0 + i]) e = hipErrorAssert;
        }
        for (u32 b = 0; b < POW_BATCH && e == hipSuccess; b++) {         // the minimum accepted nonce, found again on the host
            const u64 base = 1000 * (b + 1), got = powbuf[12 * POW_BATCH + POW_BATCH + b];
            u64 want = ~0ull;
            for (u64 k = 0; k < POW_COUNT && want == ~0ull; k++) {
                u64 s[12];
                for (int i = 0; i < 12; i++) s[i] = i == 3 ? base + k : powbuf[12 * b + i];
                host_perm(s);
                if ((s[7] >> (64 - POW_BITS)) == 0) want = base + k;
            }
            if (got != want) e = hipErrorAssert;
        }
    }
    if (d_cols) (void)hipFree(d_cols);
    if (d_dig) (void)hipFree(d_dig);
    if (d_pow) (void)hipFree(d_pow);
    if (st) (void)hipStreamDestroy(st);
    if (d) (void)hipFree(d);
    if (dp) (void)hipFree(dp);
    return e;
}
// The tables are functions of the round constants. Before anything is uploaded, the integer emulation of the device schedule
// (same table bytes, same recombination code) is held against the plain permutation on the host.
hipError_t merkle_mx_upload_constants(const u64 *rc360) {
    std::vector<unsigned char> tab(pmf::TABLE_BYTES);
    if (!pmf::build_tables(rc360, tab.data()) || !pmf::host_selfcheck(rc360, tab.data(), 64)) return hipErrorInvalidValue;
    hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(mx::c_poseidon_rc), rc360, sizeof(u64) * poseidon::ROUNDS * poseidon::WIDTH);
    if (e != hipSuccess) return e;
    e = hipMemcpyToSymbol(HIP_SYMBOL(mx::g_table), tab.data(), pmf::TABLE_BYTES);
    if (e == hipSuccess) e = allow_table_in_lds<mx::PoseidonV1>();
    return e != hipSuccess ? e : device_selftest<mx::PoseidonV1>([&](u64 (&s)[12]) { poseidon::permute(s, rc360); }, nullptr);
}
hipError_t merkle_upload_p2_tables(const poseidon2::Params &qp) {
    std::vector<unsigned char> tab(pmf::TABLE_BYTES);
    if (!pmf::build_tables_p2(qp, tab.data()) || !pmf::host_selfcheck_p2(qp, tab.data(), 64)) return hipErrorInvalidValue;
    hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(mx::g_table_p2), tab.data(), pmf::TABLE_BYTES);
    if (e == hipSuccess) e = allow_table_in_lds<mx::Poseidon2QP>();
    return e != hipSuccess ? e : device_selftest<mx::Poseidon2QP>([&](u64 (&s)[12]) { poseidon2::permute(s, qp); }, &qp);
}
hipError_t merkle_mx_leaves(const MerkleLeafArgs &a, u64 total, const HasherDev &h, hipStream_t st) {
    const u64 blocks = (total + mx::WG - 1) / mx::WG;
    if (blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    if (h.kind == hasher::POSEIDON2) hipLaunchKernelGGL(mx::leaf_hash_kernel<mx::Poseidon2QP>, dim3((unsigned)blocks), dim3(mx::WG), pmf::TABLE_BYTES, st, a, h.p2);
    else hipLaunchKernelGGL(mx::leaf_hash_kernel<mx::PoseidonV1>, dim3((unsigned)blocks), dim3(mx::WG), pmf::TABLE_BYTES, st, a, h.p2);
    return hipGetLastError();
}
hipError_t merkle_mx_nodes(const u64 *in, u64 *out, u64 n_out, u32 batch, u64 ps, const HasherDev &h, hipStream_t st) {
    const u64 total = n_out * batch;
    // 512 workgroup slots on the chip (two per CU): one node per thread up to 2^18 nodes, then more nodes per thread
    u64 per_thread = total >> 18;
    per_thread = per_thread < 1 ? 1 : per_thread > 8 ? 8 : per_thread;
    const u64 blocks = (total + mx::WG * per_thread - 1) / (mx::WG * per_thread);
    if (blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    if (h.kind == hasher::POSEIDON2) hipLaunchKernelGGL(mx::node_kernel<mx::Poseidon2QP>, dim3((unsigned)blocks), dim3(mx::WG), pmf::TABLE_BYTES, st, in, out, n_out, batch, ps, (u32)per_thread, h.p2);
    else hipLaunchKernelGGL(mx::node_kernel<mx::PoseidonV1>, dim3((unsigned)blocks), dim3(mx::WG), pmf::TABLE_BYTES, st, in, out, n_out, batch, ps, (u32)per_thread, h.p2);
    return hipGetLastError();
}
hipError_t merkle_mx_pow(const PowArgs &a, const HasherDev &h, hipStream_t st) {
    const u64 chunks = (a.count + mx::WG - 1) / mx::WG;
    if (chunks * a.batch > 0x7FFFFFFFull) return hipErrorInvalidValue;
    if (h.kind == hasher::POSEIDON2) hipLaunchKernelGGL(mx::pow_kernel<mx::Poseidon2QP>, dim3((unsigned)(chunks * a.batch)), dim3(mx::WG), pmf::TABLE_BYTES, st, a, h.p2);
    else hipLaunchKernelGGL(mx::pow_kernel<mx::PoseidonV1>, dim3((unsigned)(chunks * a.batch)), dim3(mx::WG), pmf::TABLE_BYTES, st, a, h.p2);
    return hipGetLastError();
}
