// merkle_hash_impl.hpp — the thread-per-hash kernels of stage s3 / s10 (leaf sponges, tree levels, proof of work), included by
// merkle_kernels.hip (latency build) and merkle_kernels_tp.hip (throughput build: GL_RARE_BRANCH, see gl64.hpp). The including unit
// declares `__constant__ u64 c_poseidon_rc[]` first and wraps this file in the namespace its kernels should be listed under.
//
// Layout: the LDE is column-major and already in leaf order (bit-reversed slots), so thread j reads slot j of each column: a wave
// reads 512 contiguous bytes per column, no transpose in HBM. One thread = one sponge (state in 24 VGPRs). Integer-ALU-bound:
// ceil(W/8) permutations per leaf.
namespace {

struct PoseidonV1 {
    static __device__ __forceinline__ void permute(u64 (&s)[12], const poseidon2::Params *) { poseidon::permute(s, c_poseidon_rc); }
};
struct Poseidon2P {   // the parameter plug: same sponge and tree code, other permutation; parameters of the caller's context
    static __device__ __forceinline__ void permute(u64 (&s)[12], const poseidon2::Params *p2) { poseidon2::permute(s, *p2); }
};
struct Poseidon2QP {  // the plug when the context's block is qp-poseidon-core's set: multiplication-free external layers
    static __device__ __forceinline__ void permute(u64 (&s)[12], const poseidon2::Params *p2) { poseidon2::permute_qp(s, *p2); }
};

// tree of the batch a global leaf / node index belongs to (counts are powers of two)
__device__ __forceinline__ u32 ilog2_64(u64 x) { return 63u - (u32)__clzll((long long)x); }

// leaf j = [src0 cols..., src1 cols...] at slot j (each source column-major with its own stride).
template <class Perm>
__global__ void __launch_bounds__(256) leaf_hash_kernel(MerkleLeafArgs a, const poseidon2::Params *p2) {
    const u64 gj = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (gj >= a.n_leaves * a.batch) return;
    const u64 pr = gj >> ilog2_64(a.n_leaves), j = gj & (a.n_leaves - 1);
    a.src0 += pr * a.ps_src0; a.src1 += pr * a.ps_src1; a.digests += pr * a.ps_digests;
    const u32 W = a.ncols0 + a.ncols1;
    u64 *out = a.digests + j * 4;
    auto elem = [&](u32 c) -> u64 {
        return c < a.ncols0 ? a.src0[(u64)c * a.stride0 + j] : a.src1[(u64)(c - a.ncols0) * a.stride1 + j];
    };
    if (W <= 4) {  // hash_or_noop: short rows are copied
        for (u32 c = 0; c < 4; c++) out[c] = c < W ? gl::canon(elem(c)) : 0;
        return;
    }
    u64 s[12];
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = 0;
    for (u32 c = 0; c < W; c += 8) {
        // overwrite-mode absorption of up to 8 elements
#pragma unroll
        for (int i = 0; i < 8; i++)
            if (c + i < W) s[i] = elem(c + i);
        Perm::permute(s, p2);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) out[i] = s[i];
}

// row-major leaves (FRI round trees: leaf = 2^arity ext values = contiguous felts)
template <class Perm>
__global__ void __launch_bounds__(256) leaf_hash_rows_kernel(const u64 *rows, u64 n_leaves, u32 width, u64 *digests, u32 batch, u64 ps_rows, u64 ps_digests, const poseidon2::Params *p2) {
    const u64 gj = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (gj >= n_leaves * batch) return;
    const u64 pr = gj >> ilog2_64(n_leaves), j = gj & (n_leaves - 1);
    rows += pr * ps_rows; digests += pr * ps_digests;
    const u64 *row = rows + j * width;
    u64 *out = digests + j * 4;
    if (width <= 4) {
        for (u32 c = 0; c < 4; c++) out[c] = c < width ? gl::canon(row[c]) : 0;
        return;
    }
    u64 s[12];
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = 0;
    for (u32 c = 0; c < width; c += 8) {
#pragma unroll
        for (int i = 0; i < 8; i++)
            if (c + i < width) s[i] = row[c + i];
        Perm::permute(s, p2);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) out[i] = s[i];
}

// one level: out[i] = two_to_one(in[2i], in[2i+1])
template <class Perm>
__global__ void __launch_bounds__(256) node_kernel(const u64 *in, u64 *out, u64 n_out, u32 batch, u64 ps, const poseidon2::Params *p2) {
    const u64 gi = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (gi >= n_out * batch) return;
    const u64 pr = gi >> ilog2_64(n_out), i = gi & (n_out - 1);
    in += pr * ps; out += pr * ps;
    u64 s[12];
    const ulonglong2 *p = reinterpret_cast<const ulonglong2 *>(in + i * 8);
    ulonglong2 a = p[0], b = p[1], c = p[2], d = p[3];
    s[0] = a.x; s[1] = a.y; s[2] = b.x; s[3] = b.y; s[4] = c.x; s[5] = c.y; s[6] = d.x; s[7] = d.y;
    s[8] = s[9] = s[10] = s[11] = 0;
    Perm::permute(s, p2);
    ulonglong2 *o = reinterpret_cast<ulonglong2 *>(out + i * 4);
    o[0] = make_ulonglong2(s[0], s[1]);
    o[1] = make_ulonglong2(s[2], s[3]);
}

// s10 fri_proof_of_work: candidate nonce at `pos` of the pre-absorbed duplex state; accept when the last rate
// element has >= pow_bits leading zeros; result = minimum accepted nonce in [base, base + count). Workgroups are numbered
// chunk-major over the proofs of the batch (chunk c of every proof before chunk c + 1 of any), and a workgroup whose
// candidates are all above a nonce already found for its proof leaves at once: the expected work per proof is about
// 2^pow_bits permutations plus what is in flight, not the whole span.
template <class Perm>
__global__ void __launch_bounds__(256) pow_kernel(PowArgs a, const poseidon2::Params *p2) {
    // chunk-major over the proofs, and the proof a workgroup serves rotates with the chunk: workgroups go to the 8 XCDs round
    // robin, so with a fixed assignment (batch a multiple of 8) each proof's candidates would all run on one XCD, and the XCD whose
    // proofs find their nonce last would finish the launch alone
    const u32 chunk = blockIdx.x / a.batch, pr = (blockIdx.x % a.batch + chunk) % a.batch;
    const u64 idx = (u64)chunk * blockDim.x + threadIdx.x;
    if (idx >= a.count) return;
    const u64 base = a.bases[pr];
    if (base == ~0ull) return;                 // this proof already has its nonce
    const u64 nonce = base + idx;
    if (__hip_atomic_load(&a.results[pr], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nonce) return;
    const u64 *st = a.states + 12 * (u64)pr;
    u64 s[12];
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = (i == (int)a.pos) ? nonce : st[i];
    Perm::permute(s, p2);
    if ((s[7] >> (64 - a.pow_bits)) == 0) atomicMin((unsigned long long *)&a.results[pr], (unsigned long long)nonce);
}


hipError_t hash_launch_leaves(const MerkleLeafArgs &a, u64 total, const HasherDev &h, hipStream_t st) {
    dim3 block(256), grid((unsigned)((total + 255) / 256));
    if (h.kind == hasher::POSEIDON2 && h.qp) hipLaunchKernelGGL((leaf_hash_kernel<Poseidon2QP>), grid, block, 0, st, a, h.p2);
    else if (h.kind == hasher::POSEIDON2) hipLaunchKernelGGL((leaf_hash_kernel<Poseidon2P>), grid, block, 0, st, a, h.p2);
    else hipLaunchKernelGGL((leaf_hash_kernel<PoseidonV1>), grid, block, 0, st, a, h.p2);
    return hipGetLastError();
}
hipError_t hash_launch_rows(const u64 *rows, u64 n_leaves, u32 width, u64 *digests, u32 batch, u64 ps_rows, u64 ps_digests, const HasherDev &h, hipStream_t st) {
    const u64 total = n_leaves * batch;
    dim3 block(256), grid((unsigned)((total + 255) / 256));
    if (h.kind == hasher::POSEIDON2 && h.qp) hipLaunchKernelGGL((leaf_hash_rows_kernel<Poseidon2QP>), grid, block, 0, st, rows, n_leaves, width, digests, batch, ps_rows, ps_digests, h.p2);
    else if (h.kind == hasher::POSEIDON2) hipLaunchKernelGGL((leaf_hash_rows_kernel<Poseidon2P>), grid, block, 0, st, rows, n_leaves, width, digests, batch, ps_rows, ps_digests, h.p2);
    else hipLaunchKernelGGL((leaf_hash_rows_kernel<PoseidonV1>), grid, block, 0, st, rows, n_leaves, width, digests, batch, ps_rows, ps_digests, h.p2);
    return hipGetLastError();
}
hipError_t hash_launch_nodes(const u64 *in, u64 *out, u64 n_out, u32 batch, u64 ps, const HasherDev &h, hipStream_t st) {
    const u64 total = n_out * batch;
    unsigned threads = total >= 256 ? 256 : 64;
    dim3 block(threads), grid((unsigned)((total + threads - 1) / threads));
    if (h.kind == hasher::POSEIDON2 && h.qp) hipLaunchKernelGGL((node_kernel<Poseidon2QP>), grid, block, 0, st, in, out, n_out, batch, ps, h.p2);
    else if (h.kind == hasher::POSEIDON2) hipLaunchKernelGGL((node_kernel<Poseidon2P>), grid, block, 0, st, in, out, n_out, batch, ps, h.p2);
    else hipLaunchKernelGGL((node_kernel<PoseidonV1>), grid, block, 0, st, in, out, n_out, batch, ps, h.p2);
    return hipGetLastError();
}
hipError_t hash_launch_pow(const PowArgs &a, dim3 g, const HasherDev &h, hipStream_t st) {
    if (h.kind == hasher::POSEIDON2 && h.qp) hipLaunchKernelGGL((pow_kernel<Poseidon2QP>), g, dim3(256), 0, st, a, h.p2);
    else if (h.kind == hasher::POSEIDON2) hipLaunchKernelGGL((pow_kernel<Poseidon2P>), g, dim3(256), 0, st, a, h.p2);
    else hipLaunchKernelGGL((pow_kernel<PoseidonV1>), g, dim3(256), 0, st, a, h.p2);
    return hipGetLastError();
}

}  // namespace
