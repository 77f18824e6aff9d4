// merkle_api.cpp — C ABI for stage s3 (Poseidon hashing, MerkleTree::new).
#include <hip/hip_runtime.h>
#include <mutex>
#include "ctx.hpp"
#include "merkle.hpp"
#include "poseidon.hpp"

int merkle_ensure_constants(qpgpu_ctx *ctx) {
    if (ctx->hasher_generation == hasher::generation()) return QPGPU_OK;
    QP_HIP(ctx, merkle_upload_constants(poseidon::host_hash_round_constants()));
    QP_HIP(ctx, merkle_select_hasher(hasher::kind(), &hasher::p2_params()));
    ctx->hasher_generation = hasher::generation();
    return QPGPU_OK;
}

extern "C" int qpgpu_set_hasher(int kind, const uint64_t *params, size_t n_words) {
    if (kind == hasher::POSEIDON) { hasher::set(hasher::POSEIDON, nullptr); return QPGPU_OK; }
    if (kind != hasher::POSEIDON2 || !params || n_words != (size_t)poseidon2::PARAM_WORDS) return QPGPU_EINVAL;
    poseidon2::Params p;
    const uint64_t *w = params;
    for (int i = 0; i < 96; i++) p.rc_ext[i] = gl::canon(*w++);
    for (int i = 0; i < 22; i++) p.rc_int[i] = gl::canon(*w++);
    for (int i = 0; i < 12; i++) p.diag_m1[i] = gl::canon(*w++);
    for (int i = 0; i < 16; i++) p.m4[i] = gl::canon(*w++);
    hasher::set(hasher::POSEIDON2, &p);
    return QPGPU_OK;
}
extern "C" int qpgpu_get_hasher(void) { return hasher::kind(); }

// digests: level 0 (n_leaves) then each parent level down to the cap level, concatenated.
int merkle_build(qpgpu_ctx *ctx, const MerkleLeafArgs &leaf, unsigned log_leaves, unsigned cap_height, uint64_t *d_digests) {
    int rc = merkle_ensure_constants(ctx);
    if (rc) return rc;
    if (cap_height > log_leaves) return ctx->fail(QPGPU_EINVAL, "merkle: cap_height exceeds tree height");
    ctx->prof_begin("merkle_leaf_hash");
    hipError_t e = merkle_leaf_hash(leaf, ctx->stream);
    ctx->prof_end();
    QP_HIP(ctx, e);
    ctx->prof_begin("merkle_nodes");
    e = merkle_reduce_to_cap(d_digests, 1ull << log_leaves, 1ull << cap_height, ctx->stream);
    ctx->prof_end();
    QP_HIP(ctx, e);
    return QPGPU_OK;
}

extern "C" {

size_t qpgpu_merkle_digest_count(unsigned log_leaves, unsigned cap_height) {
    if (cap_height > log_leaves) return 0;
    return (size_t)((2ull << log_leaves) - (1ull << cap_height));
}

int qpgpu_merkle_build_dev(qpgpu_ctx *ctx, const uint64_t *d_cols, uint64_t col_stride, uint32_t n_cols,
                           unsigned log_leaves, unsigned cap_height, uint64_t *d_digests, uint64_t *h_cap_out) {
    if (!ctx) return QPGPU_EINVAL;
    QP_DEV(ctx);
    if (!d_cols || !d_digests || n_cols == 0) return ctx->fail(QPGPU_EINVAL, "merkle: null buffer or no columns");
    if (log_leaves > 40) return ctx->fail(QPGPU_EINVAL, "merkle: log_leaves out of range");
    MerkleLeafArgs a{};
    a.src0 = d_cols; a.stride0 = col_stride; a.ncols0 = n_cols; a.src1 = nullptr; a.ncols1 = 0; a.stride1 = 0;
    a.n_leaves = 1ull << log_leaves; a.digests = d_digests;
    int rc = merkle_build(ctx, a, log_leaves, cap_height, d_digests);
    if (rc) return rc;
    if (h_cap_out) {
        size_t total = qpgpu_merkle_digest_count(log_leaves, cap_height), cap_n = (size_t)1 << cap_height;
        QP_HIP(ctx, hipMemcpyAsync(h_cap_out, d_digests + (total - cap_n) * 4, cap_n * 4 * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
        QP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return QPGPU_OK;
}

int qpgpu_merkle_build_rows_dev(qpgpu_ctx *ctx, const uint64_t *d_rows, uint32_t width, unsigned log_leaves,
                                unsigned cap_height, uint64_t *d_digests, uint64_t *h_cap_out) {
    if (!ctx) return QPGPU_EINVAL;
    QP_DEV(ctx);
    if (!d_rows || !d_digests || width == 0) return ctx->fail(QPGPU_EINVAL, "merkle: null buffer or zero width");
    if (cap_height > log_leaves) return ctx->fail(QPGPU_EINVAL, "merkle: cap_height exceeds tree height");
    int rc = merkle_ensure_constants(ctx);
    if (rc) return rc;
    uint64_t cnt = 1ull << log_leaves;
    QP_HIP(ctx, merkle_leaf_hash_rows(d_rows, cnt, width, d_digests, ctx->stream));
    QP_HIP(ctx, merkle_reduce_to_cap(d_digests, cnt, 1ull << cap_height, ctx->stream));
    uint64_t *lvl = d_digests;
    while (cnt > (1ull << cap_height)) { lvl += cnt * 4; cnt >>= 1; }
    if (h_cap_out) {
        size_t cap_n = (size_t)1 << cap_height;
        QP_HIP(ctx, hipMemcpyAsync(h_cap_out, lvl, cap_n * 4 * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
        QP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return QPGPU_OK;
}

int qpgpu_poseidon_permute_dev(qpgpu_ctx *ctx, uint64_t *d_states, size_t n) {
    if (!ctx) return QPGPU_EINVAL;
    QP_DEV(ctx);
    if (!d_states && n) return ctx->fail(QPGPU_EINVAL, "poseidon: null buffer");
    int rc = merkle_ensure_constants(ctx);
    if (rc) return rc;
    QP_HIP(ctx, poseidon_permute_batch(d_states, n, ctx->stream));
    return QPGPU_OK;
}

}  // extern "C"
