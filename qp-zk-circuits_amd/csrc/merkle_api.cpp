// merkle_api.cpp — C ABI for stage s3 (Poseidon hashing, MerkleTree::new).
#include <hip/hip_runtime.h>
#include <cstring>
#include <mutex>
#include "ctx.hpp"
#include "merkle.hpp"
#include "poseidon.hpp"

bool qpgpu_ctx::hasher_is_qp() const {
    const poseidon2::Params &q = poseidon2::qp_params();
    return memcmp(hasher.p2.rc_ext, q.rc_ext, sizeof q.rc_ext) == 0 && memcmp(hasher.p2.rc_int, q.rc_int, sizeof q.rc_int) == 0 &&
           memcmp(hasher.p2.diag_m1, q.diag_m1, sizeof q.diag_m1) == 0 && memcmp(hasher.p2.m4, q.m4, sizeof q.m4) == 0;
}

int merkle_ensure_constants(qpgpu_ctx *ctx) {
    // plonky2's Poseidon round constants live in __constant__ memory, the same for every context: once per device
    static std::mutex mu;
    static bool uploaded[64] = {false}, uploaded_p2[64] = {false};
    {
        std::lock_guard<std::mutex> lk(mu);
        const int dev = ctx->device;
        if (dev < 0 || dev >= 64) return ctx->fail(QPGPU_EINVAL, "device index out of range");
        if (!uploaded[dev]) {
            QP_HIP(ctx, merkle_upload_constants(poseidon::host_hash_round_constants()));
            uploaded[dev] = true;
        }
        if (ctx->hasher.kind == hasher::POSEIDON2 && ctx->hasher_is_qp() && !uploaded_p2[dev] && merkle_mx_in_use()) {
            QP_HIP(ctx, merkle_upload_p2_tables(poseidon2::qp_params()));
            uploaded_p2[dev] = true;
        }
    }
    if (ctx->hasher.kind == hasher::POSEIDON2 && !ctx->d_p2) {   // the context's Poseidon2 parameter block
        void *v = nullptr;
        QP_HIP(ctx, hipMalloc(&v, sizeof(poseidon2::Params)));
        ctx->d_p2 = (poseidon2::Params *)v;
        QP_HIP(ctx, hipMemcpyAsync(ctx->d_p2, &ctx->hasher.p2, sizeof(poseidon2::Params), hipMemcpyHostToDevice, ctx->stream));
        QP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return QPGPU_OK;
}

int qpgpu_ctx::ensure_p2_app() {
    if (d_p2_app) return QPGPU_OK;
    void *v = nullptr;
    QP_HIP(this, hipMalloc(&v, sizeof(poseidon2::Params)));
    // pageable source with static lifetime (qp_params() is a function-local static); the copy is ordered on the ctx stream
    hipError_t e = hipMemcpyAsync(v, &poseidon2::qp_params(), sizeof(poseidon2::Params), hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e != hipSuccess) { (void)hipFree(v); return hip_fail(e, "upload of the Poseidon2 application parameters"); }
    d_p2_app = (poseidon2::Params *)v;
    return QPGPU_OK;
}

static int parse_p2(const uint64_t *params, size_t n_words, poseidon2::Params &p) {
    if (!params || n_words != (size_t)poseidon2::PARAM_WORDS) return QPGPU_EINVAL;
    const uint64_t *w = params;
    for (int i = 0; i < 96; i++) p.rc_ext[i] = gl::canon(*w++);
    for (int i = 0; i < 22; i++) p.rc_int[i] = gl::canon(*w++);
    for (int i = 0; i < 12; i++) p.diag_m1[i] = gl::canon(*w++);
    for (int i = 0; i < 16; i++) p.m4[i] = gl::canon(*w++);
    return QPGPU_OK;
}

// process default (seeds new contexts; kept for callers of the first ABI)
extern "C" int qpgpu_set_hasher(int kind, const uint64_t *params, size_t n_words) {
    if (kind == hasher::POSEIDON) { hasher::set_process_default(hasher::POSEIDON, nullptr); return QPGPU_OK; }
    if (kind != hasher::POSEIDON2) return QPGPU_EINVAL;
    poseidon2::Params p;
    if (parse_p2(params, n_words, p) != QPGPU_OK) return QPGPU_EINVAL;
    hasher::set_process_default(hasher::POSEIDON2, &p);
    return QPGPU_OK;
}
extern "C" int qpgpu_get_hasher(void) { return hasher::kind(); }

extern "C" int qpgpu_ctx_set_hasher(qpgpu_ctx *ctx, int kind, const uint64_t *params, size_t n_words) {
    if (!ctx) return QPGPU_EINVAL;
    QP_DEV(ctx);
    if (ctx->hasher_in_use) return ctx->fail(QPGPU_EINVAL, "ctx_set_hasher: circuits or oracles already exist on this context; choose the hasher first");
    if (kind == hasher::POSEIDON) { ctx->hasher.kind = hasher::POSEIDON; return QPGPU_OK; }
    if (kind != hasher::POSEIDON2) return ctx->fail(QPGPU_EINVAL, "ctx_set_hasher: unknown hasher kind");
    poseidon2::Params p;
    if (parse_p2(params, n_words, p) != QPGPU_OK) return ctx->fail(QPGPU_EINVAL, "ctx_set_hasher: bad Poseidon2 parameter block");
    ctx->hasher.kind = hasher::POSEIDON2; ctx->hasher.p2 = p;
    if (ctx->d_p2) {
        QP_HIP(ctx, hipMemcpyAsync(ctx->d_p2, &ctx->hasher.p2, sizeof(poseidon2::Params), hipMemcpyHostToDevice, ctx->stream));
        QP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return QPGPU_OK;
}
extern "C" int qpgpu_ctx_get_hasher(const qpgpu_ctx *ctx) { return ctx ? ctx->hasher.kind : QPGPU_EINVAL; }

// digests: level 0 (n_leaves) then each parent level down to the cap level, concatenated; leaf.batch trees at once.
int merkle_build(qpgpu_ctx *ctx, const MerkleLeafArgs &leaf, unsigned log_leaves, unsigned cap_height, uint64_t *d_digests) {
    int rc = merkle_ensure_constants(ctx);
    if (rc) return rc;
    if (cap_height > log_leaves) return ctx->fail(QPGPU_EINVAL, "merkle: cap_height exceeds tree height");
    const HasherDev hd = ctx->hasher_dev();
    const uint32_t nb = leaf.batch ? leaf.batch : 1;
    ctx->prof_begin("merkle_leaf_hash");
    hipError_t e = merkle_leaf_hash(leaf, hd, ctx->stream);
    ctx->prof_end();
    QP_HIP(ctx, e);
    ctx->prof_begin("merkle_nodes");
    e = merkle_reduce_to_cap(d_digests, 1ull << log_leaves, 1ull << cap_height, nb, leaf.ps_digests, hd, ctx->stream);
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
    a.n_leaves = 1ull << log_leaves; a.digests = d_digests; a.batch = 1;
    ctx->hasher_in_use = true;
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
    ctx->hasher_in_use = true;
    const HasherDev hd = ctx->hasher_dev();
    QP_HIP(ctx, merkle_leaf_hash_rows(d_rows, cnt, width, d_digests, 1, 0, 0, hd, ctx->stream));
    QP_HIP(ctx, merkle_reduce_to_cap(d_digests, cnt, 1ull << cap_height, 1, 0, hd, ctx->stream));
    uint64_t *lvl = d_digests;
    while (cnt > (1ull << cap_height)) { lvl += cnt * 4; cnt >>= 1; }
    if (h_cap_out) {
        size_t cap_n = (size_t)1 << cap_height;
        QP_HIP(ctx, hipMemcpyAsync(h_cap_out, lvl, cap_n * 4 * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
        QP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return QPGPU_OK;
}

int qpgpu_poseidon2_hash_pad10_dev(qpgpu_ctx *ctx, const uint64_t *params, size_t n_words, const uint64_t *d_in, size_t len, size_t count, uint64_t *d_out) {
    if (!ctx) return QPGPU_EINVAL;
    QP_DEV(ctx);
    if ((!d_in && len && count) || (!d_out && count)) return ctx->fail(QPGPU_EINVAL, "poseidon2_hash_pad10: null buffer");
    if (len > (1u << 20)) return ctx->fail(QPGPU_EINVAL, "poseidon2_hash_pad10: preimage longer than 2^20 elements");
    if (!params && n_words == 0) {
        int rc = ctx->ensure_p2_app();
        if (rc) return rc;
        QP_HIP(ctx, poseidon2_hash_pad10_batch(d_in, len, count, d_out, ctx->d_p2_app, true, ctx->stream));
        return QPGPU_OK;
    }
    poseidon2::Params p;
    if (parse_p2(params, n_words, p) != QPGPU_OK) return ctx->fail(QPGPU_EINVAL, "poseidon2_hash_pad10: bad Poseidon2 parameter block");
    void *v = nullptr;   // a caller-supplied block: uploaded for this call only
    QP_HIP(ctx, hipMalloc(&v, sizeof p));
    hipError_t e = hipMemcpyAsync(v, &p, sizeof p, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = poseidon2_hash_pad10_batch(d_in, len, count, d_out, (const poseidon2::Params *)v, false, ctx->stream);
    const hipError_t e2 = hipStreamSynchronize(ctx->stream);   // `p` and the block outlive the kernel
    (void)hipFree(v);
    if (e != hipSuccess) return ctx->hip_fail(e, "poseidon2_hash_pad10");
    if (e2 != hipSuccess) return ctx->hip_fail(e2, "poseidon2_hash_pad10");
    return QPGPU_OK;
}

int qpgpu_poseidon_permute_dev(qpgpu_ctx *ctx, uint64_t *d_states, size_t n) {
    if (!ctx) return QPGPU_EINVAL;
    QP_DEV(ctx);
    if (!d_states && n) return ctx->fail(QPGPU_EINVAL, "poseidon: null buffer");
    int rc = merkle_ensure_constants(ctx);
    if (rc) return rc;
    ctx->hasher_in_use = true;
    QP_HIP(ctx, poseidon_permute_batch(d_states, n, ctx->hasher_dev(), ctx->stream));
    return QPGPU_OK;
}

}  // extern "C"
