// oracle_api.cpp — stage-level C ABI: polynomial-batch commitments, opening evaluations, the challenger and the FRI
// opening proof as separate calls (include/qpgpu.h, "stage-level entry points"). These are the circuit-independent
// parts of qp-plonky2's prove(); a patched prover can keep its gate evaluation in Rust and use these for the rest.
#include <hip/hip_runtime.h>
#include <string>
#include "merkle.hpp"
#include "prover_host.hpp"
#include "prover_kernels.hpp"

using gl::e2;
using gl::u64;

struct qpgpu_oracle {
    qpgpu_ctx *ctx = nullptr;
    PolyOracle o;
    u64 *block = nullptr;     // one allocation: coeffs | lde | digests | salt | eval scratch | salt key
    size_t block_words = 0;
    e2 *d_point = nullptr, *d_eval = nullptr;
    uint32_t *d_key = nullptr;
};

extern "C" {

void qpgpu_oracle_free(qpgpu_oracle *h) {
    if (!h) return;
    (void)hipSetDevice(h->ctx->device);
    if (h->block) {
        // committed columns can hold the witness (reference wormhole/circuit/src/sensitive.rs:36-44): scrub before release
        (void)hipMemsetAsync(h->block, 0, h->block_words * 8, h->ctx->stream);
        (void)hipStreamSynchronize(h->ctx->stream);
        (void)hipFree(h->block);
    }
    delete h;
}

int qpgpu_oracle_commit(qpgpu_ctx *ctx, const uint64_t *polys, uint32_t num_polys, unsigned degree_bits, unsigned rate_bits,
                        unsigned cap_height, unsigned flags, uint64_t blinding_seed, uint32_t blinding_stream, qpgpu_oracle **out) {
    if (!ctx || !out) return QPGPU_EINVAL;
    QP_DEV(ctx);
    *out = nullptr;
    if (!polys || num_polys == 0) return ctx->fail(QPGPU_EINVAL, "oracle_commit: no polynomials");
    if (degree_bits + rate_bits > 23 || degree_bits < 1 || (degree_bits + rate_bits > 20 && rate_bits > 3))
        return ctx->fail(QPGPU_EINVAL, "oracle_commit: LDE sizes up to 2^23 (rate_bits <= 3 above 2^20) supported");
    if (cap_height > degree_bits + rate_bits) return ctx->fail(QPGPU_EINVAL, "oracle_commit: cap_height exceeds the tree height");
    if (flags & ~7u) return ctx->fail(QPGPU_EINVAL, "oracle_commit: unknown flags");
    ctx->hasher_in_use = true;
    QP_TRY(merkle_ensure_constants(ctx));
    const bool blinding = (flags & QPGPU_ORACLE_BLINDING) != 0, from_coeffs = (flags & QPGPU_ORACLE_COEFFS) != 0;
    const u64 n = 1ull << degree_bits, lde_n = n << rate_bits;
    const unsigned L = degree_bits + rate_bits;
    qpgpu_oracle *h = new qpgpu_oracle();
    h->ctx = ctx;
    PolyOracle &o = h->o;
    o.ncols = num_polys; o.log_n = degree_bits; o.rate_bits = rate_bits; o.cap_h = cap_height; o.oracle_index = blinding_stream;
    const size_t w_coeffs = (size_t)num_polys * n, w_lde = (size_t)num_polys * lde_n, w_dig = digest_words(L, cap_height),
                 w_salt = blinding ? (size_t)4 * lde_n : 0, w_eval = 2 + 2 * (size_t)num_polys;
    h->block_words = w_coeffs + w_lde + w_dig + w_salt + w_eval + 4;
    void *v = nullptr;
    hipError_t e = hipMalloc(&v, h->block_words * 8);
    if (e != hipSuccess) { delete h; return ctx->hip_fail(e, "hipMalloc(oracle)"); }
    h->block = (u64 *)v;
    o.coeffs = h->block; o.lde = o.coeffs + w_coeffs; o.digests = o.lde + w_lde;
    o.salt = blinding ? o.digests + w_dig : nullptr;
    h->d_point = (e2 *)(o.digests + w_dig + w_salt); h->d_eval = h->d_point + 1;
    h->d_key = (uint32_t *)(o.digests + w_dig + w_salt + w_eval);
    o.set_batch(1, true);
    int rc = QPGPU_OK;
    if (blinding) {
        // blinding_seed 0: 256 fresh bits from the OS entropy source (the reference: thread_rng); otherwise the key derived from
        // the seed, which makes the commitment reproducible (tests)
        uint32_t key[8];
        if (blinding_seed) salt_key_from_seed(blinding_seed, key);
        else if (salt_key_random(key) != QPGPU_OK) rc = ctx->fail(QPGPU_EDEVICE, "oracle_commit: the OS entropy source failed");
        if (rc == QPGPU_OK) {
            e = hipMemcpyAsync(h->d_key, key, sizeof key, hipMemcpyHostToDevice, ctx->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
            if (e != hipSuccess) rc = ctx->hip_fail(e, "oracle_commit: key upload");
        }
    }
    const u64 *src = polys;
    if (!(flags & QPGPU_ORACLE_DEVICE_INPUT)) {
        // stage through the LDE area (large enough, overwritten afterwards); a coefficient input goes straight to its place
        u64 *dst = from_coeffs ? o.coeffs : o.lde;
        e = hipMemcpyAsync(dst, polys, w_coeffs * 8, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) { rc = ctx->hip_fail(e, "oracle_commit: upload"); }
        src = dst;
    } else if (from_coeffs) {
        e = hipMemcpyAsync(o.coeffs, polys, w_coeffs * 8, hipMemcpyDeviceToDevice, ctx->stream);
        if (e != hipSuccess) rc = ctx->hip_fail(e, "oracle_commit: copy");
    }
    if (rc == QPGPU_OK) rc = from_coeffs ? oracle_commit_coeffs(ctx, o, h->d_key) : oracle_commit_values(ctx, src, o, h->d_key);
    if (rc != QPGPU_OK) { qpgpu_oracle_free(h); return rc; }
    *out = h;
    return QPGPU_OK;
}

int qpgpu_oracle_cap(const qpgpu_oracle *h, uint64_t *out, size_t out_words) {
    if (!h || !out || out_words < h->o.cap.size()) return QPGPU_EINVAL;
    std::memcpy(out, h->o.cap.data(), h->o.cap.size() * 8);
    return QPGPU_OK;
}

int qpgpu_oracle_eval(qpgpu_oracle *h, const uint64_t point[2], uint32_t first, uint32_t count, uint64_t *out) {
    if (!h) return QPGPU_EINVAL;
    qpgpu_ctx *ctx = h->ctx;
    QP_DEV(ctx);
    if (!point || !out || count == 0 || (size_t)first + count > h->o.ncols) return ctx->fail(QPGPU_EINVAL, "oracle_eval: bad range");
    const e2 z = gl::e2_make(gl::canon(point[0]), gl::canon(point[1]));
    QP_HIP(ctx, hipMemcpyAsync(h->d_point, &z, sizeof z, hipMemcpyHostToDevice, ctx->stream));
    QP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const u64 n = 1ull << h->o.log_n;
    QP_HIP(ctx, pk_poly_eval(h->o.coeffs + (size_t)first * n, n, count, h->d_point, 1, h->d_eval, 1, 0, 0, 0, ctx->stream));
    QP_HIP(ctx, hipMemcpyAsync(out, h->d_eval, (size_t)count * sizeof(e2), hipMemcpyDeviceToHost, ctx->stream));
    QP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return QPGPU_OK;
}

int qpgpu_oracle_read(qpgpu_oracle *h, unsigned what, uint32_t first, uint32_t count, uint64_t *out) {
    if (!h) return QPGPU_EINVAL;
    qpgpu_ctx *ctx = h->ctx;
    QP_DEV(ctx);
    if (!out || count == 0 || (size_t)first + count > h->o.ncols || what > QPGPU_ORACLE_READ_LDE) return ctx->fail(QPGPU_EINVAL, "oracle_read: bad range");
    const u64 len = what == QPGPU_ORACLE_READ_LDE ? h->o.lde_n() : 1ull << h->o.log_n;
    const u64 *src = (what == QPGPU_ORACLE_READ_LDE ? h->o.lde : h->o.coeffs) + (size_t)first * len;
    QP_HIP(ctx, hipMemcpyAsync(out, src, (size_t)count * len * 8, hipMemcpyDeviceToHost, ctx->stream));
    QP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return QPGPU_OK;
}

int qpgpu_oracle_device_ptrs(const qpgpu_oracle *h, const uint64_t **d_coeffs, const uint64_t **d_lde, const uint64_t **d_digests) {
    if (!h) return QPGPU_EINVAL;
    if (d_coeffs) *d_coeffs = h->o.coeffs;
    if (d_lde) *d_lde = h->o.lde;
    if (d_digests) *d_digests = h->o.digests;
    return QPGPU_OK;
}

// ---- challenger (host only) ----
// plonky2 never holds a full input buffer (a duplex runs when the eighth element arrives) nor more than eight outputs: a
// state that says otherwise is invalid input
static bool to_host(const qpgpu_challenger *c, Challenger &ch) {
    if (c->input_len >= 8 || c->output_len > 8) return false;
    std::memcpy(ch.state, c->sponge_state, sizeof ch.state);
    ch.n_in = (int)c->input_len; ch.n_out = (int)c->output_len;
    std::memcpy(ch.in, c->input_buffer, sizeof ch.in);
    std::memcpy(ch.out, c->output_buffer, sizeof ch.out);
    return true;
}
static void from_host(const Challenger &ch, qpgpu_challenger *c) {
    std::memcpy(c->sponge_state, ch.state, sizeof ch.state);
    std::memset(c->input_buffer, 0, sizeof c->input_buffer); std::memset(c->output_buffer, 0, sizeof c->output_buffer);
    for (int i = 0; i < ch.n_in; i++) c->input_buffer[i] = ch.in[i];
    for (int i = 0; i < ch.n_out; i++) c->output_buffer[i] = ch.out[i];
    c->input_len = (uint32_t)ch.n_in; c->output_len = (uint32_t)ch.n_out;
}
void qpgpu_challenger_init(qpgpu_challenger *c) { if (c) std::memset(c, 0, sizeof *c); }
static int challenger_observe(const hasher::Config &h, qpgpu_challenger *c, const uint64_t *elements, size_t n) {
    if (!c || (!elements && n)) return QPGPU_EINVAL;
    Challenger ch(h);
    if (!to_host(c, ch)) return QPGPU_EINVAL;
    ch.observe(elements, n);
    from_host(ch, c);
    return QPGPU_OK;
}
static int challenger_get(const hasher::Config &h, qpgpu_challenger *c, uint64_t *out) {
    if (!c || !out) return QPGPU_EINVAL;
    Challenger ch(h);
    if (!to_host(c, ch)) return QPGPU_EINVAL;
    *out = ch.get();
    from_host(ch, c);
    return QPGPU_OK;
}
// under the context's hasher
int qpgpu_ctx_challenger_observe(const qpgpu_ctx *ctx, qpgpu_challenger *c, const uint64_t *elements, size_t n) {
    return ctx ? challenger_observe(ctx->hasher, c, elements, n) : QPGPU_EINVAL;
}
int qpgpu_ctx_challenger_get(const qpgpu_ctx *ctx, qpgpu_challenger *c, uint64_t *out) {
    return ctx ? challenger_get(ctx->hasher, c, out) : QPGPU_EINVAL;
}
// under the process-default hasher (first ABI; an invalid state leaves the challenger untouched and get returns 0)
void qpgpu_challenger_observe(qpgpu_challenger *c, const uint64_t *elements, size_t n) { (void)challenger_observe(hasher::process_default(), c, elements, n); }
uint64_t qpgpu_challenger_get(qpgpu_challenger *c) {
    uint64_t v = 0;
    (void)challenger_get(hasher::process_default(), c, &v);
    return v;
}

// ---- FRI ----
static int fri_setup(qpgpu_oracle *const *oracles, uint32_t n, const qpgpu_fri_params *params, FriParams &fp, std::vector<size_t> &widths, size_t &total_polys) {
    if (!oracles || n == 0 || !params || params->num_reduction_rounds > 16) return QPGPU_EINVAL;
    for (uint32_t i = 0; i < n; i++) if (!oracles[i] || oracles[i]->ctx != oracles[0]->ctx) return QPGPU_EINVAL;
    fp.degree_bits = oracles[0]->o.log_n; fp.rate_bits = params->rate_bits; fp.cap_h = params->cap_height;
    fp.pow_bits = params->proof_of_work_bits; fp.num_queries = params->num_query_rounds;
    fp.arity_bits.assign(params->reduction_arity_bits, params->reduction_arity_bits + params->num_reduction_rounds);
    unsigned tot = 0;
    for (unsigned a : fp.arity_bits) { if (a == 0 || a > 8) return QPGPU_EINVAL; tot += a; }
    if (tot > fp.degree_bits || fp.degree_bits + fp.rate_bits < tot + fp.cap_h || fp.pow_bits > 40 || fp.num_queries == 0 || fp.num_queries > 4096) return QPGPU_EINVAL;
    widths.clear(); total_polys = 0;
    for (uint32_t i = 0; i < n; i++) { widths.push_back(oracles[i]->o.ncols + (oracles[i]->o.salt ? 4 : 0)); total_polys += oracles[i]->o.ncols; }
    return QPGPU_OK;
}

size_t qpgpu_fri_proof_size(qpgpu_oracle *const *oracles, uint32_t num_oracles, const qpgpu_fri_params *params) {
    FriParams fp; std::vector<size_t> widths; size_t total = 0;
    if (fri_setup(oracles, num_oracles, params, fp, widths, total) != QPGPU_OK) return 0;
    return fri_proof_bytes(fp, widths);
}

int qpgpu_fri_prove(qpgpu_ctx *ctx, qpgpu_oracle *const *oracles, uint32_t num_oracles, const qpgpu_fri_batch *batches,
                    uint32_t num_batches, const qpgpu_fri_params *params, qpgpu_challenger *challenger,
                    uint8_t *out, size_t out_cap, size_t *out_len) {
    if (!ctx) return QPGPU_EINVAL;
    QP_DEV(ctx);
    if (!batches || num_batches == 0 || !challenger || !out) return ctx->fail(QPGPU_EINVAL, "fri_prove: null argument");
    FriParams fp; std::vector<size_t> widths; size_t total_polys = 0;
    if (fri_setup(oracles, num_oracles, params, fp, widths, total_polys) != QPGPU_OK || oracles[0]->ctx != ctx)
        return ctx->fail(QPGPU_EINVAL, "fri_prove: bad oracles or FRI parameters");
    if (challenger->input_len >= 8 || challenger->output_len > 8) return ctx->fail(QPGPU_EINVAL, "fri_prove: invalid challenger state (input_len must be < 8, output_len <= 8)");
    std::vector<FriBatch> bs(num_batches);
    size_t max_count = 0;
    for (uint32_t b = 0; b < num_batches; b++) {
        if (batches[b].num_ranges == 0 || batches[b].num_ranges > 8) return ctx->fail(QPGPU_EINVAL, "fri_prove: a batch needs 1..8 polynomial ranges");
        bs[b].points = {gl::e2_make(gl::canon(batches[b].point[0]), gl::canon(batches[b].point[1]))};
        size_t cnt = 0;
        for (uint32_t r = 0; r < batches[b].num_ranges; r++) {
            const qpgpu_fri_range &rg = batches[b].ranges[r];
            bs[b].ranges.push_back({rg.oracle, rg.first, rg.count});
            cnt += rg.count;
        }
        max_count = std::max(max_count, cnt);
    }
    std::vector<const PolyOracle *> os;
    for (uint32_t i = 0; i < num_oracles; i++) os.push_back(&oracles[i]->o);
    // workspace: one device allocation and one pinned staging block per call
    void *dv = nullptr, *hv = nullptr;
    const size_t work_words = FriWork::words(fp, widths, max_count, 1);
    QP_HIP(ctx, hipMalloc(&dv, work_words * 8));
    Stager stage;
    stage.words = FriWork::stage_words(fp, max_count, 1);
    hipError_t e = hipHostMalloc(&hv, stage.words * 8, hipHostMallocDefault);
    if (e != hipSuccess) { (void)hipFree(dv); return ctx->hip_fail(e, "hipHostMalloc(fri stage)"); }
    stage.h = (u64 *)hv;
    FriWork work;
    work.bind((u64 *)dv, fp, widths, max_count, 1);
    Challenger ch(ctx->hasher); (void)to_host(challenger, ch);
    ByteWriter w{out, out_cap};
    int rc = fri_prove(ctx, fp, os.data(), os.size(), bs, 1, &ch, work, stage, &w);
    // the workspace holds alpha-combinations of the committed (possibly secret) polynomials: clear it before release
    (void)hipMemsetAsync(dv, 0, work_words * 8, ctx->stream);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(dv); (void)hipHostFree(hv);
    if (rc != QPGPU_OK) return rc;
    from_host(ch, challenger);
    if (out_len) *out_len = w.len;
    if (w.overflow) return ctx->fail(QPGPU_EBUFSIZE, "fri_prove: output buffer too small");
    return QPGPU_OK;
}

}  // extern "C"
