// qpgpu_api.cpp — the extern "C" boundary declared in include/qpgpu.h.
#include <hip/hip_runtime.h>
#include <new>
#include "ctx.hpp"
#include "gl64.hpp"

void qpgpu_ctx::prof_begin(const char *name) {
    if (!profiling) return;
    Pending p; p.name = name;
    if (hipEventCreate(&p.e0) != hipSuccess || hipEventCreate(&p.e1) != hipSuccess) return;
    (void)hipEventRecord(p.e0, stream);
    prof_stack.push_back(pending.size());
    pending.push_back(p);
}
void qpgpu_ctx::prof_end() {
    if (!profiling || prof_stack.empty()) return;
    (void)hipEventRecord(pending[prof_stack.back()].e1, stream);
    prof_stack.pop_back();
}
int qpgpu_ctx::prof_collect() {
    QP_HIP(this, hipStreamSynchronize(stream));
    for (auto &p : pending) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, p.e0, p.e1) == hipSuccess) { kstats[p.name].ms += ms; kstats[p.name].launches++; }
        (void)hipEventDestroy(p.e0); (void)hipEventDestroy(p.e1);
    }
    pending.clear();
    prof_stack.clear();
    return QPGPU_OK;
}

extern "C" {

int qpgpu_profile_enable(qpgpu_ctx *ctx, int on) {
    if (!ctx) return QPGPU_EINVAL;
    QP_DEV(ctx);
    int rc = ctx->prof_collect();
    ctx->profiling = on != 0;
    if (on) ctx->kstats.clear();
    return rc;
}
int qpgpu_profile_read(qpgpu_ctx *ctx, const char *kernel, double *total_ms, uint64_t *launches) {
    if (!ctx || !kernel) return QPGPU_EINVAL;
    QP_DEV(ctx);
    int rc = ctx->prof_collect();
    if (rc) return rc;
    auto it = ctx->kstats.find(kernel);
    if (total_ms) *total_ms = it == ctx->kstats.end() ? 0.0 : it->second.ms;
    if (launches) *launches = it == ctx->kstats.end() ? 0 : it->second.launches;
    return QPGPU_OK;
}

const char *qpgpu_version(void) { return "qpgpu 0.1 (gfx950)"; }
int qpgpu_ctx_pci_bus_id(const qpgpu_ctx *ctx, char *out, size_t out_len) {
    if (!ctx || !out || out_len == 0) return QPGPU_EINVAL;
    out[0] = 0;
    if (out_len < 13) return QPGPU_EBUFSIZE;
    char id[32] = {0};
    if (hipDeviceGetPCIBusId(id, (int)sizeof id, ctx->device) != hipSuccess) return QPGPU_EDEVICE;
    size_t n = 0;
    while (id[n] && n + 1 < out_len) { const char ch = id[n]; out[n] = (ch >= 'A' && ch <= 'F') ? (char)(ch - 'A' + 'a') : ch; n++; }
    out[n] = 0;
    return QPGPU_OK;
}

int qpgpu_ctx_create(int device, qpgpu_ctx **out) {
    if (!out) return QPGPU_EINVAL;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) return QPGPU_EDEVICE;
    if (hipSetDevice(device) != hipSuccess) return QPGPU_EDEVICE;
    qpgpu_ctx *c = new (std::nothrow) qpgpu_ctx();
    if (!c) return QPGPU_ENOMEM;
    c->device = device;
    c->hasher = hasher::process_default();   // the context's proof-system hasher; qpgpu_ctx_set_hasher changes it
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return QPGPU_EDEVICE; }
    c->own_stream = true;
    *out = c;
    return QPGPU_OK;
}

void qpgpu_ctx_destroy(qpgpu_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (void *p : ctx->owned) (void)hipFree(p);
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    if (ctx->d_p2) (void)hipFree(ctx->d_p2);
    if (ctx->d_p2_app) (void)hipFree(ctx->d_p2_app);
    if (ctx->h_pin) (void)hipHostFree(ctx->h_pin);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char *qpgpu_last_error(const qpgpu_ctx *ctx) { return ctx ? ctx->err.c_str() : "null ctx"; }

int qpgpu_ctx_set_stream(qpgpu_ctx *ctx, void *hip_stream) {
    if (!ctx) return QPGPU_EINVAL;
    QP_DEV(ctx);
    QP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->own_stream) { QP_HIP(ctx, hipStreamDestroy(ctx->stream)); ctx->own_stream = false; }
    if (hip_stream) ctx->stream = (hipStream_t)hip_stream;
    else { QP_HIP(ctx, hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)); ctx->own_stream = true; }
    return QPGPU_OK;
}

int qpgpu_sync(qpgpu_ctx *ctx) {
    if (!ctx) return QPGPU_EINVAL;
    QP_DEV(ctx);
    QP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return QPGPU_OK;
}

int qpgpu_malloc(qpgpu_ctx *ctx, size_t bytes, void **dptr) {
    if (!ctx || !dptr) return QPGPU_EINVAL;
    QP_HIP(ctx, hipSetDevice(ctx->device));
    hipError_t e = hipMalloc(dptr, bytes ? bytes : 8);
    if (e == hipErrorOutOfMemory) return ctx->fail(QPGPU_ENOMEM, "hipMalloc: out of memory");
    if (e != hipSuccess) return ctx->hip_fail(e, "hipMalloc");
    return QPGPU_OK;
}
int qpgpu_free_scrubbed(qpgpu_ctx *ctx, void *dptr, size_t bytes) {
    if (!ctx) return QPGPU_EINVAL;
    QP_DEV(ctx);
    if (dptr && bytes) QP_HIP(ctx, hipMemsetAsync(dptr, 0, bytes, ctx->stream));
    QP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    QP_HIP(ctx, hipFree(dptr));
    return QPGPU_OK;
}
int qpgpu_free(qpgpu_ctx *ctx, void *dptr) {
    if (!ctx) return QPGPU_EINVAL;
    QP_DEV(ctx);
    QP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    QP_HIP(ctx, hipFree(dptr));
    return QPGPU_OK;
}
int qpgpu_memcpy_h2d(qpgpu_ctx *ctx, void *dst, const void *src, size_t bytes) {
    if (!ctx || (!dst && bytes) || (!src && bytes)) return QPGPU_EINVAL;
    QP_DEV(ctx);
    QP_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    QP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return QPGPU_OK;
}
int qpgpu_memcpy_d2h(qpgpu_ctx *ctx, void *dst, const void *src, size_t bytes) {
    if (!ctx || (!dst && bytes) || (!src && bytes)) return QPGPU_EINVAL;
    QP_DEV(ctx);
    QP_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    QP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return QPGPU_OK;
}

int qpgpu_memcpy_d2d(qpgpu_ctx *ctx, void *dst, const void *src, size_t bytes) {
    if (!ctx || (!dst && bytes) || (!src && bytes)) return QPGPU_EINVAL;
    QP_DEV(ctx);
    QP_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream));   // asynchronous on the ctx stream
    return QPGPU_OK;
}

int qpgpu_ntt_batch_dev(qpgpu_ctx *ctx, const uint64_t *d_in, uint64_t *d_out, unsigned log_n, size_t batch,
                        int flags, uint64_t coset_shift) {
    if (!ctx) return QPGPU_EINVAL;
    QP_DEV(ctx);
    if ((!d_in || !d_out) && batch) return ctx->fail(QPGPU_EINVAL, "ntt: null buffer");
    if (flags & ~3) return ctx->fail(QPGPU_EINVAL, "ntt: unknown flag");
    return ntt_run(ctx, d_in, d_out, log_n, log_n, batch, (flags & QPGPU_NTT_INVERSE) != 0,
                   (flags & QPGPU_NTT_OUT_BITREV) != 0, gl::canon(coset_shift));
}

int qpgpu_lde_batch_dev(qpgpu_ctx *ctx, const uint64_t *d_coeffs, uint64_t *d_out, unsigned log_n,
                        unsigned rate_bits, size_t batch, int flags, uint64_t coset_shift) {
    if (!ctx) return QPGPU_EINVAL;
    QP_DEV(ctx);
    if ((!d_coeffs || !d_out) && batch) return ctx->fail(QPGPU_EINVAL, "lde: null buffer");
    if (flags & ~QPGPU_NTT_OUT_BITREV) return ctx->fail(QPGPU_EINVAL, "lde: unknown flag");
    if (d_coeffs == d_out && rate_bits) return ctx->fail(QPGPU_EINVAL, "lde: in-place extension is not possible");
    return ntt_run(ctx, d_coeffs, d_out, log_n, log_n + rate_bits, batch, false,
                   (flags & QPGPU_NTT_OUT_BITREV) != 0, gl::canon(coset_shift));
}

int qpgpu_ntt_batch(qpgpu_ctx *ctx, uint64_t *data, unsigned log_n, size_t batch, int flags, uint64_t coset_shift) {
    if (!ctx) return QPGPU_EINVAL;
    QP_DEV(ctx);
    if (!data && batch) return ctx->fail(QPGPU_EINVAL, "ntt: null buffer");
    if (log_n > 40) return ctx->fail(QPGPU_EINVAL, "ntt: log_n out of range");
    size_t bytes = (batch << log_n) * sizeof(uint64_t);
    if (bytes == 0) return QPGPU_OK;
    void *d = nullptr;
    int rc = qpgpu_malloc(ctx, bytes, &d);
    if (rc) return rc;
    rc = qpgpu_memcpy_h2d(ctx, d, data, bytes);
    if (!rc) rc = qpgpu_ntt_batch_dev(ctx, (const uint64_t *)d, (uint64_t *)d, log_n, batch, flags, coset_shift);
    if (!rc) rc = qpgpu_memcpy_d2h(ctx, data, d, bytes);
    std::string keep = ctx->err;
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(d);
    if (rc) ctx->err = keep;
    return rc;
}

}  // extern "C"
