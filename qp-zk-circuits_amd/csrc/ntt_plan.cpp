// ntt_plan.cpp — host-side planning for the NTT passes: pass split, twiddle tables, launches.
// Replaces the root-table / dispatch logic of plonky2::field::fft (fft_root_table, fft_dispatch).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include "ctx.hpp"
#include "prover_kernels.hpp"
#include "gl64.hpp"
#include "ntt_pass.hpp"

using gl::u64;

#define QP_TRY_NTT(expr) do { int _rc = (expr); if (_rc) return _rc; } while (0)

struct NttTables {
    uint64_t *d = nullptr;  // device table
};

int qpgpu_ctx::ensure_scratch(size_t bytes) {
    if (bytes <= scratch_bytes) return QPGPU_OK;
    if (scratch) { QP_HIP(this, hipStreamSynchronize(stream)); QP_HIP(this, hipFree(scratch)); scratch = nullptr; scratch_bytes = 0; }
    QP_HIP(this, hipMalloc((void **)&scratch, bytes));
    scratch_bytes = bytes;
    return QPGPU_OK;
}

int qpgpu_ctx::reserve_read_back(size_t bytes) {
    if (bytes < (1u << 20)) bytes = 1u << 20;
    if (bytes > h_pin_bytes) {
        QP_HIP(this, hipStreamSynchronize(stream));
        if (h_pin) { (void)hipHostFree(h_pin); h_pin = nullptr; h_pin_bytes = 0; }
        QP_HIP(this, hipHostMalloc(&h_pin, bytes, hipHostMallocDefault));
        h_pin_bytes = bytes;
    }
    return QPGPU_OK;
}

int qpgpu_ctx::read_back(void *host_dst, const void *dev_src, size_t bytes) {
    if (bytes == 0) return QPGPU_OK;
    QP_TRY_NTT(reserve_read_back(bytes));
    QP_HIP(this, pk_copy(h_pin, dev_src, bytes, stream));
    QP_HIP(this, hipStreamSynchronize(stream));
    memcpy(host_dst, h_pin, bytes);
    return QPGPU_OK;
}

int qpgpu_ctx::read_back_2d(void *host_dst, const void *dev_src, size_t src_pitch, size_t width, size_t rows) {
    if (rows <= 1 || src_pitch == width) return read_back(host_dst, dev_src, width * rows);
    // the runtime splits a 2D device-to-host copy into one small copy per row (32 of them per stage for a lockstep batch of
    // 32): one kernel packs the rows straight into the pinned buffer
    const size_t bytes = width * rows;
    if ((width | src_pitch) & 7) return fail(QPGPU_EINVAL, "read_back_2d: row width and pitch must be multiples of 8 bytes");
    QP_TRY_NTT(reserve_read_back(bytes));
    QP_HIP(this, pk_pack_rows((const uint64_t *)dev_src, src_pitch / 8, width / 8, rows, (uint64_t *)h_pin, stream));
    QP_HIP(this, hipStreamSynchronize(stream));
    memcpy(host_dst, h_pin, bytes);
    return QPGPU_OK;
}

int qpgpu_ctx::upload(const std::vector<uint64_t> &host, uint64_t **dptr) {
    void *p = nullptr;
    QP_HIP(this, hipMalloc(&p, host.size() * sizeof(uint64_t)));
    owned.push_back(p);
    // tables are tiny; the copy is ordered on the context's own stream (never the legacy null stream, which every proving
    // thread of a process shares) and waited for, so the pageable source may go out of scope
    QP_HIP(this, hipMemcpyAsync(p, host.data(), host.size() * sizeof(uint64_t), hipMemcpyHostToDevice, stream));
    QP_HIP(this, hipStreamSynchronize(stream));
    *dptr = (uint64_t *)p;
    return QPGPU_OK;
}

namespace {

// powers table: out[i] = base^(i*stride_exp) for i < count
std::vector<uint64_t> powers(u64 base, u64 count) {
    std::vector<uint64_t> t(count);
    u64 acc = 1;
    for (u64 i = 0; i < count; i++) { t[i] = gl::canon(acc); acc = gl::mul(acc, base); }
    return t;
}

int cached(qpgpu_ctx *ctx, const std::string &key, u64 base, u64 count, uint64_t **out) {
    auto it = ctx->ntt_tables.find(key);
    if (it != ctx->ntt_tables.end()) { *out = it->second->d; return QPGPU_OK; }
    auto t = std::make_shared<NttTables>();
    int rc = ctx->upload(powers(base, count), &t->d);
    if (rc) return rc;
    ctx->ntt_tables[key] = t;
    *out = t->d;
    return QPGPU_OK;
}

struct Split { int ka, kb; };
Split split_round(int l) { Split s; s.ka = (l + 1) / 2; s.kb = l / 2; if (l == 1) { s.ka = 1; s.kb = 0; } return s; }

int env_int(const char *name, int dflt) { const char *e = getenv(name); return e && *e ? atoi(e) : dflt; }
int pick_log_t(int ka, int kb, u64 lanes_total) {
    static const int force = env_int("QPGPU_NTT_LOGT", -1);   // tuning knob for the 2^10-point passes
    if (force >= 0 && ka + kb == 10 && lanes_total >= (1ull << force)) return force;
    int log_t = 8 - ka;  // 256 threads per workgroup
    if (log_t < 3) log_t = 3;
    // 2^10-point passes with the split exchange: 16 lanes (full 128-byte segments on the strided pass), 512 threads, two
    // workgroups per CU = four wavefronts per SIMD (measured 1.48 -> 1.42 ms per 2^20 x 128 transform against 8 lanes)
    if (ka + kb == 10 && ntt_pass_uses_split(ka, kb) && lanes_total >= 16) log_t = 4;
    // keep LDS under ~72 KB so two workgroups share a CU
    while (log_t > 0 && ntt_pass_lds_bytes(ka, kb, log_t) > 72 * 1024) log_t--;
    (void)lanes_total;
    return log_t;
}

}  // namespace

namespace {
// Where a transform of at most 2^20 points reads and writes when it is one block of a larger one (ntt_run, L > 20)
struct Geom {
    u64 in_col_stride = 0, out_col_stride = 0;   // 0: the transform's own sizes
    u64 out_mul = 1;          // natural-order output: element k goes to k * out_mul
    u64 scale = 0;            // inverse: overrides 1/N
    u64 scratch_off = 0;      // words into ctx->scratch the natural-order intermediate may use
    uint32_t nproofs = 1;     // lockstep batch: proof b at d_in + b * in_ps, d_out + b * out_ps
    u64 in_ps = 0, out_ps = 0;
};
int ntt_core(qpgpu_ctx *ctx, const uint64_t *d_in, uint64_t *d_out, unsigned log_n_in, unsigned log_n_out,
             size_t batch, bool inverse, bool out_bitrev, uint64_t coset_shift, const Geom &g);
}  // namespace

// Forward / inverse / coset-LDE transform. log_n_in <= log_n_out; inputs beyond 2^log_n_in are zero.
int ntt_run(qpgpu_ctx *ctx, const uint64_t *d_in, uint64_t *d_out, unsigned log_n_in, unsigned log_n_out,
            size_t batch, bool inverse, bool out_bitrev, uint64_t coset_shift, NttProofs np) {
    if (log_n_out > 23) return ctx->fail(QPGPU_EINVAL, "ntt: log_n > 23 not supported");
    if (log_n_in > log_n_out) return ctx->fail(QPGPU_EINVAL, "ntt: log_n_in > log_n_out");
    if (batch == 0 || np.nproofs == 0) return QPGPU_OK;
    if (coset_shift > 1 && inverse) return ctx->fail(QPGPU_EINVAL, "ntt: coset inverse is ifft + scale; not a single call");
    // the column dimension rides on grid.y (<= 65535): wider batches go in slices (columns are contiguous, stride 2^log_n)
    if (batch > 65535) {
        if (np.nproofs > 1) return ctx->fail(QPGPU_EINVAL, "ntt: more than 65535 columns per proof");
        for (size_t c0 = 0; c0 < batch; c0 += 32768) {
            const size_t cnt = std::min<size_t>(32768, batch - c0);
            int rc = ntt_run(ctx, d_in + (c0 << log_n_in), d_out + (c0 << log_n_out), log_n_in, log_n_out, cnt, inverse, out_bitrev, coset_shift);
            if (rc) return rc;
        }
        return QPGPU_OK;
    }
    if (log_n_out <= 20) {
        Geom g;
        g.nproofs = np.nproofs; g.in_ps = np.in_ps; g.out_ps = np.out_ps;
        return ntt_core(ctx, d_in, d_out, log_n_in, log_n_out, batch, inverse, out_bitrev, coset_shift, g);
    }
    if (np.nproofs > 1) {   // the three-pass sizes take the proofs one after the other
        for (uint32_t b = 0; b < np.nproofs; b++) {
            int rc = ntt_run(ctx, d_in + b * np.in_ps, d_out + b * np.out_ps, log_n_in, log_n_out, batch, inverse, out_bitrev, coset_shift);
            if (rc) return rc;
        }
        return QPGPU_OK;
    }

    // ---- three passes: N = 8 * M0. Pass 0 transforms the top three index bits (stride M0) and applies the twiddle
    // w_N^(m k); the eight blocks are then independent M0-point transforms (two passes each) whose outputs interleave
    // (natural order: X[k0 + 8 k']) or land in bit-reversed block order (leaf order). ----
    const unsigned L = log_n_out, L0 = 3, Li = L - L0;
    const u64 N = 1ull << L, M0 = 1ull << Li, n_in = 1ull << log_n_in;
    if (log_n_in < Li) return ctx->fail(QPGPU_EINVAL, "lde: for 2^21..2^23 outputs the input must be at least 1/8 of the output");
    {
        static std::once_flag once;
        static hipError_t init_err = hipSuccess;
        std::call_once(once, [] { init_err = ntt_pass_init(); });
        QP_HIP(ctx, init_err);
    }
    const std::string dir = inverse ? "i" : "f";
    const u64 wN = inverse ? gl::inv(gl::root_of_unity(L)) : gl::root_of_unity(L);
    QP_TRY_NTT(ctx->ensure_scratch((batch * N + (out_bitrev ? 0 : batch * M0)) * sizeof(u64)));
    uint64_t *mid0 = ctx->scratch;
    {
        Split s = split_round((int)L0);
        NttPassArgs p{};
        p.inverse = inverse ? 1 : 0;
        p.ka = s.ka; p.kb = s.kb;
        p.in = d_in; p.out = mid0;
        p.in_col_stride = n_in; p.out_col_stride = N;
        p.log_m = Li; p.lanes_total = M0;
        p.in_row_stride = 0; p.in_l_stride = 1; p.in_p_stride = M0;
        p.out_row_stride = 0; p.out_l_stride = 1; p.out_p_stride = M0;
        p.log_t = pick_log_t(s.ka, s.kb, M0);
        p.p_valid = (uint32_t)(n_in / M0);
        p.load_lane_fast = 1; p.store_lane_fast = 1;
        p.out_bitrev = 0; p.has_out_scale = 0;
        QP_TRY_NTT(cached(ctx, "in" + dir + std::to_string(L0), gl::pow(wN, M0), 1ull << L0, (uint64_t **)&p.tw_inner));
        p.tw_lo_bits = (L + 1) / 2;
        QP_TRY_NTT(cached(ctx, "lo" + dir + std::to_string(L), wN, 1ull << p.tw_lo_bits, (uint64_t **)&p.tw_lo));
        QP_TRY_NTT(cached(ctx, "hi" + dir + std::to_string(L), gl::pow(wN, 1ull << p.tw_lo_bits), 1ull << (L - p.tw_lo_bits), (uint64_t **)&p.tw_hi));
        if (coset_shift > 1) {
            const std::string k = std::to_string(coset_shift) + "_" + std::to_string(Li);
            QP_TRY_NTT(cached(ctx, "csA" + k + "_" + std::to_string(p.p_valid), gl::pow(coset_shift, M0), p.p_valid, (uint64_t **)&p.in_scale_a));
            QP_TRY_NTT(cached(ctx, "csB" + k, coset_shift, M0, (uint64_t **)&p.in_scale_b));
        }
        const u64 tiles = (M0 + (1ull << p.log_t) - 1) >> p.log_t;
        ctx->prof_begin("ntt_pass_outer");
        hipError_t le = ctx->plan_only ? hipSuccess : ntt_pass_launch(p, tiles, batch, ctx->stream);
        ctx->prof_end();
        QP_HIP(ctx, le);
    }
    for (unsigned k0 = 0; k0 < (1u << L0); k0++) {
        Geom g;
        g.in_col_stride = N; g.out_col_stride = N;
        g.scale = inverse ? gl::inv(gl::canon(N % gl::P)) : 0;
        g.scratch_off = batch * N;
        const unsigned rk = ((k0 & 1) << 2) | (k0 & 2) | (k0 >> 2);
        uint64_t *out = out_bitrev ? d_out + (u64)rk * M0 : d_out + k0;
        if (!out_bitrev) g.out_mul = 1ull << L0;
        QP_TRY_NTT(ntt_core(ctx, mid0 + (u64)k0 * M0, out, Li, Li, batch, inverse, out_bitrev, 0, g));
    }
    return QPGPU_OK;
}

namespace {
int ntt_core(qpgpu_ctx *ctx, const uint64_t *d_in, uint64_t *d_out, unsigned log_n_in, unsigned log_n_out,
             size_t batch, bool inverse, bool out_bitrev, uint64_t coset_shift, const Geom &g) {
    const unsigned L = log_n_out;
    const u64 N = 1ull << L, n_in = 1ull << log_n_in;
    const bool coset = coset_shift > 1;
    if (L == 0) {
        if (d_in != d_out && !ctx->plan_only) QP_HIP(ctx, hipMemcpyAsync(d_out, d_in, batch * sizeof(u64), hipMemcpyDeviceToDevice, ctx->stream));
        return QPGPU_OK;
    }
    {   // one-time kernel attribute setup, safe when several proving threads start together
        static std::once_flag once;
        static hipError_t init_err = hipSuccess;
        std::call_once(once, [] { init_err = ntt_pass_init(); });
        QP_HIP(ctx, init_err);
    }

    const std::string dir = inverse ? "i" : "f";
    const u64 wN = inverse ? gl::inv(gl::root_of_unity(L)) : gl::root_of_unity(L);
    u64 out_scale = 1;
    if (inverse) out_scale = g.scale ? g.scale : gl::inv(gl::canon(N % gl::P));

    const int n_pass = (L <= 10) ? 1 : 2;
    const int L1 = (n_pass == 1) ? (int)L : (int)(L + 1) / 2;
    const int L2 = (int)L - L1;

    NttPassArgs a{};
    a.inverse = inverse ? 1 : 0;

    if (n_pass == 1) {
        Split s = split_round(L1);
        a.ka = s.ka; a.kb = s.kb;
        a.in = d_in; a.out = d_out;
        a.in_col_stride = 0; a.out_col_stride = 0;
        a.log_m = 0; a.lanes_total = batch;  // every polynomial is one row = one lane
        a.in_row_stride = n_in; a.in_l_stride = 0; a.in_p_stride = 1;
        a.out_row_stride = N; a.out_l_stride = 0; a.out_p_stride = 1;
        a.log_t = pick_log_t(s.ka, s.kb, batch);
        a.p_valid = (uint32_t)n_in;
        a.load_lane_fast = 0; a.store_lane_fast = 0;
        a.out_bitrev = out_bitrev ? 1 : 0;
        a.has_out_scale = inverse ? 1 : 0; a.out_scale = out_scale;
        if (s.kb > 0) {
            u64 wR = wN;  // R == N
            int rc = cached(ctx, "in" + dir + std::to_string(L1), wR, 1ull << L1, (uint64_t **)&a.tw_inner);
            if (rc) return rc;
        }
        if (coset) {
            int rc = cached(ctx, "csA" + std::to_string(coset_shift) + "_1_" + std::to_string(log_n_in), coset_shift, n_in, (uint64_t **)&a.in_scale_a);
            if (rc) return rc;
            rc = cached(ctx, "one", 1, 1, (uint64_t **)&a.in_scale_b);
            if (rc) return rc;
        }
        // when in == out and the row strides differ (LDE in place) the caller must not alias; same size is safe
        // because a workgroup reads its whole tile before writing it.
        u64 tiles = (batch + (1ull << a.log_t) - 1) >> a.log_t;
        a.in_proof_stride = g.in_ps; a.out_proof_stride = g.out_ps;
        ctx->prof_begin("ntt_pass_single");
        hipError_t le = ctx->plan_only ? hipSuccess : ntt_pass_launch(a, tiles, 1, ctx->stream, g.nproofs);
        ctx->prof_end();
        QP_HIP(ctx, le);
        return QPGPU_OK;
    }

    // ---- two passes: N = R1 * M1, pass 1 over the top L1 bits (stride M1), pass 2 on contiguous rows ----
    const u64 R1 = 1ull << L1, M1 = 1ull << L2;
    const u64 in_cs = g.in_col_stride ? g.in_col_stride : n_in, out_cs = g.out_col_stride ? g.out_col_stride : N;
    uint64_t *mid = d_out;
    bool folded_scale = false;
    u64 mid_cs = out_cs, mid_ps = g.out_ps;
    if (!out_bitrev) {
        int rc = ctx->ensure_scratch((g.scratch_off + (u64)g.nproofs * batch * N) * sizeof(u64));
        if (rc) return rc;
        mid = ctx->scratch + g.scratch_off; mid_cs = N; mid_ps = batch * N;
    }
    {
        Split s = split_round(L1);
        NttPassArgs p = a;
        p.ka = s.ka; p.kb = s.kb;
        p.in = d_in; p.out = mid;
        p.in_col_stride = in_cs; p.out_col_stride = mid_cs;
        p.log_m = L2; p.lanes_total = M1;
        p.in_row_stride = 0; p.in_l_stride = 1; p.in_p_stride = M1;
        p.out_row_stride = 0; p.out_l_stride = 1; p.out_p_stride = M1;
        p.log_t = pick_log_t(s.ka, s.kb, M1);
        { static const int f = env_int("QPGPU_NTT_LOGT_S", -1); if (f >= 0 && s.ka + s.kb == 10) p.log_t = f; }
        if ((1ull << p.log_t) > M1) p.log_t = L2;
        // zero padding: coefficient index n = p*M1 + mm < n_in  <=>  p < n_in / M1 (n_in >= M1 required)
        if (n_in < M1) return ctx->fail(QPGPU_EINVAL, "lde: input shorter than one pass-1 row is not supported");
        p.p_valid = (uint32_t)(n_in / M1);
        p.load_lane_fast = 1; p.store_lane_fast = 1;
        p.out_bitrev = out_bitrev ? 1 : 0;  // natural mode: row k1 in natural order so pass-2 lanes are adjacent k1
        p.has_out_scale = 0;
        int rc = cached(ctx, "in" + dir + std::to_string(L1), gl::pow(wN, M1), R1, (uint64_t **)&p.tw_inner);
        if (rc) return rc;
        // inter-pass twiddle w_N^(mm*k): e < N, split at lo_bits
        p.tw_lo_bits = (L + 1) / 2;
        // inter-pass twiddle of the strided pass: one read per element from the full table w_N^(mm k) (N words per transform size and
        // direction: 8 MB at 2^20, 512 KB at 2^16) and one product, instead of a running product per thread plus the element's
        // product (mode 1). Measured on the 2^20 x 128 transform, alternated in one call: strided launch 0.763 -> 0.715 ms, transform
        // 1.404 -> 1.354 ms, same outputs (profiles/r04_notes.txt item 6). QPGPU_NTT_TW=1 keeps the running product.
        { static const int tw_mode = env_int("QPGPU_NTT_TW", 3); p.tw_mode = (uint32_t)tw_mode; }
        p.out_loose = 1;      // the rows pass reduces whatever representative it is handed
        if (inverse && (p.tw_mode == 1 || p.tw_mode == 3)) p.tw_scale = out_scale;   // 1/N rides on the running twiddle product (mode 3: in the table)
        if (p.tw_mode == 3) {
            const std::string key = "full" + dir + std::to_string(L) + "_" + std::to_string(L2) + "_" + std::to_string(p.tw_scale);   // (the scale of an inner transform of a three-pass size is the outer 1/N)
            auto it = ctx->ntt_tables.find(key);
            if (it == ctx->ntt_tables.end()) {
                std::vector<u64> tab((size_t)R1 * M1);
                u64 wk = 1;                                      // wN^k
                for (u64 k = 0; k < R1; k++) {
                    u64 v = p.tw_scale ? p.tw_scale : 1;
                    for (u64 mm = 0; mm < M1; mm++) { tab[(size_t)k * M1 + mm] = gl::canon(v); v = gl::mul(v, wk); }
                    wk = gl::mul(wk, wN);
                }
                auto t = std::make_shared<NttTables>();
                rc = ctx->upload(tab, &t->d);
                if (rc) return rc;
                ctx->ntt_tables[key] = t;
                it = ctx->ntt_tables.find(key);
            }
            p.tw_full = it->second->d;
        }
        rc = cached(ctx, "lo" + dir + std::to_string(L), wN, 1ull << p.tw_lo_bits, (uint64_t **)&p.tw_lo);
        if (rc) return rc;
        rc = cached(ctx, "hi" + dir + std::to_string(L), gl::pow(wN, 1ull << p.tw_lo_bits), 1ull << (L - p.tw_lo_bits), (uint64_t **)&p.tw_hi);
        if (rc) return rc;
        if (coset) {
            const std::string k = std::to_string(coset_shift) + "_" + std::to_string(L2);
            rc = cached(ctx, "csA" + k + "_" + std::to_string(p.p_valid), gl::pow(coset_shift, M1), p.p_valid, (uint64_t **)&p.in_scale_a);
            if (rc) return rc;
            rc = cached(ctx, "csB" + k, coset_shift, M1, (uint64_t **)&p.in_scale_b);
            if (rc) return rc;
        }
        u64 tiles = (M1 + (1ull << p.log_t) - 1) >> p.log_t;
        folded_scale = p.tw_scale != 0;
        p.in_proof_stride = g.in_ps; p.out_proof_stride = mid_ps;
        ctx->prof_begin("ntt_pass_strided");
        hipError_t le = ctx->plan_only ? hipSuccess : ntt_pass_launch(p, tiles, batch, ctx->stream, g.nproofs);
        ctx->prof_end();
        QP_HIP(ctx, le);
    }
    {
        Split s = split_round(L2);
        NttPassArgs p = a;
        p.ka = s.ka; p.kb = s.kb;
        p.in = mid; p.out = d_out;
        p.in_col_stride = mid_cs; p.out_col_stride = out_cs;
        p.log_m = 0; p.lanes_total = R1;
        p.in_row_stride = M1; p.in_l_stride = 0; p.in_p_stride = 1;
        p.log_t = pick_log_t(s.ka, s.kb, R1);
        { static const int f = env_int("QPGPU_NTT_LOGT_R", -1); if (f >= 0 && s.ka + s.kb == 10) p.log_t = f; }
        p.p_valid = (uint32_t)M1;
        p.load_lane_fast = 0;
        if (out_bitrev) { p.out_row_stride = M1; p.out_p_stride = 1; p.store_lane_fast = 0; p.out_bitrev = 1; }
        else { p.out_row_stride = g.out_mul; p.out_p_stride = R1 * g.out_mul; p.store_lane_fast = 1; p.out_bitrev = 0; }
        p.out_l_stride = 0;
        p.has_out_scale = inverse && !folded_scale ? 1 : 0; p.out_scale = out_scale;
        if (s.kb > 0) {
            int rc = cached(ctx, "in" + dir + std::to_string(L2), gl::pow(wN, R1), M1, (uint64_t **)&p.tw_inner);
            if (rc) return rc;
        }
        u64 tiles = (R1 + (1ull << p.log_t) - 1) >> p.log_t;
        p.in_proof_stride = mid_ps; p.out_proof_stride = g.out_ps;
        ctx->prof_begin("ntt_pass_rows");
        hipError_t le = ctx->plan_only ? hipSuccess : ntt_pass_launch(p, tiles, batch, ctx->stream, g.nproofs);
        ctx->prof_end();
        QP_HIP(ctx, le);
    }
    return QPGPU_OK;
}
}  // namespace
