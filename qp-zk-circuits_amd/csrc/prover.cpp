// prover.cpp — host orchestration of plonky2's prove() on the GPU (stages s2..s12) and the C ABI for it.
//
// Replaces qp-plonky2 1.5.5 `plonk::prover::prove` after witness generation, i.e. what the reference reaches
// through `self.circuit_data.prove(self.partial_witness)` (wormhole/prover/src/lib.rs:171-175; aggregator call
// sites in include/qpgpu.h). Transcript order (Fiat-Shamir) is plonky2's: circuit digest, public-input hash,
// wires cap -> betas, gammas -> Z/partial-products cap -> alphas -> quotient cap -> zeta -> openings -> FRI alpha ->
// per-round cap, beta -> final polynomial -> proof-of-work nonce -> query indices. The challenger is a few hundred
// permutations per proof and stays on the host; everything that scales with the trace runs in HIP kernels.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstring>
#include <random>
#include <string>
#include <vector>
#include "circuit.hpp"
#include "ctx.hpp"
#include "gl64.hpp"
#include "merkle.hpp"
#include "poseidon.hpp"
#include "prover_kernels.hpp"

using gl::e2;
using gl::u64;

namespace {

// ---- host duplex challenger (plonky2::iop::challenger::Challenger) ----
struct Challenger {
    u64 state[12] = {0};
    u64 in[8]; int n_in = 0;
    u64 out[8]; int n_out = 0;
    void duplex() {
        for (int i = 0; i < n_in; i++) state[i] = in[i];
        n_in = 0;
        poseidon::permute(state, poseidon::host_round_constants());
        std::memcpy(out, state, sizeof out);
        n_out = 8;
    }
    void observe(const u64 *x, size_t n) {
        for (size_t i = 0; i < n; i++) { n_out = 0; in[n_in++] = gl::canon(x[i]); if (n_in == 8) duplex(); }
    }
    u64 get() { if (n_in > 0 || n_out == 0) duplex(); return out[--n_out]; }
    e2 get_ext() { u64 a = get(), b = get(); return gl::e2_make(a, b); }
};

void host_hash_no_pad(const u64 *in, size_t n, u64 out[4]) {
    u64 st[12] = {0};
    for (size_t i = 0; i < n; i += 8) {
        size_t len = std::min<size_t>(8, n - i);
        for (size_t k = 0; k < len; k++) st[k] = gl::canon(in[i + k]);
        poseidon::permute(st, poseidon::host_round_constants());
    }
    std::memcpy(out, st, 32);
}

std::vector<u64> powers_table(u64 base, u64 count) {
    std::vector<u64> t(count);
    u64 a = 1;
    for (u64 i = 0; i < count; i++) { t[i] = gl::canon(a); a = gl::mul(a, base); }
    return t;
}

struct DevBatch {
    uint32_t ncols = 0;
    unsigned log_n = 0;
    u64 *coeffs = nullptr, *lde = nullptr, *digests = nullptr;
    u64 *salt = nullptr;            // [4][lde_n] when the oracle is blinded
    uint32_t oracle_index = 0;
    std::vector<u64> cap;
};

}  // namespace

struct qpgpu_circuit {
    qpgpu_ctx *ctx = nullptr;
    CircuitPack pack;
    std::vector<void *> allocs;
    // setup-time residents
    u64 *d_cs_values = nullptr;
    DevBatch cs;
    GateDev *d_gates = nullptr;
    std::vector<GateDev> h_gates;
    u64 *d_qacc = nullptr;
    u64 *d_poseidon_rc = nullptr, *d_poseidon_fast = nullptr;
    u64 *d_omega = nullptr, *d_x_coset = nullptr, *d_l0_coset = nullptr, *d_zh_inv = nullptr;
    u64 *d_ginv_lo = nullptr, *d_ginv_hi = nullptr; uint32_t ginv_lo_bits = 0;
    // per-proof workspace
    u64 *d_wires_vals = nullptr;
    DevBatch wires, zs, quot;
    u64 *d_qcp = nullptr, *d_rowprod = nullptr, *d_z = nullptr, *d_zs_vals = nullptr;
    u64 *d_small = nullptr;          // betas, gammas, beta_k_is, alpha pows, pi hash
    e2 *d_points = nullptr, *d_open = nullptr, *d_alpha_ext = nullptr;
    u64 *d_comp = nullptr, *d_fin = nullptr;     // [2][n] each
    u64 *d_fri_vals = nullptr, *d_fri_rows = nullptr, *d_fri_coeffs[2] = {nullptr, nullptr};
    std::vector<u64 *> d_fri_digests, d_fri_leafrows;
    u64 *d_pow = nullptr, *d_qidx = nullptr, *d_gather = nullptr;
    size_t gather_words = 0;
    u64 *h_stage = nullptr;          // pinned host staging for the small per-proof tables (no sync on upload)
    size_t stage_words = 0, stage_pos = 0;
    bool seed_set = false;
    bool check_witness = false;
    u64 *d_check = nullptr;          // [2]: first bad row, permutation flag
    u64 blinding_seed = 0;

    template <class T> int alloc(T **p, size_t count) {
        void *v = nullptr;
        hipError_t e = hipMalloc(&v, std::max<size_t>(count * sizeof(T), 8));
        if (e != hipSuccess) return ctx->hip_fail(e, "hipMalloc(circuit)");
        allocs.push_back(v);
        *p = (T *)v;
        return QPGPU_OK;
    }
};

namespace {

size_t digest_words(unsigned log_leaves, unsigned cap_h) { return ((2ull << log_leaves) - (1ull << cap_h)) * 4; }

#define QP_TRY(expr) do { int _rc = (expr); if (_rc) return _rc; } while (0)

int alloc_batch(qpgpu_circuit *c, DevBatch &b, uint32_t ncols, bool need_coeffs = true, uint32_t oracle_index = 0) {
    b.oracle_index = oracle_index;
    if (oracle_index > 0 && c->pack.zero_knowledge) QP_TRY(c->alloc(&b.salt, (size_t)4 << (c->pack.degree_bits + c->pack.rate_bits)));
    const CircuitPack &p = c->pack;
    const u64 n = p.n(), lde_n = n << p.rate_bits;
    b.ncols = ncols; b.log_n = (unsigned)p.degree_bits;
    if (need_coeffs) QP_TRY(c->alloc(&b.coeffs, (size_t)ncols * n));
    QP_TRY(c->alloc(&b.lde, (size_t)ncols * lde_n));
    QP_TRY(c->alloc(&b.digests, digest_words((unsigned)(p.degree_bits + p.rate_bits), (unsigned)p.cap_height)));
    b.cap.resize((1ull << p.cap_height) * 4);
    return QPGPU_OK;
}

// PolynomialBatch::from_coeffs: LDE on the coset g<w>, leaf order, Merkle tree, cap to the host (syncs)
int commit_coeffs(qpgpu_circuit *c, DevBatch &b) {
    qpgpu_ctx *ctx = c->ctx;
    const CircuitPack &p = c->pack;
    const unsigned L = (unsigned)(p.degree_bits + p.rate_bits);
    QP_TRY(ntt_run(ctx, b.coeffs, b.lde, b.log_n, L, b.ncols, false, true, gl::MULT_GEN));
    MerkleLeafArgs a{};
    a.src0 = b.lde; a.stride0 = 1ull << L; a.ncols0 = b.ncols; a.n_leaves = 1ull << L; a.digests = b.digests;
    if (b.salt) {
        QP_HIP(ctx, pk_salt(c->blinding_seed, b.oracle_index, 1ull << L, b.salt, ctx->stream));
        a.src1 = b.salt; a.stride1 = 1ull << L; a.ncols1 = 4;
    }
    QP_TRY(merkle_build(ctx, a, L, (unsigned)p.cap_height, b.digests));
    const size_t total = digest_words(L, (unsigned)p.cap_height);
    QP_HIP(ctx, hipMemcpyAsync(b.cap.data(), b.digests + total - b.cap.size(), b.cap.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
    QP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return QPGPU_OK;
}
// PolynomialBatch::from_values
int commit_values(qpgpu_circuit *c, const u64 *d_values, DevBatch &b) {
    QP_TRY(ntt_run(c->ctx, d_values, b.coeffs, b.log_n, b.log_n, b.ncols, true, false, 0));
    return commit_coeffs(c, b);
}

// Upload a small table through the circuit's pinned staging area: asynchronous, no stream sync. Each proof uses a
// fresh region per table; regions are recycled at the start of the next proof (the previous one has completed).
int h2d_staged(qpgpu_circuit *c, void *dst, const void *src, size_t bytes) {
    const size_t words = (bytes + 7) / 8;
    if (c->stage_pos + words > c->stage_words) {   // should not happen: sized at load time
        QP_HIP(c->ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->ctx->stream));
        QP_HIP(c->ctx, hipStreamSynchronize(c->ctx->stream));
        return QPGPU_OK;
    }
    u64 *slot = c->h_stage + c->stage_pos;
    c->stage_pos += words;
    std::memcpy(slot, src, bytes);
    QP_HIP(c->ctx, hipMemcpyAsync(dst, slot, bytes, hipMemcpyHostToDevice, c->ctx->stream));
    return QPGPU_OK;
}

int h2d(qpgpu_ctx *ctx, void *dst, const void *src, size_t bytes) {
    QP_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    QP_HIP(ctx, hipStreamSynchronize(ctx->stream));   // source is pageable and may go out of scope
    return QPGPU_OK;
}

struct ByteWriter {
    uint8_t *p; size_t cap, len = 0; bool overflow = false;
    void u64le(u64 v) { if (len + 8 > cap) { overflow = true; len += 8; return; } std::memcpy(p + len, &v, 8); len += 8; }
    void u8(uint8_t v) { if (len + 1 > cap) { overflow = true; len += 1; return; } p[len++] = v; }
    void vec(const u64 *v, size_t n) { for (size_t i = 0; i < n; i++) u64le(v[i]); }
    void ext(e2 v) { u64le(v.a); u64le(v.b); }
};

}  // namespace

extern "C" {

size_t qpgpu_proof_size(const qpgpu_circuit *c) {
    if (!c) return 0;
    const CircuitPack &p = c->pack;
    const size_t ncs = p.num_cs_cols(), nch = p.num_challenges, cap = (1ull << p.cap_height) * 32;
    const size_t L = p.degree_bits + p.rate_bits;
    size_t sz = 3 * cap + (ncs + p.num_wires + nch * 2 + nch * p.num_partial_products + p.num_quotient_cols()) * 16;
    const size_t salt = p.zero_knowledge ? 4 : 0;
    const size_t widths[4] = {ncs, (size_t)p.num_wires + salt, (size_t)p.num_zs_pp_cols() + salt, (size_t)p.num_quotient_cols() + salt};
    size_t q = 0, lvl = L, fin = p.degree_bits;
    for (size_t w : widths) q += w * 8 + 1 + (L - p.cap_height) * 32;
    for (u64 a : p.arity_bits) { sz += cap; lvl -= a; fin -= a; q += (16ull << a) + 1 + (lvl - p.cap_height) * 32; }
    return sz + p.num_query_rounds * q + (16ull << fin) + 8 + p.num_public_inputs * 8;
}

void qpgpu_circuit_free(qpgpu_circuit *c) {
    if (!c) return;
    (void)hipSetDevice(c->ctx->device);
    (void)hipStreamSynchronize(c->ctx->stream);
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    for (void *p : c->allocs) (void)hipFree(p);
    delete c;
}

int qpgpu_circuit_load(qpgpu_ctx *ctx, const uint64_t *pack_words, size_t n_words, qpgpu_circuit **out) {
    if (!ctx || !out) return QPGPU_EINVAL;
    QP_DEV(ctx);
    *out = nullptr;
    if (!pack_words) return ctx->fail(QPGPU_EINVAL, "circuit_load: null pack");
    qpgpu_circuit *c = new qpgpu_circuit();
    c->ctx = ctx;
    std::string err = c->pack.parse(pack_words, n_words);
    if (!err.empty()) { delete c; return ctx->fail(QPGPU_EINVAL, "circuit_load: " + err); }
    const CircuitPack &p = c->pack;
    if (p.num_chunks() > 16 || p.num_challenges > 4) { delete c; return ctx->fail(QPGPU_EINVAL, "circuit_load: too many chunks/challenges"); }
    if (p.degree_bits + p.rate_bits > 20) { delete c; return ctx->fail(QPGPU_EINVAL, "circuit_load: LDE larger than 2^20 not supported yet"); }
    const u64 n = p.n(), lde_n = n << p.rate_bits, R = p.num_routed_wires, nch = p.num_challenges;
    const unsigned d = (unsigned)p.degree_bits, L = (unsigned)(p.degree_bits + p.rate_bits);
    int rc = QPGPU_OK;
    auto fail = [&](int code) { qpgpu_circuit_free(c); return code; };
#define CK(expr) do { rc = (expr); if (rc) return fail(rc); } while (0)
    CK(merkle_ensure_constants(ctx));
    // constants / sigmas: values (for the sigma columns of s5) and the setup commitment
    CK(c->alloc(&c->d_cs_values, (size_t)p.num_cs_cols() * n));
    CK(h2d(ctx, c->d_cs_values, p.constants_sigmas.data(), p.constants_sigmas.size() * 8));
    CK(alloc_batch(c, c->cs, (uint32_t)p.num_cs_cols()));
    CK(commit_values(c, c->d_cs_values, c->cs));
    // gates
    std::vector<GateDev> gd(p.gates.size());
    for (size_t i = 0; i < gd.size(); i++) {
        const GateInfo &g = p.gates[i];
        gd[i] = {(uint32_t)g.type, (uint32_t)g.param0, (uint32_t)g.param1, (uint32_t)g.selector_index, (uint32_t)g.group_start,
                 (uint32_t)g.group_end, (uint32_t)g.num_constraints, 0};
    }
    c->h_gates = gd;
    CK(c->alloc(&c->d_gates, gd.size()));
    CK(h2d(ctx, c->d_gates, gd.data(), gd.size() * sizeof(GateDev)));
    CK(c->alloc(&c->d_poseidon_rc, 360));
    CK(h2d(ctx, c->d_poseidon_rc, poseidon::host_round_constants(), 360 * 8));
    CK(c->alloc(&c->d_poseidon_fast, poseidon::FP_WORDS));
    CK(h2d(ctx, c->d_poseidon_fast, poseidon::host_fast_partial(), poseidon::FP_WORDS * 8));
    // subgroup, coset and vanishing tables
    CK(c->alloc(&c->d_omega, n));
    { auto t = powers_table(gl::root_of_unity(d), n); CK(h2d(ctx, c->d_omega, t.data(), n * 8)); }
    const uint32_t rate = 1u << p.rate_bits;
    std::vector<u64> zh(rate), zh_inv(rate);
    { u64 gn = gl::pow(gl::MULT_GEN, n), wr = gl::root_of_unity((unsigned)p.rate_bits), a = 1;
      for (uint32_t i = 0; i < rate; i++) { zh[i] = gl::canon(gl::sub(gl::mul(gn, a), 1)); zh_inv[i] = gl::inv(zh[i]); a = gl::mul(a, wr); } }
    u64 *d_zh = nullptr;
    CK(c->alloc(&d_zh, rate)); CK(c->alloc(&c->d_zh_inv, rate));
    CK(h2d(ctx, d_zh, zh.data(), rate * 8)); CK(h2d(ctx, c->d_zh_inv, zh_inv.data(), rate * 8));
    {
        const uint32_t lo_bits = (L + 1) / 2;
        u64 *d_lo = nullptr, *d_hi = nullptr;
        const u64 wL = gl::root_of_unity(L);
        auto lo = powers_table(wL, 1ull << lo_bits), hi = powers_table(gl::pow(wL, 1ull << lo_bits), 1ull << (L - lo_bits));
        CK(c->alloc(&d_lo, lo.size())); CK(c->alloc(&d_hi, hi.size()));
        CK(h2d(ctx, d_lo, lo.data(), lo.size() * 8)); CK(h2d(ctx, d_hi, hi.data(), hi.size() * 8));
        CK(c->alloc(&c->d_x_coset, lde_n)); CK(c->alloc(&c->d_l0_coset, lde_n));
        hipError_t e = pk_coset_tables(lde_n, L, d_lo, d_hi, lo_bits, d_zh, rate, gl::canon(n % gl::P), c->d_x_coset, c->d_l0_coset, ctx->stream);
        if (e != hipSuccess) return fail(ctx->hip_fail(e, "coset_tables"));
        const u64 ginv = gl::inv(gl::MULT_GEN);
        c->ginv_lo_bits = lo_bits;
        auto glo = powers_table(ginv, 1ull << lo_bits), ghi = powers_table(gl::pow(ginv, 1ull << lo_bits), 1ull << (L - lo_bits));
        CK(c->alloc(&c->d_ginv_lo, glo.size())); CK(c->alloc(&c->d_ginv_hi, ghi.size()));
        CK(h2d(ctx, c->d_ginv_lo, glo.data(), glo.size() * 8)); CK(h2d(ctx, c->d_ginv_hi, ghi.data(), ghi.size() * 8));
    }
    // per-proof workspace
    CK(c->alloc(&c->d_wires_vals, (size_t)p.num_wires * n));
    CK(alloc_batch(c, c->wires, (uint32_t)p.num_wires, true, 1));
    CK(alloc_batch(c, c->zs, (uint32_t)p.num_zs_pp_cols(), true, 2));
    CK(alloc_batch(c, c->quot, (uint32_t)p.num_quotient_cols(), false, 3));
    CK(c->alloc(&c->quot.coeffs, (size_t)nch * lde_n));       // quotient values -> coefficients, = nq chunks of n
    CK(c->alloc(&c->d_qacc, (size_t)nch * lde_n));
    CK(c->alloc(&c->d_qcp, (size_t)nch * p.num_chunks() * n));
    CK(c->alloc(&c->d_rowprod, (size_t)nch * n));
    CK(c->alloc(&c->d_z, (size_t)nch * n));
    CK(c->alloc(&c->d_zs_vals, (size_t)p.num_zs_pp_cols() * n));
    const size_t nterms = nch + nch * p.num_chunks() + p.num_gate_constraints;
    CK(c->alloc(&c->d_small, 2 * nch + nch * R + nch * nterms + 4 + 64));
    const size_t n_open = p.num_cs_cols() + p.num_wires + p.num_zs_pp_cols() + p.num_quotient_cols();
    CK(c->alloc(&c->d_points, 2));
    CK(c->alloc(&c->d_open, n_open + nch));
    CK(c->alloc(&c->d_alpha_ext, n_open));
    CK(c->alloc(&c->d_comp, 2 * n));
    CK(c->alloc(&c->d_fin, 2 * n));
    CK(c->alloc(&c->d_fri_vals, 2 * lde_n));
    CK(c->alloc(&c->d_fri_rows, 2 * lde_n));
    CK(c->alloc(&c->d_fri_coeffs[0], 2 * n)); CK(c->alloc(&c->d_fri_coeffs[1], 2 * n));
    {
        unsigned lvl = L;
        size_t gw = 0;
        const size_t salt = p.zero_knowledge ? 4 : 0;
        const size_t widths[4] = {(size_t)p.num_cs_cols(), (size_t)p.num_wires + salt, (size_t)p.num_zs_pp_cols() + salt, (size_t)p.num_quotient_cols() + salt};
        for (size_t w : widths) gw += w + (L - p.cap_height) * 4;
        for (u64 a : p.arity_bits) {
            lvl -= (unsigned)a;
            u64 *dg = nullptr, *rows = nullptr;
            CK(c->alloc(&dg, digest_words(lvl, (unsigned)p.cap_height)));
            CK(c->alloc(&rows, (size_t)2 << (lvl + a)));
            c->d_fri_digests.push_back(dg); c->d_fri_leafrows.push_back(rows);
            gw += (2ull << a) + (lvl - p.cap_height) * 4;
        }
        c->gather_words = gw * p.num_query_rounds;
        CK(c->alloc(&c->d_gather, c->gather_words));
        CK(c->alloc(&c->d_qidx, p.num_query_rounds));
        CK(c->alloc(&c->d_pow, 1));
        CK(c->alloc(&c->d_check, 2));
    }
    {
        c->stage_words = 2 * nch + nch * R + nch * nterms + 4 + 4 + 2 * n_open + p.num_query_rounds + 64 + 8;
        void *hp = nullptr;
        hipError_t e = hipHostMalloc(&hp, c->stage_words * 8, hipHostMallocDefault);
        if (e != hipSuccess) return fail(ctx->hip_fail(e, "hipHostMalloc(stage)"));
        c->h_stage = (u64 *)hp;
    }
    QP_HIP(ctx, hipStreamSynchronize(ctx->stream));
#undef CK
    *out = c;
    return QPGPU_OK;
}

int qpgpu_circuit_set_witness_check(qpgpu_circuit *c, int on) {
    if (!c) return QPGPU_EINVAL;
    c->check_witness = on != 0;
    return QPGPU_OK;
}

int qpgpu_circuit_set_blinding_seed(qpgpu_circuit *c, uint64_t seed) {
    if (!c) return QPGPU_EINVAL;
    c->blinding_seed = seed; c->seed_set = true;
    return QPGPU_OK;
}

int qpgpu_circuit_constants_sigmas_cap(const qpgpu_circuit *c, uint64_t *out, size_t out_words) {
    if (!c || !out || out_words < c->cs.cap.size()) return QPGPU_EINVAL;
    std::memcpy(out, c->cs.cap.data(), c->cs.cap.size() * 8);
    return QPGPU_OK;
}

static int prove_impl(qpgpu_circuit *c, const u64 *d_wires, const u64 *public_inputs, uint8_t *out, size_t out_cap, size_t *out_len) {
    qpgpu_ctx *ctx = c->ctx;
    const CircuitPack &p = c->pack;
    hipStream_t st = ctx->stream;
    const u64 n = p.n(), lde_n = n << p.rate_bits, R = p.num_routed_wires;
    const uint32_t nch = (uint32_t)p.num_challenges, npp = (uint32_t)p.num_partial_products, nchunks = npp + 1;
    const unsigned d = (unsigned)p.degree_bits, L = (unsigned)(p.degree_bits + p.rate_bits), cap_h = (unsigned)p.cap_height;
    const size_t ncs = p.num_cs_cols(), NW = p.num_wires, nzp = p.num_zs_pp_cols(), nq = p.num_quotient_cols();
    const size_t sig0 = p.num_selectors + p.num_constants, cap_words = (1ull << cap_h) * 4;
    const size_t nterms = nch + nch * nchunks + p.num_gate_constraints;

    c->stage_pos = 0;
    u64 pih[4];
    host_hash_no_pad(public_inputs, p.num_public_inputs, pih);
    if (p.zero_knowledge && !c->seed_set) {   // fresh randomness per proof unless the caller injected a seed
        std::random_device rd;
        c->blinding_seed = ((u64)rd() << 32) ^ (u64)rd() ^ ((u64)rd() << 17);
    }
    c->seed_set = false;

    // ---- s2/s3 wires ----
    ctx->prof_begin("prove_commit_wires");
    QP_TRY(commit_values(c, d_wires, c->wires));
    ctx->prof_end();
    Challenger ch;
    ch.observe(p.circuit_digest, 4);
    ch.observe(pih, 4);
    ch.observe(c->wires.cap.data(), cap_words);
    u64 betas[4], gammas[4], alphas[4];
    for (uint32_t k = 0; k < nch; k++) betas[k] = ch.get();
    for (uint32_t k = 0; k < nch; k++) gammas[k] = ch.get();

    // small tables: [betas nch][gammas nch][beta_k_is nch*R][alpha_pows nch*nterms][pi_hash 4]
    u64 *d_betas = c->d_small, *d_gammas = d_betas + nch, *d_bk = d_gammas + nch, *d_apow = d_bk + (size_t)nch * R, *d_pih = d_apow + (size_t)nch * nterms;
    std::vector<u64> small(2 * nch + (size_t)nch * R);
    for (uint32_t k = 0; k < nch; k++) { small[k] = betas[k]; small[nch + k] = gammas[k]; for (u64 j = 0; j < R; j++) small[2 * nch + k * R + j] = gl::canon(gl::mul(betas[k], p.k_is[j])); }
    QP_TRY(h2d_staged(c, c->d_small, small.data(), small.size() * 8));

    // ---- s5 partial products ----
    ctx->prof_begin("prove_partial_products");
    PpArgs pa{};
    pa.wires = d_wires; pa.sigmas = c->d_cs_values + sig0 * n; pa.omega_pows = c->d_omega; pa.beta_k_is = d_bk;
    pa.betas = d_betas; pa.gammas = d_gammas; pa.qcp = c->d_qcp; pa.rowprod = c->d_rowprod; pa.n = n;
    pa.num_routed = (uint32_t)R; pa.chunk = (uint32_t)p.quotient_degree_factor; pa.nchunks = nchunks; pa.nch = nch;
    QP_HIP(ctx, pk_pp_rows(pa, st));
    QP_HIP(ctx, pk_pp_scan(c->d_rowprod, c->d_z, n, nch, st));
    QP_HIP(ctx, pk_pp_finish(pa, c->d_z, c->d_zs_vals, st));
    ctx->prof_end();
    ctx->prof_begin("prove_commit_zs");
    QP_TRY(commit_values(c, c->d_zs_vals, c->zs));
    ctx->prof_end();
    ch.observe(c->zs.cap.data(), cap_words);
    for (uint32_t k = 0; k < nch; k++) alphas[k] = ch.get();

    // ---- s6 quotient ----
    {
        std::vector<u64> ap((size_t)nch * nterms + 4);
        for (uint32_t k = 0; k < nch; k++) { u64 a = 1; for (size_t t = 0; t < nterms; t++) { ap[k * nterms + t] = gl::canon(a); a = gl::mul(a, alphas[k]); } }
        std::memcpy(ap.data() + (size_t)nch * nterms, pih, 32);
        QP_TRY(h2d_staged(c, d_apow, ap.data(), ap.size() * 8));
    }
    if (c->check_witness) {
        // the analogue of plonky2's debug assertions: filtered gate constraints must vanish on every trace row and the
        // permutation product must close; alpha-weighted sums are zero iff every constraint is (alpha is a transcript challenge)
        QuotientArgs ta{};
        ta.wires = d_wires; ta.cs = c->d_cs_values; ta.alpha_pows = d_apow; ta.pi_hash = d_pih; ta.gates = c->d_gates;
        ta.acc = c->d_qacc; ta.out = c->d_qacc; ta.poseidon_rc = c->d_poseidon_rc; ta.poseidon_fast = c->d_poseidon_fast;
        ta.zh_inv = c->d_zh_inv; ta.lde_n = n; ta.log_lde = d; ta.rate = 1; ta.nch = nch; ta.num_routed = (uint32_t)R;
        ta.chunk = (uint32_t)p.quotient_degree_factor; ta.nchunks = nchunks; ta.sig0 = (uint32_t)sig0;
        ta.num_selectors = (uint32_t)p.num_selectors; ta.num_gates = (uint32_t)p.gates.size(); ta.nterms = (uint32_t)nterms;
        QP_HIP(ctx, hipMemsetAsync(c->d_qacc, 0, (size_t)nch * n * 8, st));
        const u64 init[2] = {~0ull, 0};
        QP_TRY(h2d_staged(c, c->d_check, init, sizeof init));
        QP_HIP(ctx, pk_gate_sums(ta, c->h_gates.data(), st));
        QP_HIP(ctx, pk_witness_check(c->d_qacc, n, nch, c->d_z, c->d_rowprod, c->d_check, st));
        u64 res[2];
        QP_HIP(ctx, hipMemcpyAsync(res, c->d_check, sizeof res, hipMemcpyDeviceToHost, st));
        QP_HIP(ctx, hipStreamSynchronize(st));
        if (res[0] != ~0ull) return ctx->fail(QPGPU_EUNSAT, "witness does not satisfy the circuit: gate constraints fail at row " + std::to_string(res[0]));
        if (res[1]) return ctx->fail(QPGPU_EUNSAT, "witness does not satisfy the circuit: a copy constraint is violated (permutation product != 1)");
    }
    ctx->prof_begin("prove_quotient");
    QuotientArgs qa{};
    qa.wires = c->wires.lde; qa.cs = c->cs.lde; qa.zs_pp = c->zs.lde; qa.x_coset = c->d_x_coset; qa.l0_coset = c->d_l0_coset;
    qa.zh_inv = c->d_zh_inv; qa.alpha_pows = d_apow; qa.beta_k_is = d_bk; qa.betas = d_betas; qa.gammas = d_gammas; qa.pi_hash = d_pih;
    qa.gates = c->d_gates; qa.acc = c->d_qacc; qa.poseidon_rc = c->d_poseidon_rc; qa.poseidon_fast = c->d_poseidon_fast; qa.out = c->quot.coeffs; qa.lde_n = lde_n; qa.log_lde = L; qa.rate = 1u << p.rate_bits; qa.nch = nch;
    qa.num_routed = (uint32_t)R; qa.chunk = (uint32_t)p.quotient_degree_factor; qa.nchunks = nchunks; qa.sig0 = (uint32_t)sig0;
    qa.num_selectors = (uint32_t)p.num_selectors; qa.num_gates = (uint32_t)p.gates.size(); qa.nterms = (uint32_t)nterms;
    QP_HIP(ctx, pk_quotient(qa, c->h_gates.data(), st));
    // coset_ifft(g): ifft then scale coefficient i by g^-i; the 8n coefficients are the qdf chunks of n, contiguous
    QP_TRY(ntt_run(ctx, c->quot.coeffs, c->quot.coeffs, L, L, nch, true, false, 0));
    QP_HIP(ctx, pk_scale_powers(c->quot.coeffs, lde_n, nch, c->d_ginv_lo, c->d_ginv_hi, c->ginv_lo_bits, st));
    ctx->prof_end();
    ctx->prof_begin("prove_commit_quotient");
    QP_TRY(commit_coeffs(c, c->quot));
    ctx->prof_end();
    ch.observe(c->quot.cap.data(), cap_words);
    const e2 zeta = ch.get_ext();
    {   // plonky2 rejects an opening point inside the subgroup
        e2 zn = zeta;
        for (unsigned i = 0; i < d; i++) zn = gl::e2_mul(zn, zn);
        zn = gl::e2_canon(zn);
        if (zn.a == 1 && zn.b == 0) return ctx->fail(QPGPU_EINVAL, "prove: opening point is in the subgroup");
    }
    const e2 g_zeta = gl::e2_canon(gl::e2_scale(zeta, gl::root_of_unity(d)));

    // ---- s7 openings ----
    ctx->prof_begin("prove_openings");
    const size_t n_open = ncs + NW + nzp + nq;
    {
        e2 pts[2] = {zeta, g_zeta};
        QP_TRY(h2d_staged(c, c->d_points, pts, sizeof pts));
        const DevBatch *bs[4] = {&c->cs, &c->wires, &c->zs, &c->quot};
        size_t off = 0;
        for (const DevBatch *b : bs) { QP_HIP(ctx, pk_poly_eval(b->coeffs, n, b->ncols, c->d_points, 1, nullptr, c->d_open + off, st)); off += b->ncols; }
        QP_HIP(ctx, pk_poly_eval(c->zs.coeffs, n, nch, c->d_points + 1, 1, nullptr, c->d_open + n_open, st));
    }
    std::vector<e2> open(n_open + nch);
    QP_HIP(ctx, hipMemcpyAsync(open.data(), c->d_open, open.size() * sizeof(e2), hipMemcpyDeviceToHost, st));
    QP_HIP(ctx, hipStreamSynchronize(st));
    ctx->prof_end();
    ch.observe((const u64 *)open.data(), n_open * 2);
    ch.observe((const u64 *)(open.data() + n_open), (size_t)nch * 2);
    const e2 fri_alpha = ch.get_ext();

    // ---- s8 batched opening polynomial ----
    ctx->prof_begin("prove_fri_batch");
    {
        std::vector<e2> apw(n_open);
        e2 a = gl::e2_from(1);
        for (size_t i = 0; i < n_open; i++) { apw[i] = gl::e2_canon(a); a = gl::e2_mul(a, fri_alpha); }
        QP_TRY(h2d_staged(c, c->d_alpha_ext, apw.data(), apw.size() * sizeof(e2)));
        ReduceArgs ra{};
        ra.src[0] = c->cs.coeffs; ra.src[1] = c->wires.coeffs; ra.src[2] = c->zs.coeffs; ra.src[3] = c->quot.coeffs;
        ra.ncols[0] = (uint32_t)ncs; ra.ncols[1] = (uint32_t)NW; ra.ncols[2] = (uint32_t)nzp; ra.ncols[3] = (uint32_t)nq; ra.nsrc = 4;
        ra.alpha_pows = c->d_alpha_ext; ra.comp_a = c->d_comp; ra.comp_b = c->d_comp + n; ra.n = n;
        QP_HIP(ctx, pk_reduce_polys(ra, st));
        QP_HIP(ctx, pk_divide_linear(c->d_comp, c->d_comp + n, n, zeta, gl::e2_from(1), 0, c->d_fin, c->d_fin + n, st));
        ra.src[0] = c->zs.coeffs; ra.ncols[0] = nch; ra.nsrc = 1;
        QP_HIP(ctx, pk_reduce_polys(ra, st));
        QP_HIP(ctx, pk_divide_linear(c->d_comp, c->d_comp + n, n, g_zeta, gl::e2_pow(fri_alpha, nch), 1, c->d_fin, c->d_fin + n, st));
    }
    ctx->prof_end();

    // ---- s9 FRI commit phase ----
    ctx->prof_begin("prove_fri_commit");
    std::vector<std::vector<u64>> fri_caps;
    std::vector<unsigned> tree_log_leaves;
    u64 shift = gl::MULT_GEN;
    u64 *coef = c->d_fin;           // [2][valid]
    u64 valid = n; unsigned log_len = L;
    size_t fri_slot = 0;
    // values of the first layer: LDE of the two component columns, leaf order
    QP_TRY(ntt_run(ctx, coef, c->d_fri_vals, d, L, 2, false, true, shift));
    for (size_t r = 0; r < p.arity_bits.size(); r++) {
        const unsigned ab = (unsigned)p.arity_bits[r];
        const u64 len = 1ull << log_len, arity = 1ull << ab;
        const unsigned log_leaves = log_len - ab;
        u64 *rows = c->d_fri_leafrows[r];
        QP_HIP(ctx, pk_interleave_ext(c->d_fri_vals, c->d_fri_vals + len, len, rows, st));
        // leaves = chunks of `arity` extension values = 2*arity consecutive felts
        MerkleLeafArgs unused{}; (void)unused;
        QP_HIP(ctx, merkle_leaf_hash_rows(rows, 1ull << log_leaves, (uint32_t)(2 * arity), c->d_fri_digests[r], st));
        {
            u64 cnt = 1ull << log_leaves; u64 *lvl = c->d_fri_digests[r];
            while (cnt > (1ull << cap_h)) { QP_HIP(ctx, merkle_reduce_level(lvl, lvl + cnt * 4, cnt / 2, st)); lvl += cnt * 4; cnt >>= 1; }
            std::vector<u64> capv(cap_words);
            QP_HIP(ctx, hipMemcpyAsync(capv.data(), lvl, cap_words * 8, hipMemcpyDeviceToHost, st));
            QP_HIP(ctx, hipStreamSynchronize(st));
            fri_caps.push_back(capv);
        }
        tree_log_leaves.push_back(log_leaves);
        ch.observe(fri_caps.back().data(), cap_words);
        const e2 beta = ch.get_ext();
        const u64 new_valid = valid >> ab;
        u64 *ncoef = c->d_fri_coeffs[fri_slot]; fri_slot ^= 1;
        QP_HIP(ctx, pk_fri_fold(coef, coef + valid, new_valid, (uint32_t)arity, beta, ncoef, ncoef + new_valid, st));
        coef = ncoef; valid = new_valid; log_len -= ab;
        shift = gl::pow(shift, arity);
        if (r + 1 < p.arity_bits.size()) {
            unsigned lv = 0; while ((1ull << lv) < valid) lv++;
            QP_TRY(ntt_run(ctx, coef, c->d_fri_vals, lv, log_len, 2, false, true, shift));
        }
    }
    std::vector<u64> final_coeffs(2 * valid);   // component arrays [a...][b...]
    QP_HIP(ctx, hipMemcpyAsync(final_coeffs.data(), coef, final_coeffs.size() * 8, hipMemcpyDeviceToHost, st));
    QP_HIP(ctx, hipStreamSynchronize(st));
    ctx->prof_end();
    std::vector<e2> final_poly(valid);
    for (u64 i = 0; i < valid; i++) final_poly[i] = gl::e2_make(final_coeffs[i], final_coeffs[valid + i]);
    ch.observe((const u64 *)final_poly.data(), 2 * valid);

    // ---- s10 proof of work: minimum nonce ----
    ctx->prof_begin("prove_pow");
    u64 pow_witness = 0;
    {
        PowArgs pw{};
        std::memcpy(pw.state, ch.state, sizeof pw.state);
        for (int i = 0; i < ch.n_in; i++) pw.state[i] = ch.in[i];
        pw.pos = (uint32_t)ch.n_in; pw.pow_bits = (uint32_t)p.proof_of_work_bits; pw.result = c->d_pow;
        if (pw.pow_bits == 0) pow_witness = 0;
        else {
            // expected 2^pow_bits candidates; a batch of 2x that finds it 86% of the time and costs one wave per SIMD
            const u64 batch = std::max<u64>(1ull << 16, 2ull << pw.pow_bits);
            bool found = false;
            for (u64 base = 0; !found; base += batch) {
                const u64 sentinel = ~0ull;
                QP_HIP(ctx, hipMemcpyAsync(c->d_pow, &sentinel, 8, hipMemcpyHostToDevice, st));
                pw.base = base; pw.count = batch;
                QP_HIP(ctx, pk_pow(pw, st));
                u64 res = 0;
                QP_HIP(ctx, hipMemcpyAsync(&res, c->d_pow, 8, hipMemcpyDeviceToHost, st));
                QP_HIP(ctx, hipStreamSynchronize(st));
                if (res != sentinel) { pow_witness = res; found = true; }
                if (base > (1ull << 40)) return ctx->fail(QPGPU_EDEVICE, "prove: proof of work not found");
            }
        }
    }
    ctx->prof_end();
    ch.observe(&pow_witness, 1);
    (void)ch.get();   // the response, re-derived by the verifier

    // ---- s11 queries ----
    ctx->prof_begin("prove_queries");
    const uint32_t nqr = (uint32_t)p.num_query_rounds;
    std::vector<u64> qidx(nqr);
    for (auto &x : qidx) x = ch.get() % lde_n;
    QP_TRY(h2d_staged(c, c->d_qidx, qidx.data(), nqr * 8));
    // gather layout (per section, all queries contiguous): for each oracle rows then paths; for each FRI round evals then paths
    struct Sec { size_t off, words; bool is_path; };
    std::vector<Sec> secs;
    size_t goff = 0;
    const DevBatch *bs[4] = {&c->cs, &c->wires, &c->zs, &c->quot};
    const uint32_t plen0 = L - cap_h;
    for (const DevBatch *b : bs) {
        QP_HIP(ctx, pk_gather_rows(b->lde, lde_n, b->ncols, c->d_qidx, nqr, c->d_gather + goff, st));
        secs.push_back({goff, b->ncols, false}); goff += (size_t)b->ncols * nqr;
        if (b->salt) {
            QP_HIP(ctx, pk_gather_rows(b->salt, lde_n, 4, c->d_qidx, nqr, c->d_gather + goff, st));
            secs.push_back({goff, 4, false}); goff += (size_t)4 * nqr;
        }
        QP_HIP(ctx, pk_gather_paths(b->digests, lde_n, plen0, c->d_qidx, 0, nqr, c->d_gather + goff, st));
        secs.push_back({goff, (size_t)plen0 * 4, true}); goff += (size_t)plen0 * 4 * nqr;
    }
    {
        uint32_t sh = 0;
        for (size_t r = 0; r < p.arity_bits.size(); r++) {
            const uint32_t ab = (uint32_t)p.arity_bits[r], width = 2u << ab, pl = tree_log_leaves[r] - cap_h;
            sh += ab;
            QP_HIP(ctx, pk_gather_leaf_rows(c->d_fri_leafrows[r], width, c->d_qidx, sh, nqr, c->d_gather + goff, st));
            secs.push_back({goff, width, false}); goff += (size_t)width * nqr;
            QP_HIP(ctx, pk_gather_paths(c->d_fri_digests[r], 1ull << tree_log_leaves[r], pl, c->d_qidx, sh, nqr, c->d_gather + goff, st));
            secs.push_back({goff, (size_t)pl * 4, true}); goff += (size_t)pl * 4 * nqr;
        }
    }
    if (goff != c->gather_words) return ctx->fail(QPGPU_EDEVICE, "prove: internal gather size mismatch");
    std::vector<u64> gathered(goff);
    QP_HIP(ctx, hipMemcpyAsync(gathered.data(), c->d_gather, goff * 8, hipMemcpyDeviceToHost, st));
    QP_HIP(ctx, hipStreamSynchronize(st));
    ctx->prof_end();

    // ---- s12 ProofWithPublicInputs::to_bytes ----
    ByteWriter w{out, out_cap};
    w.vec(c->wires.cap.data(), cap_words); w.vec(c->zs.cap.data(), cap_words); w.vec(c->quot.cap.data(), cap_words);
    const e2 *o_cs = open.data(), *o_w = o_cs + ncs, *o_zs = o_w + NW, *o_pp = o_zs + nch, *o_q = o_pp + (size_t)nch * npp, *o_zn = open.data() + n_open;
    for (size_t i = 0; i < ncs; i++) w.ext(o_cs[i]);          // constants, plonk_sigmas
    for (size_t i = 0; i < NW; i++) w.ext(o_w[i]);            // wires
    for (size_t i = 0; i < nch; i++) w.ext(o_zs[i]);          // plonk_zs
    for (size_t i = 0; i < nch; i++) w.ext(o_zn[i]);          // plonk_zs_next
    for (size_t i = 0; i < (size_t)nch * npp; i++) w.ext(o_pp[i]);   // partial_products
    for (size_t i = 0; i < nq; i++) w.ext(o_q[i]);            // quotient_polys (lookup vectors are empty)
    for (auto &cp : fri_caps) w.vec(cp.data(), cap_words);
    for (uint32_t q = 0; q < nqr; q++) {
        for (const Sec &sc : secs) {
            if (sc.is_path) w.u8((uint8_t)(sc.words / 4));   // write_merkle_proof: one-byte sibling count
            w.vec(gathered.data() + sc.off + (size_t)q * sc.words, sc.words);
        }
    }
    for (u64 i = 0; i < valid; i++) w.ext(final_poly[i]);
    w.u64le(pow_witness);
    for (size_t i = 0; i < p.num_public_inputs; i++) w.u64le(gl::canon(public_inputs[i]));
    if (out_len) *out_len = w.len;
    if (w.overflow) return ctx->fail(QPGPU_EBUFSIZE, "prove: output buffer too small");
    return QPGPU_OK;
}

int qpgpu_prove_dev(qpgpu_circuit *c, const uint64_t *d_wires, const uint64_t *public_inputs, uint8_t *out, size_t out_cap, size_t *out_len) {
    if (!c) return QPGPU_EINVAL;
    QP_DEV(c->ctx);
    if (!d_wires || (!public_inputs && c->pack.num_public_inputs) || !out) return c->ctx->fail(QPGPU_EINVAL, "prove: null argument");
    return prove_impl(c, d_wires, public_inputs, out, out_cap, out_len);
}

int qpgpu_prove(qpgpu_circuit *c, const uint64_t *wires, const uint64_t *public_inputs, uint8_t *out, size_t out_cap, size_t *out_len) {
    if (!c) return QPGPU_EINVAL;
    QP_DEV(c->ctx);
    if (!wires || (!public_inputs && c->pack.num_public_inputs) || !out) return c->ctx->fail(QPGPU_EINVAL, "prove: null argument");
    const size_t bytes = (size_t)c->pack.num_wires * c->pack.n() * 8;
    QP_TRY(h2d(c->ctx, c->d_wires_vals, wires, bytes));
    int rc = prove_impl(c, c->d_wires_vals, public_inputs, out, out_cap, out_len);
    // the witness carries the spend secret (reference wormhole/circuit/src/sensitive.rs:36-44): scrub the device copy
    (void)hipMemsetAsync(c->d_wires_vals, 0, bytes, c->ctx->stream);
    (void)hipMemsetAsync(c->wires.coeffs, 0, bytes, c->ctx->stream);
    (void)hipMemsetAsync(c->wires.lde, 0, bytes << c->pack.rate_bits, c->ctx->stream);
    (void)hipStreamSynchronize(c->ctx->stream);
    return rc;
}

}  // extern "C"
