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
#include "prover_host.hpp"
#include "circuit_state.hpp"
#include "prover_kernels.hpp"

using gl::e2;
using gl::u64;

namespace {

void host_hash_no_pad(const u64 *in, size_t n, u64 out[4]) {
    u64 st[12] = {0};
    for (size_t i = 0; i < n; i += 8) {
        size_t len = std::min<size_t>(8, n - i);
        for (size_t k = 0; k < len; k++) st[k] = gl::canon(in[i + k]);
        hasher::host_permute(st);
    }
    std::memcpy(out, st, 32);
}

std::vector<u64> powers_table(u64 base, u64 count) {
    std::vector<u64> t(count);
    u64 a = 1;
    for (u64 i = 0; i < count; i++) { t[i] = gl::canon(a); a = gl::mul(a, base); }
    return t;
}

}  // namespace


namespace {

int alloc_batch(qpgpu_circuit *c, PolyOracle &b, uint32_t ncols, bool need_coeffs = true, uint32_t oracle_index = 0) {
    b.oracle_index = oracle_index;
    if (oracle_index > 0 && c->pack.zero_knowledge) QP_TRY(c->alloc(&b.salt, (size_t)4 << (c->pack.degree_bits + c->pack.rate_bits)));
    const CircuitPack &p = c->pack;
    const u64 n = p.n(), lde_n = n << p.rate_bits;
    b.ncols = ncols; b.log_n = (unsigned)p.degree_bits; b.rate_bits = (unsigned)p.rate_bits; b.cap_h = (unsigned)p.cap_height;
    if (need_coeffs) QP_TRY(c->alloc(&b.coeffs, (size_t)ncols * n));
    QP_TRY(c->alloc(&b.lde, (size_t)ncols * lde_n));
    QP_TRY(c->alloc(&b.digests, digest_words((unsigned)(p.degree_bits + p.rate_bits), (unsigned)p.cap_height)));
    b.cap.resize((1ull << p.cap_height) * 4);
    return QPGPU_OK;
}

int commit_coeffs(qpgpu_circuit *c, PolyOracle &b) { return oracle_commit_coeffs(c->ctx, b, c->blinding_seed); }
int commit_values(qpgpu_circuit *c, const u64 *d_values, PolyOracle &b) { return oracle_commit_values(c->ctx, d_values, b, c->blinding_seed); }
int h2d_staged(qpgpu_circuit *c, void *dst, const void *src, size_t bytes) { return c->stage.put(c->ctx, dst, src, bytes); }

int h2d(qpgpu_ctx *ctx, void *dst, const void *src, size_t bytes) {
    QP_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    QP_HIP(ctx, hipStreamSynchronize(ctx->stream));   // source is pageable and may go out of scope
    return QPGPU_OK;
}

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
    if (c->stage.h) (void)hipHostFree(c->stage.h);
    for (void *p : c->allocs) (void)hipFree(p);
    witness_plan_free(c->wplan);
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
    if (p.degree_bits + p.rate_bits > 23) { delete c; return ctx->fail(QPGPU_EINVAL, "circuit_load: LDE larger than 2^23 not supported"); }
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
                 (uint32_t)g.group_end, (uint32_t)g.num_constraints, (uint32_t)g.param2};
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
    {
        // FRI opening proof over the four oracles: workspace carved out of one allocation
        c->fri.degree_bits = d; c->fri.rate_bits = (unsigned)p.rate_bits; c->fri.cap_h = (unsigned)p.cap_height;
        c->fri.pow_bits = (unsigned)p.proof_of_work_bits; c->fri.num_queries = (uint32_t)p.num_query_rounds;
        for (u64 a : p.arity_bits) c->fri.arity_bits.push_back((unsigned)a);
        const size_t salt = p.zero_knowledge ? 4 : 0;
        const std::vector<size_t> widths = {(size_t)p.num_cs_cols(), (size_t)p.num_wires + salt, (size_t)p.num_zs_pp_cols() + salt, (size_t)p.num_quotient_cols() + salt};
        u64 *base = nullptr;
        CK(c->alloc(&base, FriWork::words(c->fri, widths, n_open)));
        c->fri_work.bind(base, c->fri, widths, n_open);
        CK(c->alloc(&c->d_check, 2));
    }
    {
        c->stage.words = 2 * nch + nch * R + nch * nterms + 4 + 4 + FriWork::stage_words(c->fri, n_open) + 64 + 8;
        void *hp = nullptr;
        hipError_t e = hipHostMalloc(&hp, c->stage.words * 8, hipHostMallocDefault);
        if (e != hipSuccess) return fail(ctx->hip_fail(e, "hipHostMalloc(stage)"));
        c->stage.h = (u64 *)hp;
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
    // the quotient lives on the coset of size n * quotient_degree_factor: every 2^q_shift-th point of the LDE, i.e. its first q_n slots
    unsigned qbits = 0; while ((1ull << qbits) < p.quotient_degree_factor) qbits++;
    const unsigned Lq = d + qbits, q_shift = (unsigned)p.rate_bits - qbits;
    const u64 q_n = 1ull << Lq;

    c->stage.pos = 0;
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
        ta.zh_inv = c->d_zh_inv; ta.lde_n = n; ta.q_n = n; ta.q_shift = 0; ta.log_lde = d; ta.rate = 1; ta.nch = nch; ta.num_routed = (uint32_t)R;
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
    qa.gates = c->d_gates; qa.acc = c->d_qacc; qa.poseidon_rc = c->d_poseidon_rc; qa.poseidon_fast = c->d_poseidon_fast; qa.out = c->quot.coeffs; qa.lde_n = lde_n; qa.q_n = q_n; qa.q_shift = q_shift; qa.log_lde = L; qa.rate = 1u << p.rate_bits; qa.nch = nch;
    qa.num_routed = (uint32_t)R; qa.chunk = (uint32_t)p.quotient_degree_factor; qa.nchunks = nchunks; qa.sig0 = (uint32_t)sig0;
    qa.num_selectors = (uint32_t)p.num_selectors; qa.num_gates = (uint32_t)p.gates.size(); qa.nterms = (uint32_t)nterms;
    QP_HIP(ctx, pk_quotient(qa, c->h_gates.data(), st));
    // coset_ifft(g) on the quotient domain: ifft then scale coefficient i by g^-i; the qdf*n coefficients of a challenge
    // are its qdf chunks of n, contiguous
    QP_TRY(ntt_run(ctx, c->quot.coeffs, c->quot.coeffs, Lq, Lq, nch, true, false, 0));
    QP_HIP(ctx, pk_scale_powers(c->quot.coeffs, q_n, nch, c->d_ginv_lo, c->d_ginv_hi, c->ginv_lo_bits, st));
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
        const PolyOracle *bs[4] = {&c->cs, &c->wires, &c->zs, &c->quot};
        size_t off = 0;
        for (const PolyOracle *b : bs) { QP_HIP(ctx, pk_poly_eval(b->coeffs, n, b->ncols, c->d_points, 1, nullptr, c->d_open + off, st)); off += b->ncols; }
        QP_HIP(ctx, pk_poly_eval(c->zs.coeffs, n, nch, c->d_points + 1, 1, nullptr, c->d_open + n_open, st));
    }
    std::vector<e2> open(n_open + nch);
    QP_TRY(ctx->read_back(open.data(), c->d_open, open.size() * sizeof(e2)));
    ctx->prof_end();
    ch.observe((const u64 *)open.data(), n_open * 2);
    ch.observe((const u64 *)(open.data() + n_open), (size_t)nch * 2);

    // ---- s12 ProofWithPublicInputs::to_bytes: caps and openings, then the FRI proof, then the public inputs ----
    ByteWriter w{out, out_cap};
    w.vec(c->wires.cap.data(), cap_words); w.vec(c->zs.cap.data(), cap_words); w.vec(c->quot.cap.data(), cap_words);
    const e2 *o_cs = open.data(), *o_w = o_cs + ncs, *o_zs = o_w + NW, *o_pp = o_zs + nch, *o_q = o_pp + (size_t)nch * npp, *o_zn = open.data() + n_open;
    for (size_t i = 0; i < ncs; i++) w.ext(o_cs[i]);          // constants, plonk_sigmas
    for (size_t i = 0; i < NW; i++) w.ext(o_w[i]);            // wires
    for (size_t i = 0; i < nch; i++) w.ext(o_zs[i]);          // plonk_zs
    for (size_t i = 0; i < nch; i++) w.ext(o_zn[i]);          // plonk_zs_next
    for (size_t i = 0; i < (size_t)nch * npp; i++) w.ext(o_pp[i]);   // partial_products
    for (size_t i = 0; i < nq; i++) w.ext(o_q[i]);            // quotient_polys (lookup vectors are empty)

    // ---- s8..s11 PolynomialBatch::prove_openings: every polynomial at zeta, the Zs also at g*zeta ----
    const PolyOracle *bs[4] = {&c->cs, &c->wires, &c->zs, &c->quot};
    std::vector<FriBatch> batches(2);
    batches[0].point = zeta;
    batches[0].ranges = {{0, 0, (uint32_t)ncs}, {1, 0, (uint32_t)NW}, {2, 0, (uint32_t)nzp}, {3, 0, (uint32_t)nq}};
    batches[1].point = g_zeta;
    batches[1].ranges = {{2, 0, nch}};
    QP_TRY(fri_prove(ctx, c->fri, bs, 4, batches, ch, c->fri_work, c->stage, w));
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
