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
#include <array>
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

void host_hash_no_pad(const hasher::Config &h, const u64 *in, size_t n, u64 out[4]) {
    u64 st[12] = {0};
    for (size_t i = 0; i < n; i += 8) {
        size_t len = std::min<size_t>(8, n - i);
        for (size_t k = 0; k < len; k++) st[k] = gl::canon(in[i + k]);
        h.permute(st);
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

int alloc_batch(qpgpu_circuit *c, PolyOracle &b, uint32_t ncols, uint32_t nb, bool need_coeffs = true, uint32_t oracle_index = 0) {
    const CircuitPack &p = c->pack;
    const u64 n = p.n(), lde_n = n << p.rate_bits;
    const bool secret = oracle_index > 0;     // everything but the constants/sigmas oracle derives from the witness
    b.oracle_index = oracle_index;
    b.ncols = ncols; b.log_n = (unsigned)p.degree_bits; b.rate_bits = (unsigned)p.rate_bits; b.cap_h = (unsigned)p.cap_height;
    if (oracle_index > 0 && p.zero_knowledge) QP_TRY(c->alloc(&b.salt, (size_t)nb * 4 * lde_n, true));
    if (need_coeffs) QP_TRY(c->alloc(&b.coeffs, (size_t)nb * ncols * n, secret));
    QP_TRY(c->alloc(&b.lde, (size_t)nb * ncols * lde_n, secret));
    QP_TRY(c->alloc(&b.digests, (size_t)nb * digest_words((unsigned)(p.degree_bits + p.rate_bits), (unsigned)p.cap_height), secret));
    b.set_batch(nb, oracle_index == 0);
    b.cap.resize(b.cap_words() * b.nb);
    return QPGPU_OK;
}

int h2d(qpgpu_ctx *ctx, void *dst, const void *src, size_t bytes) {
    QP_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    QP_HIP(ctx, hipStreamSynchronize(ctx->stream));   // source is pageable and may go out of scope
    return QPGPU_OK;
}

void scrub(qpgpu_circuit *c) {
    for (auto &r : c->secret_allocs) (void)hipMemsetAsync(r.first, 0, r.second, c->ctx->stream);
    (void)hipStreamSynchronize(c->ctx->stream);
    if (c->stage.h) std::memset(c->stage.h, 0, c->stage.words * 8);   // the staging ring held challenges and salt keys
}

// Replays, in planning mode, every transform a lockstep batch of max_batch proofs will issue (prove_batch_impl, fri_prove):
// the context caches their twiddle and coset tables and sizes its scratch here, on the loading thread, so that a proving
// thread never calls hipMalloc / hipFree / an upload (with six of them starting together, each used to create its tables and
// grow its scratch inside its first batch, next to the other threads' launches).
int plan_transforms(qpgpu_circuit *c) {
    qpgpu_ctx *ctx = c->ctx;
    const CircuitPack &p = c->pack;
    const unsigned d = (unsigned)p.degree_bits, L = (unsigned)(p.degree_bits + p.rate_bits);
    unsigned qbits = 0; while ((1ull << qbits) < p.quotient_degree_factor) qbits++;
    const uint32_t B = c->max_batch, nch = (uint32_t)p.num_challenges;
    struct Restore { qpgpu_ctx *c; ~Restore() { c->plan_only = false; } } restore{ctx};
    ctx->plan_only = true;
    const PolyOracle *os[3] = {&c->wires, &c->zs, &c->quot};
    for (const PolyOracle *o : os) {
        if (o != &c->quot) QP_TRY(ntt_run(ctx, o->lde, o->coeffs, d, d, (size_t)B * o->ncols, true, false, 0));   // from_values
        QP_TRY(ntt_run(ctx, o->coeffs, o->lde, d, L, (size_t)B * o->ncols, false, true, gl::MULT_GEN));            // from_coeffs
    }
    QP_TRY(ntt_run(ctx, c->quot.coeffs, c->quot.coeffs, d + qbits, d + qbits, (size_t)nch * B, true, false, 0));   // quotient values -> coefficients
    // FRI commit phase: the batched opening polynomial and every folded polynomial, as two component columns per proof
    NttProofs np; np.nproofs = B; np.in_ps = c->fri_work.ws; np.out_ps = c->fri_work.ws;
    u64 shift = gl::MULT_GEN, valid = p.n();
    unsigned log_len = L;
    QP_TRY(ntt_run(ctx, c->fri_work.fin, c->fri_work.vals, d, L, 2, false, true, shift, np));
    for (size_t r = 0; r < p.arity_bits.size(); r++) {
        const unsigned ab = (unsigned)p.arity_bits[r];
        valid >>= ab; log_len -= ab;
        shift = gl::pow(shift, 1ull << ab);
        if (r + 1 < p.arity_bits.size()) {
            unsigned lv = 0; while ((1ull << lv) < valid) lv++;
            QP_TRY(ntt_run(ctx, c->fri_work.coeffs[0], c->fri_work.vals, lv, log_len, 2, false, true, shift, np));
        }
    }
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
    // the workspace held witness-derived data (reference wormhole/circuit/src/sensitive.rs:36-44): clear it before the
    // allocator can hand the memory to someone else
    scrub(c);                                    // includes the pinned staging ring
    if (c->stage.h) (void)hipHostFree(c->stage.h);
    for (void *p : c->allocs) (void)hipFree(p);
    witness_plan_free(c->wplan);
    delete c;
}

int qpgpu_circuit_scrub(qpgpu_circuit *c) {
    if (!c) return QPGPU_EINVAL;
    QP_DEV(c->ctx);
    scrub(c);
    return QPGPU_OK;
}

int qpgpu_circuit_load_batch(qpgpu_ctx *ctx, const uint64_t *pack_words, size_t n_words, unsigned max_batch, qpgpu_circuit **out) {
    if (!ctx || !out) return QPGPU_EINVAL;
    QP_DEV(ctx);
    *out = nullptr;
    if (!pack_words) return ctx->fail(QPGPU_EINVAL, "circuit_load: null pack");
    if (max_batch == 0 || max_batch > 1024) return ctx->fail(QPGPU_EINVAL, "circuit_load: max_batch must be 1..1024");
    qpgpu_circuit *c = new qpgpu_circuit();
    c->ctx = ctx;
    c->max_batch = max_batch;
    std::string err = c->pack.parse(pack_words, n_words);
    if (!err.empty()) { delete c; return ctx->fail(QPGPU_EINVAL, "circuit_load: " + err); }
    const CircuitPack &p = c->pack;
    if (p.num_chunks() > 16 || p.num_challenges > 4) { delete c; return ctx->fail(QPGPU_EINVAL, "circuit_load: too many chunks/challenges"); }
    if (p.degree_bits + p.rate_bits > 23) { delete c; return ctx->fail(QPGPU_EINVAL, "circuit_load: LDE larger than 2^23 not supported"); }
    if (p.degree_bits + p.rate_bits > 20 && p.rate_bits > 3) { delete c; return ctx->fail(QPGPU_EINVAL, "circuit_load: an LDE beyond 2^20 points needs rate_bits <= 3"); }
    const u64 n = p.n(), lde_n = n << p.rate_bits, R = p.num_routed_wires, nch = p.num_challenges;
    const unsigned d = (unsigned)p.degree_bits, L = (unsigned)(p.degree_bits + p.rate_bits);
    const uint32_t B = max_batch;
    int rc = QPGPU_OK;
    auto fail = [&](int code) { std::string keep = ctx->err; qpgpu_circuit_free(c); ctx->err = keep; return code; };
#define CK(expr) do { rc = (expr); if (rc) return fail(rc); } while (0)
    ctx->hasher_in_use = true;
    CK(merkle_ensure_constants(ctx));
    // constants / sigmas: values (for the sigma columns of s5) and the setup commitment, shared by all proofs
    CK(c->alloc(&c->d_cs_values, (size_t)p.num_cs_cols() * n));
    CK(h2d(ctx, c->d_cs_values, p.constants_sigmas.data(), p.constants_sigmas.size() * 8));
    CK(alloc_batch(c, c->cs, (uint32_t)p.num_cs_cols(), 1));
    CK(oracle_commit_values(ctx, c->d_cs_values, c->cs, nullptr));
    // gates
    std::vector<GateDev> gd(p.gates.size());
    for (size_t i = 0; i < gd.size(); i++) {
        const GateInfo &g = p.gates[i];
        gd[i] = {(uint32_t)g.type, (uint32_t)g.param0, (uint32_t)g.param1, (uint32_t)g.selector_index, (uint32_t)g.group_start,
                 (uint32_t)g.group_end, (uint32_t)g.num_constraints, (uint32_t)g.param2};
    }
    c->h_gates = gd;
    for (const GateInfo &g : p.gates) c->has_p2_gate = c->has_p2_gate || g.type == GATE_POSEIDON2;
    if (c->has_p2_gate) CK(ctx->ensure_p2_app());
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
    // per-proof workspace, B proofs
    unsigned qbits = 0; while ((1ull << qbits) < p.quotient_degree_factor) qbits++;
    const u64 q_n = n << qbits;
    CK(c->alloc(&c->d_wires_vals, (size_t)B * p.num_wires * n, true));
    CK(c->alloc(&c->d_salt_keys, (size_t)B * 8, true));
    CK(alloc_batch(c, c->wires, (uint32_t)p.num_wires, B, true, 1));
    CK(alloc_batch(c, c->zs, (uint32_t)p.num_zs_pp_cols(), B, true, 2));
    CK(alloc_batch(c, c->quot, (uint32_t)p.num_quotient_cols(), B, false, 3));
    CK(c->alloc(&c->quot.coeffs, (size_t)B * nch * q_n, true));      // quotient values -> coefficients, = nq chunks of n, dense per proof
    CK(c->alloc(&c->d_qacc, (size_t)B * nch * lde_n, true));
    CK(c->alloc(&c->d_qcp, (size_t)B * nch * p.num_chunks() * n, true));
    CK(c->alloc(&c->d_rowprod, (size_t)B * nch * n, true));
    CK(c->alloc(&c->d_z, (size_t)B * nch * n, true));
    CK(c->alloc(&c->d_zs_vals, (size_t)B * p.num_zs_pp_cols() * n, true));
    const size_t nterms = nch + nch * p.num_chunks() + p.num_gate_constraints;
    c->small_words = (2 * nch + nch * R + nch * nterms + 4 + 3) & ~(size_t)3;
    CK(c->alloc(&c->d_small, (size_t)B * c->small_words));
    const size_t n_open = p.num_cs_cols() + p.num_wires + p.num_zs_pp_cols() + p.num_quotient_cols();
    CK(c->alloc(&c->d_points, (size_t)B * 2));
    CK(c->alloc(&c->d_open, (size_t)B * (n_open + nch), true));
    {
        // FRI opening proof over the four oracles: workspace carved out of one allocation
        c->fri.degree_bits = d; c->fri.rate_bits = (unsigned)p.rate_bits; c->fri.cap_h = (unsigned)p.cap_height;
        c->fri.pow_bits = (unsigned)p.proof_of_work_bits; c->fri.num_queries = (uint32_t)p.num_query_rounds;
        for (u64 a : p.arity_bits) c->fri.arity_bits.push_back((unsigned)a);
        const size_t salt = p.zero_knowledge ? 4 : 0;
        const std::vector<size_t> widths = {(size_t)p.num_cs_cols(), (size_t)p.num_wires + salt, (size_t)p.num_zs_pp_cols() + salt, (size_t)p.num_quotient_cols() + salt};
        u64 *base = nullptr;
        CK(c->alloc(&base, FriWork::words(c->fri, widths, n_open, B), true));
        c->fri_work.bind(base, c->fri, widths, n_open, B);
        CK(c->alloc(&c->d_check, (size_t)B * 2));
    }
    {
        c->stage.words = (size_t)B * (c->small_words + 8 + 4 + 8) + FriWork::stage_words(c->fri, n_open, B) + 64;
        // test hook: a staging ring too small for anything sends every table through the synchronous bounce path of Stager::put
        if (const char *e = getenv("QPGPU_STAGE_FALLBACK")) if (*e == '1') c->stage.words = 1;
        void *hp = nullptr;
        hipError_t e = hipHostMalloc(&hp, c->stage.words * 8, hipHostMallocDefault);
        if (e != hipSuccess) return fail(ctx->hip_fail(e, "hipHostMalloc(stage)"));
        c->stage.h = (u64 *)hp;
    }
    // everything a batch reads back ends up in its proofs: their total size bounds any single read
    if ((rc = ctx->reserve_read_back((size_t)B * qpgpu_proof_size(c) + (1u << 16))) != QPGPU_OK) return fail(rc);
    CK(plan_transforms(c));
    {   // asynchronous faults of the setup kernels surface here: release everything on the way out
        const hipError_t e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) return fail(ctx->hip_fail(e, "circuit_load: hipStreamSynchronize"));
    }
#undef CK
    *out = c;
    return QPGPU_OK;
}

int qpgpu_circuit_load(qpgpu_ctx *ctx, const uint64_t *pack_words, size_t n_words, qpgpu_circuit **out) {
    return qpgpu_circuit_load_batch(ctx, pack_words, n_words, 1, out);
}

unsigned qpgpu_circuit_max_batch(const qpgpu_circuit *c) { return c ? c->max_batch : 0; }
size_t qpgpu_circuit_num_public_inputs(const qpgpu_circuit *c) { return c ? (size_t)c->pack.num_public_inputs : 0; }

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

}  // extern "C"

// prove() for a lockstep batch of nb proofs of the circuit: d_wires = [nb][num_wires][n] on the device (dense), one
// public-input vector, output buffer and length per proof. Every stage is launched once for the whole batch; the nb
// Fiat-Shamir transcripts are advanced together on the host, one stream synchronisation per stage.
static int prove_batch_impl(qpgpu_circuit *c, uint32_t nb, const u64 *d_wires, const u64 *const *public_inputs, uint8_t *const *outs, size_t out_cap, size_t *out_lens) {
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
    const size_t SW = c->small_words;

    c->stage.pos = 0;
    u64 *ring_keys = nullptr; size_t ring_key_words = 0;
    c->wires.set_batch(nb); c->zs.set_batch(nb); c->quot.set_batch(nb);
    c->quot.ps_coeffs = (u64)nch * q_n;    // = num_quotient_cols * n: the same dense array seen as nq chunk polynomials
    std::vector<std::array<u64, 4>> pih(nb);
    for (uint32_t b = 0; b < nb; b++) host_hash_no_pad(ctx->hasher, public_inputs[b], p.num_public_inputs, pih[b].data());
    if (p.zero_knowledge) {
        // 256 fresh bits per proof from the OS entropy source (the reference: thread_rng); a seed injected for the next
        // batch makes its bytes reproducible (tests)
        std::vector<uint32_t> keys((size_t)nb * 8);
        for (uint32_t b = 0; b < nb; b++) {
            if (c->seed_set) salt_key_from_seed(c->blinding_seed + b, keys.data() + 8 * (size_t)b);
            else if (salt_key_random(keys.data() + 8 * (size_t)b) != QPGPU_OK) return ctx->fail(QPGPU_EDEVICE, "prove: the OS entropy source failed");
        }
        const size_t ring_at = c->stage.pos;
        const int put_rc = c->stage.put(ctx, c->d_salt_keys, keys.data(), keys.size() * 4);
        // the keys are secrets: the host vector goes now, the ring slot as soon as the device has read it (ring_keys below)
        volatile uint32_t *kz = keys.data();
        for (size_t i = 0; i < keys.size(); i++) kz[i] = 0;
        QP_TRY(put_rc);
        if (c->stage.pos > ring_at) { ring_keys = c->stage.h + ring_at; ring_key_words = c->stage.pos - ring_at; }
    }
    c->seed_set = false;

    // ---- s2/s3 wires ----
    ctx->prof_begin("prove_commit_wires");
    const int wires_rc = oracle_commit_values(ctx, d_wires, c->wires, c->d_salt_keys);
    ctx->prof_end();
    if (ring_keys) {   // the commit read its cap back (a stream sync), so the copy kernel that pulled the keys in is done; on a failure wait for it
        if (wires_rc) (void)hipStreamSynchronize(st);
        volatile u64 *rz = ring_keys;
        for (size_t i = 0; i < ring_key_words; i++) rz[i] = 0;
    }
    QP_TRY(wires_rc);
    std::vector<Challenger> chs(nb, Challenger(ctx->hasher));
    std::vector<std::array<u64, 4>> betas(nb), gammas(nb), alphas(nb);

    // small tables per proof: [betas nch][gammas nch][beta_k_is nch*R][alpha_pows nch*nterms][pi_hash 4]
    u64 *d_betas = c->d_small, *d_gammas = d_betas + nch, *d_bk = d_gammas + nch, *d_apow = d_bk + (size_t)nch * R, *d_pih = d_apow + (size_t)nch * nterms;
    std::vector<u64> small((size_t)nb * SW, 0);
    for (uint32_t b = 0; b < nb; b++) {
        Challenger &ch = chs[b];
        ch.observe(p.circuit_digest, 4);
        ch.observe(pih[b].data(), 4);
        ch.observe(c->wires.cap_of(b), cap_words);
        for (uint32_t k = 0; k < nch; k++) betas[b][k] = ch.get();
        for (uint32_t k = 0; k < nch; k++) gammas[b][k] = ch.get();
        u64 *sm = small.data() + (size_t)b * SW;
        for (uint32_t k = 0; k < nch; k++) { sm[k] = betas[b][k]; sm[nch + k] = gammas[b][k]; for (u64 j = 0; j < R; j++) sm[2 * nch + k * R + j] = gl::canon(gl::mul(betas[b][k], p.k_is[j])); }
    }
    // (the alpha part of the table is uploaded after the Z commitment; this upload covers betas .. beta_k_is)
    QP_TRY(c->stage.put_rows(ctx, c->d_small, SW, small.data(), SW, 2 * nch + (size_t)nch * R, nb));

    // ---- s5 partial products ----
    ctx->prof_begin("prove_partial_products");
    PpArgs pa{};
    pa.wires = d_wires; pa.sigmas = c->d_cs_values + sig0 * n; pa.omega_pows = c->d_omega; pa.beta_k_is = d_bk;
    pa.betas = d_betas; pa.gammas = d_gammas; pa.qcp = c->d_qcp; pa.rowprod = c->d_rowprod; pa.n = n;
    pa.num_routed = (uint32_t)R; pa.chunk = (uint32_t)p.quotient_degree_factor; pa.nchunks = nchunks; pa.nch = nch;
    pa.batch = nb; pa.ps_wires = NW * n; pa.ps_small = SW; pa.ps_qcp = (u64)nch * nchunks * n; pa.ps_rowprod = (u64)nch * n;
    QP_HIP(ctx, pk_pp_rows(pa, st));
    QP_HIP(ctx, pk_pp_scan(c->d_rowprod, c->d_z, n, nch * nb, st));
    QP_HIP(ctx, pk_pp_finish(pa, c->d_z, c->d_zs_vals, (u64)nch * n, (u64)nzp * n, st));
    ctx->prof_end();
    ctx->prof_begin("prove_commit_zs");
    QP_TRY(oracle_commit_values(ctx, c->d_zs_vals, c->zs, c->d_salt_keys));
    ctx->prof_end();

    // ---- s6 quotient ----
    for (uint32_t b = 0; b < nb; b++) {
        Challenger &ch = chs[b];
        ch.observe(c->zs.cap_of(b), cap_words);
        for (uint32_t k = 0; k < nch; k++) alphas[b][k] = ch.get();
        u64 *ap = small.data() + (size_t)b * SW + 2 * nch + (size_t)nch * R;
        for (uint32_t k = 0; k < nch; k++) { u64 a = 1; for (size_t t = 0; t < nterms; t++) { ap[k * nterms + t] = gl::canon(a); a = gl::mul(a, alphas[b][k]); } }
        std::memcpy(ap + (size_t)nch * nterms, pih[b].data(), 32);
    }
    QP_TRY(c->stage.put_rows(ctx, d_apow, SW, small.data() + 2 * nch + (size_t)nch * R, SW, (size_t)nch * nterms + 4, nb));
    if (c->check_witness) {
        // the analogue of plonky2's debug assertions: filtered gate constraints must vanish on every trace row and the
        // permutation product must close; alpha-weighted sums are zero iff every constraint is (alpha is a transcript challenge)
        QuotientArgs ta{};
        ta.wires = d_wires; ta.cs = c->d_cs_values; ta.alpha_pows = d_apow; ta.pi_hash = d_pih; ta.gates = c->d_gates;
        ta.acc = c->d_qacc; ta.out = c->d_qacc; ta.poseidon_rc = c->d_poseidon_rc; ta.poseidon_fast = c->d_poseidon_fast;
        ta.p2_gate = ctx->d_p2_app; ta.p2_layout = p.p2_layout;
        ta.zh_inv = c->d_zh_inv; ta.lde_n = n; ta.q_n = n; ta.q_shift = 0; ta.log_lde = d; ta.rate = 1; ta.nch = nch; ta.num_routed = (uint32_t)R;
        ta.chunk = (uint32_t)p.quotient_degree_factor; ta.nchunks = nchunks; ta.sig0 = (uint32_t)sig0;
        ta.num_selectors = (uint32_t)p.num_selectors; ta.num_gates = (uint32_t)p.gates.size(); ta.nterms = (uint32_t)nterms;
        ta.batch = nb; ta.ps_wires = NW * n; ta.ps_zs = 0; ta.ps_small = SW; ta.ps_acc = (u64)nch * n; ta.ps_out = (u64)nch * n;
        QP_HIP(ctx, hipMemsetAsync(c->d_qacc, 0, (size_t)nb * nch * n * 8, st));
        std::vector<u64> init(2 * (size_t)nb);
        for (uint32_t b = 0; b < nb; b++) { init[2 * b] = ~0ull; init[2 * b + 1] = 0; }
        QP_TRY(c->stage.put(ctx, c->d_check, init.data(), init.size() * 8));
        QP_HIP(ctx, pk_gate_sums(ta, c->h_gates.data(), st));
        QP_HIP(ctx, pk_witness_check(c->d_qacc, n, nch, c->d_z, c->d_rowprod, c->d_check, nb, st));
        std::vector<u64> res(2 * (size_t)nb);
        QP_TRY(ctx->read_back(res.data(), c->d_check, res.size() * 8));
        for (uint32_t b = 0; b < nb; b++) {
            const std::string who = nb > 1 ? " (proof " + std::to_string(b) + " of the batch)" : "";
            if (res[2 * b] != ~0ull) return ctx->fail(QPGPU_EUNSAT, "witness does not satisfy the circuit: gate constraints fail at row " + std::to_string(res[2 * b]) + who);
            if (res[2 * b + 1]) return ctx->fail(QPGPU_EUNSAT, "witness does not satisfy the circuit: a copy constraint is violated (permutation product != 1)" + who);
        }
    }
    ctx->prof_begin("prove_quotient");
    QuotientArgs qa{};
    qa.wires = c->wires.lde; qa.cs = c->cs.lde; qa.zs_pp = c->zs.lde; qa.x_coset = c->d_x_coset; qa.l0_coset = c->d_l0_coset;
    qa.zh_inv = c->d_zh_inv; qa.alpha_pows = d_apow; qa.beta_k_is = d_bk; qa.betas = d_betas; qa.gammas = d_gammas; qa.pi_hash = d_pih;
    qa.gates = c->d_gates; qa.acc = c->d_qacc; qa.poseidon_rc = c->d_poseidon_rc; qa.poseidon_fast = c->d_poseidon_fast; qa.out = c->quot.coeffs;
    qa.p2_gate = ctx->d_p2_app; qa.p2_layout = p.p2_layout; qa.lde_n = lde_n; qa.q_n = q_n; qa.q_shift = q_shift; qa.log_lde = L; qa.rate = 1u << p.rate_bits; qa.nch = nch;
    qa.num_routed = (uint32_t)R; qa.chunk = (uint32_t)p.quotient_degree_factor; qa.nchunks = nchunks; qa.sig0 = (uint32_t)sig0;
    qa.num_selectors = (uint32_t)p.num_selectors; qa.num_gates = (uint32_t)p.gates.size(); qa.nterms = (uint32_t)nterms;
    qa.batch = nb; qa.ps_wires = c->wires.ps_lde; qa.ps_zs = c->zs.ps_lde; qa.ps_small = SW; qa.ps_acc = (u64)nch * lde_n; qa.ps_out = (u64)nch * q_n;
    QP_HIP(ctx, pk_quotient(qa, c->h_gates.data(), st));
    // coset_ifft(g) on the quotient domain: ifft then scale coefficient i by g^-i; the qdf*n coefficients of a challenge
    // are its qdf chunks of n, contiguous
    QP_TRY(ntt_run(ctx, c->quot.coeffs, c->quot.coeffs, Lq, Lq, (size_t)nch * nb, true, false, 0));
    QP_HIP(ctx, pk_scale_powers(c->quot.coeffs, q_n, (u64)nch * nb, c->d_ginv_lo, c->d_ginv_hi, c->ginv_lo_bits, st));
    ctx->prof_end();
    ctx->prof_begin("prove_commit_quotient");
    QP_TRY(oracle_commit_coeffs(ctx, c->quot, c->d_salt_keys));
    ctx->prof_end();
    std::vector<e2> zetas(nb), g_zetas(nb), pts(2 * (size_t)nb);
    for (uint32_t b = 0; b < nb; b++) {
        Challenger &ch = chs[b];
        ch.observe(c->quot.cap_of(b), cap_words);
        const e2 zeta = ch.get_ext();
        {   // plonky2 rejects an opening point inside the subgroup
            e2 zn = zeta;
            for (unsigned i = 0; i < d; i++) zn = gl::e2_mul(zn, zn);
            zn = gl::e2_canon(zn);
            if (zn.a == 1 && zn.b == 0) return ctx->fail(QPGPU_EINVAL, "prove: opening point is in the subgroup");
        }
        zetas[b] = zeta;
        g_zetas[b] = gl::e2_canon(gl::e2_scale(zeta, gl::root_of_unity(d)));
        pts[2 * b] = zeta; pts[2 * b + 1] = g_zetas[b];
    }

    // ---- s7 openings ----
    ctx->prof_begin("prove_openings");
    const size_t n_open = ncs + NW + nzp + nq, OW = n_open + nch;      // per proof: [n_open at zeta][nch Z's at g*zeta]
    {
        QP_TRY(c->stage.put(ctx, c->d_points, pts.data(), pts.size() * sizeof(e2)));
        const PolyOracle *bs[4] = {&c->cs, &c->wires, &c->zs, &c->quot};
        size_t off = 0;
        for (const PolyOracle *o : bs) { QP_HIP(ctx, pk_poly_eval(o->coeffs, n, o->ncols, c->d_points, 1, c->d_open + off, nb, o->ps_coeffs, 2, OW, st)); off += o->ncols; }
        QP_HIP(ctx, pk_poly_eval(c->zs.coeffs, n, nch, c->d_points + 1, 1, c->d_open + n_open, nb, c->zs.ps_coeffs, 2, OW, st));
    }
    std::vector<e2> open((size_t)nb * OW);
    QP_TRY(ctx->read_back(open.data(), c->d_open, open.size() * sizeof(e2)));
    ctx->prof_end();

    // ---- s12 ProofWithPublicInputs::to_bytes: caps and openings, then the FRI proof, then the public inputs ----
    std::vector<ByteWriter> ws;
    for (uint32_t b = 0; b < nb; b++) {
        Challenger &ch = chs[b];
        const e2 *op = open.data() + (size_t)b * OW;
        ch.observe((const u64 *)op, n_open * 2);
        ch.observe((const u64 *)(op + n_open), (size_t)nch * 2);
        ws.push_back(ByteWriter{outs[b], out_cap});
        ByteWriter &w = ws.back();
        w.vec(c->wires.cap_of(b), cap_words); w.vec(c->zs.cap_of(b), cap_words); w.vec(c->quot.cap_of(b), cap_words);
        const e2 *o_cs = op, *o_w = o_cs + ncs, *o_zs = o_w + NW, *o_pp = o_zs + nch, *o_q = o_pp + (size_t)nch * npp, *o_zn = op + n_open;
        for (size_t i = 0; i < ncs; i++) w.ext(o_cs[i]);          // constants, plonk_sigmas
        for (size_t i = 0; i < NW; i++) w.ext(o_w[i]);            // wires
        for (size_t i = 0; i < nch; i++) w.ext(o_zs[i]);          // plonk_zs
        for (size_t i = 0; i < nch; i++) w.ext(o_zn[i]);          // plonk_zs_next
        for (size_t i = 0; i < (size_t)nch * npp; i++) w.ext(o_pp[i]);   // partial_products
        for (size_t i = 0; i < nq; i++) w.ext(o_q[i]);            // quotient_polys (lookup vectors are empty)
    }

    // ---- s8..s11 PolynomialBatch::prove_openings: every polynomial at zeta, the Zs also at g*zeta ----
    const PolyOracle *bs[4] = {&c->cs, &c->wires, &c->zs, &c->quot};
    std::vector<FriBatch> batches(2);
    batches[0].points = zetas;
    batches[0].ranges = {{0, 0, (uint32_t)ncs}, {1, 0, (uint32_t)NW}, {2, 0, (uint32_t)nzp}, {3, 0, (uint32_t)nq}};
    batches[1].points = g_zetas;
    batches[1].ranges = {{2, 0, nch}};
    QP_TRY(fri_prove(ctx, c->fri, bs, 4, batches, nb, chs.data(), c->fri_work, c->stage, ws.data()));
    bool overflow = false;
    for (uint32_t b = 0; b < nb; b++) {
        ByteWriter &w = ws[b];
        for (size_t i = 0; i < p.num_public_inputs; i++) w.u64le(gl::canon(public_inputs[b][i]));
        if (out_lens) out_lens[b] = w.len;
        overflow = overflow || w.overflow;
    }
    if (overflow) return ctx->fail(QPGPU_EBUFSIZE, "prove: output buffer too small");
    return QPGPU_OK;
}

extern "C" {

int qpgpu_prove_batch_dev(qpgpu_circuit *c, const uint64_t *const *d_wires, uint32_t batch, const uint64_t *const *public_inputs,
                          uint8_t *const *outs, size_t out_cap, size_t *out_lens) {
    if (!c) return QPGPU_EINVAL;
    qpgpu_ctx *ctx = c->ctx;
    QP_DEV(ctx);
    if (batch == 0 || batch > c->max_batch) return ctx->fail(QPGPU_EINVAL, "prove_batch: batch size outside 1..max_batch of the circuit handle");
    if (!d_wires || !outs || (!public_inputs && c->pack.num_public_inputs)) return ctx->fail(QPGPU_EINVAL, "prove_batch: null argument");
    const size_t mat = (size_t)c->pack.num_wires * c->pack.n();
    bool dense = true;
    for (uint32_t b = 0; b < batch; b++) {
        if (!d_wires[b] || !outs[b] || (c->pack.num_public_inputs && !public_inputs[b])) return ctx->fail(QPGPU_EINVAL, "prove_batch: null argument");
        dense = dense && d_wires[b] == d_wires[0] + b * mat;
    }
    const u64 *w = d_wires[0];
    if (!dense) {   // witnesses scattered over the heap: gather them into the workspace (8.8 MB per leaf-sized witness, ~2 us each)
        for (uint32_t b = 0; b < batch; b++) QP_HIP(ctx, hipMemcpyAsync(c->d_wires_vals + b * mat, d_wires[b], mat * 8, hipMemcpyDeviceToDevice, ctx->stream));
        w = c->d_wires_vals;
    }
    static const u64 *no_pis[1024] = {nullptr};
    return prove_batch_impl(c, batch, w, c->pack.num_public_inputs ? public_inputs : no_pis, outs, out_cap, out_lens);
}

int qpgpu_prove_dev(qpgpu_circuit *c, const uint64_t *d_wires, const uint64_t *public_inputs, uint8_t *out, size_t out_cap, size_t *out_len) {
    if (!c) return QPGPU_EINVAL;
    QP_DEV(c->ctx);
    if (!d_wires || (!public_inputs && c->pack.num_public_inputs) || !out) return c->ctx->fail(QPGPU_EINVAL, "prove: null argument");
    return prove_batch_impl(c, 1, d_wires, &public_inputs, &out, out_cap, out_len);
}

int qpgpu_prove(qpgpu_circuit *c, const uint64_t *wires, const uint64_t *public_inputs, uint8_t *out, size_t out_cap, size_t *out_len) {
    if (!c) return QPGPU_EINVAL;
    QP_DEV(c->ctx);
    if (!wires || (!public_inputs && c->pack.num_public_inputs) || !out) return c->ctx->fail(QPGPU_EINVAL, "prove: null argument");
    const size_t bytes = (size_t)c->pack.num_wires * c->pack.n() * 8;
    QP_TRY(h2d(c->ctx, c->d_wires_vals, wires, bytes));
    int rc = prove_batch_impl(c, 1, c->d_wires_vals, &public_inputs, &out, out_cap, out_len);
    // the witness carries the spend secret (reference wormhole/circuit/src/sensitive.rs:36-44): scrub every device region
    // derived from it (witness, Z / quotient values, coefficients, LDEs, salts, FRI workspace), on success and on failure
    std::string keep = c->ctx->err;
    scrub(c);
    c->ctx->err = keep;
    return rc;
}

}  // extern "C"
