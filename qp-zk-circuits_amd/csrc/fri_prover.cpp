// fri_prover.cpp — committed polynomial batches and the FRI opening prover (stages s2/s3 and s8..s11 of
// SURVEY.md §8a), independent of any circuit: used by the all-in-one prover (prover.cpp) and exported stage by stage
// through the C ABI (oracle_api.cpp). Follows qp-plonky2 1.5.5 `fri::oracle::PolynomialBatch::{from_values,
// from_coeffs, prove_openings}` and `fri::prover::{fri_proof, fri_committed_trees, fri_proof_of_work,
// fri_prover_query_rounds}` as reached from the reference's `prove()` call (wormhole/prover/src/lib.rs:171-175).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <string>
#include "merkle.hpp"
#include "prover_host.hpp"
#include "prover_kernels.hpp"

using gl::e2;
using gl::u64;

int Stager::put(qpgpu_ctx *ctx, void *dst, const void *src, size_t bytes) {
    const size_t w = (bytes + 7) / 8;
    if (!h || pos + w > words) {   // not sized for this table: fall back to a synchronous upload
        QP_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
        QP_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return QPGPU_OK;
    }
    u64 *slot = h + pos;
    pos += w;
    std::memcpy(slot, src, bytes);
    QP_HIP(ctx, hipMemcpyAsync(dst, slot, bytes, hipMemcpyHostToDevice, ctx->stream));
    return QPGPU_OK;
}

int oracle_commit_coeffs(qpgpu_ctx *ctx, PolyOracle &o, u64 blinding_seed) {
    const unsigned L = o.log_lde();
    QP_TRY(ntt_run(ctx, o.coeffs, o.lde, o.log_n, L, o.ncols, false, true, gl::MULT_GEN));
    MerkleLeafArgs a{};
    a.src0 = o.lde; a.stride0 = 1ull << L; a.ncols0 = o.ncols; a.n_leaves = 1ull << L; a.digests = o.digests;
    if (o.salt) {
        QP_HIP(ctx, pk_salt(blinding_seed, o.oracle_index, 1ull << L, o.salt, ctx->stream));
        a.src1 = o.salt; a.stride1 = 1ull << L; a.ncols1 = 4;
    }
    QP_TRY(merkle_build(ctx, a, L, o.cap_h, o.digests));
    const size_t total = digest_words(L, o.cap_h);
    o.cap.resize((1ull << o.cap_h) * 4);
    QP_TRY(ctx->read_back(o.cap.data(), o.digests + total - o.cap.size(), o.cap.size() * 8));
    return QPGPU_OK;
}

int oracle_commit_values(qpgpu_ctx *ctx, const u64 *d_values, PolyOracle &o, u64 blinding_seed) {
    QP_TRY(ntt_run(ctx, d_values, o.coeffs, o.log_n, o.log_n, o.ncols, true, false, 0));
    return oracle_commit_coeffs(ctx, o, blinding_seed);
}

namespace {
struct Layout {
    size_t comp, fin, vals, coeffs0, coeffs1, pow, qidx, gather, alpha, total;
    std::vector<size_t> digests, leafrows;
    size_t gather_words;
};
Layout layout(const FriParams &p, const std::vector<size_t> &leaf_widths, size_t max_batch_polys) {
    Layout l;
    const size_t n = 1ull << p.degree_bits, lde_n = n << p.rate_bits;
    const unsigned L = p.degree_bits + p.rate_bits;
    size_t off = 0;
    auto take = [&](size_t words) { size_t o = off; off += (words + 1) & ~(size_t)1; return o; };   // 16-byte aligned
    l.comp = take(2 * n); l.fin = take(2 * n); l.vals = take(2 * lde_n); l.coeffs0 = take(2 * n); l.coeffs1 = take(2 * n);
    size_t gw = 0;
    for (size_t w : leaf_widths) gw += w + (size_t)(L - p.cap_h) * 4;
    unsigned lvl = L;
    for (unsigned a : p.arity_bits) {
        lvl -= a;
        l.digests.push_back(take(digest_words(lvl, p.cap_h)));
        l.leafrows.push_back(take((size_t)2 << (lvl + a)));
        gw += (2ull << a) + (size_t)(lvl - p.cap_h) * 4;
    }
    l.gather_words = gw * p.num_queries;
    l.pow = take(2); l.qidx = take(p.num_queries); l.gather = take(l.gather_words); l.alpha = take(2 * max_batch_polys);
    l.total = off;
    return l;
}
}  // namespace

size_t FriWork::words(const FriParams &p, const std::vector<size_t> &leaf_widths, size_t max_batch_polys) {
    return layout(p, leaf_widths, max_batch_polys).total;
}
void FriWork::bind(u64 *base, const FriParams &p, const std::vector<size_t> &leaf_widths, size_t mbp) {
    const Layout l = layout(p, leaf_widths, mbp);
    comp = base + l.comp; fin = base + l.fin; vals = base + l.vals; coeffs[0] = base + l.coeffs0; coeffs[1] = base + l.coeffs1;
    digests.clear(); leafrows.clear();
    for (size_t o : l.digests) digests.push_back(base + o);
    for (size_t o : l.leafrows) leafrows.push_back(base + o);
    pow = base + l.pow; qidx = base + l.qidx; gather = base + l.gather; alpha_ext = (e2 *)(base + l.alpha);
    gather_words = l.gather_words; max_batch_polys = mbp;
}

size_t fri_proof_bytes(const FriParams &p, const std::vector<size_t> &leaf_widths) {
    const size_t cap = (1ull << p.cap_h) * 32, L = p.degree_bits + p.rate_bits;
    size_t sz = 0, q = 0, lvl = L, fin = p.degree_bits;
    for (size_t w : leaf_widths) q += w * 8 + 1 + (L - p.cap_h) * 32;
    for (unsigned a : p.arity_bits) { sz += cap; lvl -= a; fin -= a; q += (16ull << a) + 1 + (lvl - p.cap_h) * 32; }
    return sz + p.num_queries * q + (16ull << fin) + 8;
}

int fri_prove(qpgpu_ctx *ctx, const FriParams &p, const PolyOracle *const *oracles, size_t n_oracles,
              const std::vector<FriBatch> &batches, Challenger &ch, FriWork &w, Stager &stage, ByteWriter &out) {
    hipStream_t st = ctx->stream;
    const unsigned d = p.degree_bits, L = d + p.rate_bits, cap_h = p.cap_h;
    const u64 n = 1ull << d, lde_n = n << p.rate_bits;
    const size_t cap_words = (1ull << cap_h) * 4;
    for (size_t i = 0; i < n_oracles; i++)
        if (oracles[i]->log_n != d || oracles[i]->rate_bits != p.rate_bits || oracles[i]->cap_h != cap_h)
            return ctx->fail(QPGPU_EINVAL, "fri_prove: oracle " + std::to_string(i) + " does not match the FRI parameters");
    { unsigned tot = 0; for (unsigned a : p.arity_bits) { tot += a; if (a == 0 || a > 8) return ctx->fail(QPGPU_EINVAL, "fri_prove: bad reduction arity"); }
      if (tot > d || L - tot < cap_h) return ctx->fail(QPGPU_EINVAL, "fri_prove: reduction schedule does not fit the degree / cap height"); }

    const e2 fri_alpha = ch.get_ext();

    // ---- s8 batched opening polynomial: final = sum over batches, each shifted by alpha^(#polys of the later ones) ----
    size_t max_count = 0;
    for (const FriBatch &fb : batches) {
        if (fb.ranges.empty() || fb.ranges.size() > 8) return ctx->fail(QPGPU_EINVAL, "fri_prove: a batch needs 1..8 polynomial ranges");
        size_t cnt = 0;
        for (const FriRange &rg : fb.ranges) {
            if (rg.oracle >= n_oracles || (size_t)rg.first + rg.count > oracles[rg.oracle]->ncols || rg.count == 0)
                return ctx->fail(QPGPU_EINVAL, "fri_prove: polynomial range outside its oracle");
            cnt += rg.count;
        }
        max_count = std::max(max_count, cnt);
    }
    if (max_count == 0 || max_count > w.max_batch_polys) return ctx->fail(QPGPU_EINVAL, "fri_prove: empty batch or workspace too small");
    ctx->prof_begin("prove_fri_batch");
    {   // alpha powers always start at 1: one table, every batch reads a prefix
        std::vector<e2> apw(max_count);
        e2 a = gl::e2_from(1);
        for (size_t i = 0; i < max_count; i++) { apw[i] = gl::e2_canon(a); a = gl::e2_mul(a, fri_alpha); }
        QP_TRY(stage.put(ctx, w.alpha_ext, apw.data(), apw.size() * sizeof(e2)));
    }
    for (size_t b = 0; b < batches.size(); b++) {
        const FriBatch &fb = batches[b];
        ReduceArgs ra{};
        size_t count = 0;
        for (size_t r = 0; r < fb.ranges.size(); r++) {
            const FriRange &rg = fb.ranges[r];
            ra.src[r] = oracles[rg.oracle]->coeffs + (size_t)rg.first * n; ra.ncols[r] = rg.count;
            count += rg.count;
        }
        ra.nsrc = (uint32_t)fb.ranges.size();
        ra.alpha_pows = w.alpha_ext; ra.comp_a = w.comp; ra.comp_b = w.comp + n; ra.n = n;
        QP_HIP(ctx, pk_reduce_polys(ra, st));
        // alpha.shift_poly(final) multiplies what is there by alpha^count of THIS batch, then the quotient is added
        QP_HIP(ctx, pk_divide_linear(w.comp, w.comp + n, n, gl::e2_canon(fb.point), b == 0 ? gl::e2_from(1) : gl::e2_canon(gl::e2_pow(fri_alpha, count)),
                                     b == 0 ? 0 : 1, w.fin, w.fin + n, st));
    }
    ctx->prof_end();

    // ---- s9 FRI commit phase ----
    ctx->prof_begin("prove_fri_commit");
    std::vector<std::vector<u64>> fri_caps;
    std::vector<unsigned> tree_log_leaves;
    u64 shift = gl::MULT_GEN;
    u64 *coef = w.fin;           // [2][valid]
    u64 valid = n; unsigned log_len = L;
    size_t slot = 0;
    // values of the first layer: LDE of the two component columns, leaf order
    QP_TRY(ntt_run(ctx, coef, w.vals, d, L, 2, false, true, shift));
    for (size_t r = 0; r < p.arity_bits.size(); r++) {
        const unsigned ab = p.arity_bits[r];
        const u64 len = 1ull << log_len, arity = 1ull << ab;
        const unsigned log_leaves = log_len - ab;
        u64 *rows = w.leafrows[r];
        QP_HIP(ctx, pk_interleave_ext(w.vals, w.vals + len, len, rows, st));
        // leaves = chunks of `arity` extension values = 2*arity consecutive felts
        QP_HIP(ctx, merkle_leaf_hash_rows(rows, 1ull << log_leaves, (uint32_t)(2 * arity), w.digests[r], st));
        {
            u64 cnt = 1ull << log_leaves; u64 *lvl = w.digests[r];
            QP_HIP(ctx, merkle_reduce_to_cap(lvl, cnt, 1ull << cap_h, st));
            while (cnt > (1ull << cap_h)) { lvl += cnt * 4; cnt >>= 1; }
            std::vector<u64> capv(cap_words);
            QP_TRY(ctx->read_back(capv.data(), lvl, cap_words * 8));
            fri_caps.push_back(capv);
        }
        tree_log_leaves.push_back(log_leaves);
        ch.observe(fri_caps.back().data(), cap_words);
        const e2 beta = ch.get_ext();
        const u64 new_valid = valid >> ab;
        u64 *ncoef = w.coeffs[slot]; slot ^= 1;
        QP_HIP(ctx, pk_fri_fold(coef, coef + valid, new_valid, (uint32_t)arity, beta, ncoef, ncoef + new_valid, st));
        coef = ncoef; valid = new_valid; log_len -= ab;
        shift = gl::pow(shift, arity);
        if (r + 1 < p.arity_bits.size()) {
            unsigned lv = 0; while ((1ull << lv) < valid) lv++;
            QP_TRY(ntt_run(ctx, coef, w.vals, lv, log_len, 2, false, true, shift));
        }
    }
    std::vector<u64> final_coeffs(2 * valid);   // component arrays [a...][b...]
    QP_TRY(ctx->read_back(final_coeffs.data(), coef, final_coeffs.size() * 8));
    ctx->prof_end();
    std::vector<e2> final_poly(valid);
    for (u64 i = 0; i < valid; i++) final_poly[i] = gl::e2_make(final_coeffs[i], final_coeffs[valid + i]);
    ch.observe((const u64 *)final_poly.data(), 2 * valid);

    // ---- s10 proof of work: minimum nonce ----
    ctx->prof_begin("prove_pow");
    u64 pow_witness = 0;
    if (p.pow_bits > 0) {
        PowArgs pw{};
        std::memcpy(pw.state, ch.state, sizeof pw.state);
        for (int i = 0; i < ch.n_in; i++) pw.state[i] = ch.in[i];
        pw.pos = (uint32_t)ch.n_in; pw.pow_bits = p.pow_bits; pw.result = w.pow;
        // expected 2^pow_bits candidates; a batch of 2x that finds it 86% of the time and costs one wave per SIMD
        const u64 batch = std::max<u64>(1ull << 16, 2ull << pw.pow_bits);
        bool found = false;
        for (u64 base = 0; !found; base += batch) {
            const u64 sentinel = ~0ull;
            QP_HIP(ctx, hipMemsetAsync(w.pow, 0xFF, 8, st));
            pw.base = base; pw.count = batch;
            QP_HIP(ctx, pk_pow(pw, st));
            u64 res = 0;
            QP_TRY(ctx->read_back(&res, w.pow, 8));
            if (res != sentinel) { pow_witness = res; found = true; }
            if (base > (1ull << 40)) return ctx->fail(QPGPU_EDEVICE, "prove: proof of work not found");
        }
    }
    ctx->prof_end();
    ch.observe(&pow_witness, 1);
    (void)ch.get();   // the response, re-derived by the verifier

    // ---- s11 queries ----
    ctx->prof_begin("prove_queries");
    const uint32_t nqr = p.num_queries;
    std::vector<u64> qidx(nqr);
    for (auto &x : qidx) x = ch.get() % lde_n;
    QP_TRY(stage.put(ctx, w.qidx, qidx.data(), nqr * 8));
    // gather layout (per section, all queries contiguous): for each oracle rows then paths; for each FRI round evals then paths
    struct Sec { size_t off, words; bool is_path; };
    std::vector<Sec> secs;
    size_t goff = 0;
    const uint32_t plen0 = L - cap_h;
    for (size_t i = 0; i < n_oracles; i++) {
        const PolyOracle *b = oracles[i];
        QP_HIP(ctx, pk_gather_rows(b->lde, lde_n, b->ncols, w.qidx, nqr, w.gather + goff, st));
        secs.push_back({goff, b->ncols, false}); goff += (size_t)b->ncols * nqr;
        if (b->salt) {
            QP_HIP(ctx, pk_gather_rows(b->salt, lde_n, 4, w.qidx, nqr, w.gather + goff, st));
            secs.push_back({goff, 4, false}); goff += (size_t)4 * nqr;
        }
        QP_HIP(ctx, pk_gather_paths(b->digests, lde_n, plen0, w.qidx, 0, nqr, w.gather + goff, st));
        secs.push_back({goff, (size_t)plen0 * 4, true}); goff += (size_t)plen0 * 4 * nqr;
    }
    {
        uint32_t sh = 0;
        for (size_t r = 0; r < p.arity_bits.size(); r++) {
            const uint32_t ab = p.arity_bits[r], width = 2u << ab, pl = tree_log_leaves[r] - cap_h;
            sh += ab;
            QP_HIP(ctx, pk_gather_leaf_rows(w.leafrows[r], width, w.qidx, sh, nqr, w.gather + goff, st));
            secs.push_back({goff, width, false}); goff += (size_t)width * nqr;
            QP_HIP(ctx, pk_gather_paths(w.digests[r], 1ull << tree_log_leaves[r], pl, w.qidx, sh, nqr, w.gather + goff, st));
            secs.push_back({goff, (size_t)pl * 4, true}); goff += (size_t)pl * 4 * nqr;
        }
    }
    if (goff != w.gather_words) return ctx->fail(QPGPU_EDEVICE, "prove: internal gather size mismatch");
    std::vector<u64> gathered(goff);
    QP_TRY(ctx->read_back(gathered.data(), w.gather, goff * 8));
    ctx->prof_end();

    // ---- FriProof bytes (util::serialization write_fri_proof): caps, query rounds, final poly, pow witness ----
    for (auto &cp : fri_caps) out.vec(cp.data(), cap_words);
    for (uint32_t q = 0; q < nqr; q++) {
        for (const Sec &sc : secs) {
            if (sc.is_path) out.u8((uint8_t)(sc.words / 4));   // write_merkle_proof: one-byte sibling count
            out.vec(gathered.data() + sc.off + (size_t)q * sc.words, sc.words);
        }
    }
    for (u64 i = 0; i < valid; i++) out.ext(final_poly[i]);
    out.u64le(pow_witness);
    return QPGPU_OK;
}
