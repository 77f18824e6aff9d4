// fri_prover.cpp — committed polynomial batches and the FRI opening prover (stages s2/s3 and s8..s11 of
// SURVEY.md §8a), independent of any circuit: used by the all-in-one prover (prover.cpp) and exported stage by stage
// through the C ABI (oracle_api.cpp). Follows qp-plonky2 1.5.5 `fri::oracle::PolynomialBatch::{from_values,
// from_coeffs, prove_openings}` and `fri::prover::{fri_proof, fri_committed_trees, fri_proof_of_work,
// fri_prover_query_rounds}` as reached from the reference's `prove()` call (wormhole/prover/src/lib.rs:171-175).
//
// All of it runs on a lockstep batch of nb proofs (prover_host.hpp): one launch per stage, nb transcripts on the host.
#include <hip/hip_runtime.h>
#include <sys/random.h>
#include <algorithm>
#include <string>
#include "merkle.hpp"
#include "prover_host.hpp"
#include "prover_kernels.hpp"

using gl::e2;
using gl::u64;

int Stager::put(qpgpu_ctx *ctx, void *dst, const void *src, size_t bytes) {
    const size_t w = (bytes + 7) / 8;
    if (!h || pos + w > words) {   // not sized for this table: through the context's pinned bounce buffer, synchronously
        QP_TRY(ctx->reserve_read_back(bytes));
        QP_HIP(ctx, hipStreamSynchronize(ctx->stream));   // nothing queued earlier still reads the bounce buffer
        std::memcpy(ctx->h_pin, src, bytes);
        QP_HIP(ctx, pk_copy(dst, ctx->h_pin, bytes, ctx->stream));
        QP_HIP(ctx, hipStreamSynchronize(ctx->stream));   // the bounce buffer is shared with read_back
        return QPGPU_OK;
    }
    u64 *slot = h + pos;
    pos += w;
    std::memcpy(slot, src, bytes);
    QP_HIP(ctx, pk_copy(dst, slot, bytes, ctx->stream));   // the device reads the pinned slot in place (see ctx.hpp: read_back)
    return QPGPU_OK;
}

int Stager::put_rows(qpgpu_ctx *ctx, u64 *dst, size_t dst_pitch, const u64 *src, size_t src_pitch, size_t row_words, size_t rows) {
    const size_t w = row_words * rows;
    if (rows <= 1 || !h || pos + w > words) {       // one row, or no room in the ring: row by row through put()
        for (size_t r = 0; r < rows; r++) QP_TRY(put(ctx, dst + r * dst_pitch, src + r * src_pitch, row_words * 8));
        return QPGPU_OK;
    }
    u64 *slot = h + pos;
    pos += w;
    for (size_t r = 0; r < rows; r++) std::memcpy(slot + r * row_words, src + r * src_pitch, row_words * 8);
    QP_HIP(ctx, pk_unpack_rows(slot, dst_pitch, row_words, rows, dst, ctx->stream));
    return QPGPU_OK;
}

int salt_key_random(uint32_t key[8]) {
    size_t got = 0;
    while (got < 32) {
        const ssize_t r = getrandom((uint8_t *)key + got, 32 - got, 0);
        if (r <= 0) return QPGPU_EDEVICE;
        got += (size_t)r;
    }
    return QPGPU_OK;
}
void salt_key_from_seed(uint64_t seed, uint32_t key[8]) {
    key[0] = (uint32_t)seed; key[1] = (uint32_t)(seed >> 32);
    key[2] = 0x51504750u; key[3] = 0x53414c54u;   // "QPGP" "SALT"
    key[4] = key[5] = key[6] = key[7] = 0;
}

int oracle_commit_coeffs(qpgpu_ctx *ctx, PolyOracle &o, const uint32_t *d_keys) {
    const unsigned L = o.log_lde();
    const uint32_t nb = o.nb;
    if (nb > 1 && (o.ps_coeffs != ((u64)o.ncols << o.log_n) || o.ps_lde != ((u64)o.ncols << L)))
        return ctx->fail(QPGPU_EINVAL, "oracle_commit: batched oracles are dense arrays");
    // the nb * ncols columns of the batch are one column batch for the transform
    QP_TRY(ntt_run(ctx, o.coeffs, o.lde, o.log_n, L, (size_t)nb * o.ncols, false, true, gl::MULT_GEN));
    MerkleLeafArgs a{};
    a.src0 = o.lde; a.stride0 = 1ull << L; a.ncols0 = o.ncols; a.n_leaves = 1ull << L; a.digests = o.digests;
    a.batch = nb; a.ps_src0 = o.ps_lde; a.ps_digests = o.ps_digests;
    if (o.salt) {
        if (!d_keys) return ctx->fail(QPGPU_EINVAL, "oracle_commit: a blinded oracle needs salt keys");
        QP_HIP(ctx, pk_salt(d_keys, o.oracle_index, 1ull << L, o.salt, nb, ctx->stream));
        a.src1 = o.salt; a.stride1 = 1ull << L; a.ncols1 = 4; a.ps_src1 = o.ps_salt;
    }
    QP_TRY(merkle_build(ctx, a, L, o.cap_h, o.digests));
    const size_t total = digest_words(L, o.cap_h), cw = o.cap_words();
    o.cap.resize(cw * nb);
    QP_TRY(ctx->read_back_2d(o.cap.data(), o.digests + total - cw, o.ps_digests * 8, cw * 8, nb));
    return QPGPU_OK;
}

int oracle_commit_values(qpgpu_ctx *ctx, const u64 *d_values, PolyOracle &o, const uint32_t *d_keys) {
    QP_TRY(ntt_run(ctx, d_values, o.coeffs, o.log_n, o.log_n, (size_t)o.nb * o.ncols, true, false, 0));
    return oracle_commit_coeffs(ctx, o, d_keys);
}

namespace {
struct Layout {
    size_t comp, fin, vals, coeffs0, coeffs1, gather, alpha, ws;
    std::vector<size_t> digests, leafrows;
    size_t gather_words;
    // batch-level tables, offsets from the end of the per-proof blocks
    size_t t_points, t_shifts, t_betas, pow_states, pow_bases, pow_results, qidx, total;
};
Layout layout(const FriParams &p, const std::vector<size_t> &leaf_widths, size_t max_batch_polys, uint32_t nb) {
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
    l.gather = take(l.gather_words); l.alpha = take(2 * max_batch_polys);
    l.ws = off;
    off = l.ws * nb;
    l.t_points = take(2 * (size_t)nb); l.t_shifts = take(2 * (size_t)nb); l.t_betas = take(2 * (size_t)nb);
    l.pow_states = take(12 * (size_t)nb); l.pow_bases = take(nb); l.pow_results = take(nb); l.qidx = take((size_t)nb * p.num_queries);
    l.total = off;
    return l;
}
}  // namespace

size_t FriWork::words(const FriParams &p, const std::vector<size_t> &leaf_widths, size_t max_batch_polys, uint32_t nb) {
    return layout(p, leaf_widths, max_batch_polys, nb).total;
}
void FriWork::bind(u64 *base, const FriParams &p, const std::vector<size_t> &leaf_widths, size_t mbp, uint32_t nb) {
    const Layout l = layout(p, leaf_widths, mbp, nb);
    ws = l.ws; nb_cap = nb;
    comp = base + l.comp; fin = base + l.fin; vals = base + l.vals; coeffs[0] = base + l.coeffs0; coeffs[1] = base + l.coeffs1;
    digests.clear(); leafrows.clear();
    for (size_t o : l.digests) digests.push_back(base + o);
    for (size_t o : l.leafrows) leafrows.push_back(base + o);
    gather = base + l.gather; alpha_ext = (e2 *)(base + l.alpha);
    gather_words = l.gather_words; max_batch_polys = mbp;
    t_points = (e2 *)(base + l.t_points); t_shifts = (e2 *)(base + l.t_shifts); t_betas = (e2 *)(base + l.t_betas);
    pow_states = base + l.pow_states; pow_bases = base + l.pow_bases; pow_results = base + l.pow_results; qidx = base + l.qidx;
}

size_t fri_proof_bytes(const FriParams &p, const std::vector<size_t> &leaf_widths) {
    const size_t cap = (1ull << p.cap_h) * 32, L = p.degree_bits + p.rate_bits;
    size_t sz = 0, q = 0, lvl = L, fin = p.degree_bits;
    for (size_t w : leaf_widths) q += w * 8 + 1 + (L - p.cap_h) * 32;
    for (unsigned a : p.arity_bits) { sz += cap; lvl -= a; fin -= a; q += (16ull << a) + 1 + (lvl - p.cap_h) * 32; }
    return sz + p.num_queries * q + (16ull << fin) + 8;
}

int fri_prove(qpgpu_ctx *ctx, const FriParams &p, const PolyOracle *const *oracles, size_t n_oracles,
              const std::vector<FriBatch> &batches, uint32_t nb, Challenger *chs, FriWork &w, Stager &stage, ByteWriter *outs) {
    hipStream_t st = ctx->stream;
    const HasherDev hd = ctx->hasher_dev();
    const unsigned d = p.degree_bits, L = d + p.rate_bits, cap_h = p.cap_h;
    const u64 n = 1ull << d, lde_n = n << p.rate_bits;
    const size_t cap_words = (1ull << cap_h) * 4;
    if (nb == 0 || nb > w.nb_cap) return ctx->fail(QPGPU_EINVAL, "fri_prove: batch larger than the workspace");
    for (size_t i = 0; i < n_oracles; i++) {
        if (oracles[i]->log_n != d || oracles[i]->rate_bits != p.rate_bits || oracles[i]->cap_h != cap_h)
            return ctx->fail(QPGPU_EINVAL, "fri_prove: oracle " + std::to_string(i) + " does not match the FRI parameters");
        if (oracles[i]->nb != 1 && oracles[i]->nb < nb) return ctx->fail(QPGPU_EINVAL, "fri_prove: oracle committed for fewer proofs than the batch");
    }
    { unsigned tot = 0; for (unsigned a : p.arity_bits) { tot += a; if (a == 0 || a > 8) return ctx->fail(QPGPU_EINVAL, "fri_prove: bad reduction arity"); }
      if (tot > d || L - tot < cap_h) return ctx->fail(QPGPU_EINVAL, "fri_prove: reduction schedule does not fit the degree / cap height"); }
    if (p.pow_bits > 40) return ctx->fail(QPGPU_EINVAL, "fri_prove: proof_of_work_bits above 40");

    std::vector<e2> fri_alpha(nb);
    for (uint32_t b = 0; b < nb; b++) fri_alpha[b] = chs[b].get_ext();

    // ---- s8 batched opening polynomial: final = sum over batches, each shifted by alpha^(#polys of the later ones) ----
    size_t max_count = 0;
    for (const FriBatch &fb : batches) {
        if (fb.ranges.empty() || fb.ranges.size() > 8) return ctx->fail(QPGPU_EINVAL, "fri_prove: a batch needs 1..8 polynomial ranges");
        if (fb.points.size() != nb) return ctx->fail(QPGPU_EINVAL, "fri_prove: one opening point per proof expected");
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
    {   // alpha powers always start at 1: one table per proof, every batch reads a prefix
        std::vector<e2> apw((size_t)nb * max_count);
        for (uint32_t b = 0; b < nb; b++) {
            e2 a = gl::e2_from(1);
            for (size_t i = 0; i < max_count; i++) { apw[(size_t)b * max_count + i] = gl::e2_canon(a); a = gl::e2_mul(a, fri_alpha[b]); }
        }
        QP_TRY(stage.put_rows(ctx, (u64 *)w.alpha_ext, w.ws, (const u64 *)apw.data(), 2 * max_count, 2 * max_count, nb));
    }
    for (size_t bi = 0; bi < batches.size(); bi++) {
        const FriBatch &fb = batches[bi];
        ReduceArgs ra{};
        size_t count = 0;
        for (size_t r = 0; r < fb.ranges.size(); r++) {
            const FriRange &rg = fb.ranges[r];
            const PolyOracle *o = oracles[rg.oracle];
            ra.src[r] = o->coeffs + (size_t)rg.first * n; ra.ps_src[r] = o->ps_coeffs; ra.ncols[r] = rg.count;
            count += rg.count;
        }
        ra.nsrc = (uint32_t)fb.ranges.size();
        ra.alpha_pows = w.alpha_ext; ra.comp_a = w.comp; ra.comp_b = w.comp + n; ra.n = n;
        ra.batch = nb; ra.ps_alpha = w.ws / 2; ra.ps_comp = w.ws;      // alpha table in e2 units (ws is even)
        QP_HIP(ctx, pk_reduce_polys(ra, st));
        // alpha.shift_poly(final) multiplies what is there by alpha^count of THIS batch, then the quotient is added
        std::vector<e2> pts(nb), shifts(nb);
        for (uint32_t b = 0; b < nb; b++) {
            pts[b] = gl::e2_canon(fb.points[b]);
            shifts[b] = bi == 0 ? gl::e2_from(1) : gl::e2_canon(gl::e2_pow(fri_alpha[b], count));
        }
        QP_TRY(stage.put(ctx, w.t_points, pts.data(), nb * sizeof(e2)));
        QP_TRY(stage.put(ctx, w.t_shifts, shifts.data(), nb * sizeof(e2)));
        QP_HIP(ctx, pk_divide_linear(w.comp, w.comp + n, n, w.t_points, w.t_shifts, bi == 0 ? 0 : 1, w.fin, w.fin + n, nb, w.ws, w.ws, st));
    }
    ctx->prof_end();

    // ---- s9 FRI commit phase ----
    ctx->prof_begin("prove_fri_commit");
    std::vector<std::vector<u64>> fri_caps;     // per round: [nb][cap_words]
    std::vector<unsigned> tree_log_leaves;
    u64 shift = gl::MULT_GEN;
    u64 *coef = w.fin;           // [2][valid] per proof
    u64 valid = n; unsigned log_len = L;
    size_t slot = 0;
    NttProofs np; np.nproofs = nb; np.in_ps = w.ws; np.out_ps = w.ws;
    // values of the first layer: LDE of the two component columns, leaf order
    QP_TRY(ntt_run(ctx, coef, w.vals, d, L, 2, false, true, shift, np));
    for (size_t r = 0; r < p.arity_bits.size(); r++) {
        const unsigned ab = p.arity_bits[r];
        const u64 len = 1ull << log_len, arity = 1ull << ab;
        const unsigned log_leaves = log_len - ab;
        u64 *rows = w.leafrows[r];
        QP_HIP(ctx, pk_interleave_ext(w.vals, w.vals + len, len, rows, nb, w.ws, w.ws, st));
        // leaves = chunks of `arity` extension values = 2*arity consecutive felts
        QP_HIP(ctx, merkle_leaf_hash_rows(rows, 1ull << log_leaves, (uint32_t)(2 * arity), w.digests[r], nb, w.ws, w.ws, hd, st));
        {
            u64 cnt = 1ull << log_leaves; u64 *lvl = w.digests[r];
            QP_HIP(ctx, merkle_reduce_to_cap(lvl, cnt, 1ull << cap_h, nb, w.ws, hd, st));
            while (cnt > (1ull << cap_h)) { lvl += cnt * 4; cnt >>= 1; }
            std::vector<u64> capv(cap_words * nb);
            QP_TRY(ctx->read_back_2d(capv.data(), lvl, w.ws * 8, cap_words * 8, nb));
            fri_caps.push_back(capv);
        }
        tree_log_leaves.push_back(log_leaves);
        std::vector<e2> betas(nb);
        for (uint32_t b = 0; b < nb; b++) {
            chs[b].observe(fri_caps.back().data() + (size_t)b * cap_words, cap_words);
            betas[b] = gl::e2_canon(chs[b].get_ext());
        }
        QP_TRY(stage.put(ctx, w.t_betas, betas.data(), nb * sizeof(e2)));
        const u64 new_valid = valid >> ab;
        u64 *ncoef = w.coeffs[slot]; slot ^= 1;
        QP_HIP(ctx, pk_fri_fold(coef, coef + valid, new_valid, (uint32_t)arity, w.t_betas, ncoef, ncoef + new_valid, nb, w.ws, w.ws, st));
        coef = ncoef; valid = new_valid; log_len -= ab;
        shift = gl::pow(shift, arity);
        if (r + 1 < p.arity_bits.size()) {
            unsigned lv = 0; while ((1ull << lv) < valid) lv++;
            QP_TRY(ntt_run(ctx, coef, w.vals, lv, log_len, 2, false, true, shift, np));
        }
    }
    std::vector<u64> final_coeffs(2 * valid * nb);   // per proof: component arrays [a...][b...]
    QP_TRY(ctx->read_back_2d(final_coeffs.data(), coef, w.ws * 8, 2 * valid * 8, nb));
    ctx->prof_end();
    std::vector<std::vector<e2>> final_poly(nb, std::vector<e2>(valid));
    for (uint32_t b = 0; b < nb; b++) {
        const u64 *fc = final_coeffs.data() + (size_t)b * 2 * valid;
        for (u64 i = 0; i < valid; i++) final_poly[b][i] = gl::e2_make(fc[i], fc[valid + i]);
        chs[b].observe((const u64 *)final_poly[b].data(), 2 * valid);
    }

    // ---- s10 proof of work: minimum nonce per proof ----
    ctx->prof_begin("prove_pow");
    std::vector<u64> pow_witness(nb, 0);
    if (p.pow_bits > 0) {
        std::vector<u64> states(12 * (size_t)nb);
        for (uint32_t b = 0; b < nb; b++) {
            std::memcpy(states.data() + 12 * (size_t)b, chs[b].state, 12 * 8);
            for (int i = 0; i < chs[b].n_in; i++) states[12 * (size_t)b + i] = chs[b].in[i];
            if (chs[b].n_in != chs[0].n_in) return ctx->fail(QPGPU_EDEVICE, "prove: transcripts of a batch diverged in length");
        }
        QP_TRY(stage.put(ctx, w.pow_states, states.data(), states.size() * 8));
        PowArgs pw{};
        pw.states = w.pow_states; pw.bases = w.pow_bases; pw.results = w.pow_results;
        pw.pos = (uint32_t)chs[0].n_in; pw.pow_bits = p.pow_bits; pw.batch = nb;
        // expected 2^pow_bits candidates per proof; a span of 4x that finds a proof's nonce 98 % of the time (all 32 of a batch
        // in a little over half of the batches; the rest take a second, nearly empty launch), and the kernel drops the candidates
        // above a nonce already found, so the span costs little beyond the expected work
        const u64 span = std::max<u64>(1ull << 16, 4ull << pw.pow_bits);
        pw.count = span;
        std::vector<u64> bases(nb, 0), res(nb);
        std::vector<char> found(nb, 0);
        uint32_t n_found = 0;
        QP_HIP(ctx, hipMemsetAsync(w.pow_results, 0xFF, 8 * (size_t)nb, st));
        const size_t stage_pos = stage.pos;
        for (u64 round = 0; n_found < nb; round++) {
            stage.pos = stage_pos;   // the previous round's upload has completed (read_back below syncs): its slot is free again
            for (uint32_t b = 0; b < nb; b++) bases[b] = found[b] ? ~0ull : round * span;
            QP_TRY(stage.put(ctx, w.pow_bases, bases.data(), 8 * (size_t)nb));
            QP_HIP(ctx, pk_pow(pw, hd, st));
            QP_TRY(ctx->read_back(res.data(), w.pow_results, 8 * (size_t)nb));
            for (uint32_t b = 0; b < nb; b++)
                if (!found[b] && res[b] != ~0ull) { pow_witness[b] = res[b]; found[b] = 1; n_found++; }
            if (round * span > (1ull << 41)) return ctx->fail(QPGPU_EDEVICE, "prove: proof of work not found");
        }
    }
    ctx->prof_end();
    for (uint32_t b = 0; b < nb; b++) {
        chs[b].observe(&pow_witness[b], 1);
        (void)chs[b].get();   // the response, re-derived by the verifier
    }

    // ---- s11 queries ----
    ctx->prof_begin("prove_queries");
    const uint32_t nqr = p.num_queries;
    std::vector<u64> qidx((size_t)nqr * nb);
    for (uint32_t b = 0; b < nb; b++)
        for (uint32_t q = 0; q < nqr; q++) qidx[(size_t)b * nqr + q] = chs[b].get() % lde_n;
    QP_TRY(stage.put(ctx, w.qidx, qidx.data(), qidx.size() * 8));
    // gather layout (per section, all queries contiguous): for each oracle rows then paths; for each FRI round evals then paths
    struct Sec { size_t off, words; bool is_path; };
    std::vector<Sec> secs;
    size_t goff = 0;
    const uint32_t plen0 = L - cap_h;
    for (size_t i = 0; i < n_oracles; i++) {
        const PolyOracle *o = oracles[i];
        QP_HIP(ctx, pk_gather_rows(o->lde, lde_n, o->ncols, w.qidx, nqr, w.gather + goff, nb, o->ps_lde, w.ws, st));
        secs.push_back({goff, o->ncols, false}); goff += (size_t)o->ncols * nqr;
        if (o->salt) {
            QP_HIP(ctx, pk_gather_rows(o->salt, lde_n, 4, w.qidx, nqr, w.gather + goff, nb, o->ps_salt, w.ws, st));
            secs.push_back({goff, 4, false}); goff += (size_t)4 * nqr;
        }
        QP_HIP(ctx, pk_gather_paths(o->digests, lde_n, plen0, w.qidx, 0, nqr, w.gather + goff, nb, o->ps_digests, w.ws, st));
        secs.push_back({goff, (size_t)plen0 * 4, true}); goff += (size_t)plen0 * 4 * nqr;
    }
    {
        uint32_t sh = 0;
        for (size_t r = 0; r < p.arity_bits.size(); r++) {
            const uint32_t ab = p.arity_bits[r], width = 2u << ab, pl = tree_log_leaves[r] - cap_h;
            sh += ab;
            QP_HIP(ctx, pk_gather_leaf_rows(w.leafrows[r], width, w.qidx, sh, nqr, w.gather + goff, nb, w.ws, w.ws, st));
            secs.push_back({goff, width, false}); goff += (size_t)width * nqr;
            QP_HIP(ctx, pk_gather_paths(w.digests[r], 1ull << tree_log_leaves[r], pl, w.qidx, sh, nqr, w.gather + goff, nb, w.ws, w.ws, st));
            secs.push_back({goff, (size_t)pl * 4, true}); goff += (size_t)pl * 4 * nqr;
        }
    }
    if (goff != w.gather_words) return ctx->fail(QPGPU_EDEVICE, "prove: internal gather size mismatch");
    std::vector<u64> gathered(goff * nb);
    QP_TRY(ctx->read_back_2d(gathered.data(), w.gather, w.ws * 8, goff * 8, nb));
    ctx->prof_end();

    // ---- FriProof bytes (util::serialization write_fri_proof): caps, query rounds, final poly, pow witness ----
    for (uint32_t b = 0; b < nb; b++) {
        ByteWriter &out = outs[b];
        const u64 *g = gathered.data() + (size_t)b * goff;
        for (auto &cp : fri_caps) out.vec(cp.data() + (size_t)b * cap_words, cap_words);
        for (uint32_t q = 0; q < nqr; q++) {
            for (const Sec &sc : secs) {
                if (sc.is_path) out.u8((uint8_t)(sc.words / 4));   // write_merkle_proof: one-byte sibling count
                out.vec(g + sc.off + (size_t)q * sc.words, sc.words);
            }
        }
        for (u64 i = 0; i < valid; i++) out.ext(final_poly[b][i]);
        out.u64le(pow_witness[b]);
    }
    return QPGPU_OK;
}
