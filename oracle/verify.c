/*
 * oracle/verify.c — CPU restatement of qp-plonky2's verifier (plonk::verifier::verify_with_challenges and
 * fri::verifier::verify_fri_proof) over a circuit pack. This is the acceptance test the reference itself
 * applies to proofs ("verifies": wormhole/tests/src/prover/verifier_tests.rs:40-66,
 * wormhole/aggregator/src/aggregator.rs:224-225). TEST INFRASTRUCTURE ONLY.
 *
 * Return codes: 0 accepted; 1 size; 2 proof-of-work; 3 quotient identity; 4 initial Merkle path;
 * 5 FRI round consistency; 6 FRI round Merkle path; 7 final polynomial; 8 public inputs.
 */
#include "plonk.h"
#include "challenger.h"
#include <stdlib.h>
#include <string.h>

void orc_eval_gates_ext(const orc_circuit *c, const gl2_t *cs_row, const gl2_t *wires, const gl_t pih[4], gl2_t *acc);

typedef struct { const uint8_t *p; size_t len, pos; int bad; } rbuf;
static uint64_t r_u64(rbuf *b) { if (b->pos + 8 > b->len) { b->bad = 1; return 0; } uint64_t v = 0; for (int k = 0; k < 8; k++) v |= (uint64_t)b->p[b->pos + k] << (8 * k); b->pos += 8; return v; }
static uint8_t r_u8(rbuf *b) { if (b->pos + 1 > b->len) { b->bad = 1; return 0; } return b->p[b->pos++]; }
static void r_vec(rbuf *b, gl_t *out, size_t n) { for (size_t i = 0; i < n; i++) out[i] = r_u64(b); }
static gl2_t r_ext(rbuf *b) { gl_t a = r_u64(b), c = r_u64(b); return gl2_make(a, c); }

static int verify_path(const gl_t *leaf, size_t width, size_t index, const gl_t *path, size_t plen, const gl_t *cap, unsigned cap_h, size_t log_leaves) {
    if (plen != log_leaves - cap_h) return 0;
    gl_t cur[4], nxt[4];
    orc_hash_or_noop(leaf, width, cur);
    for (size_t i = 0; i < plen; i++) {
        if (index & 1) orc_two_to_one(path + 4 * i, cur, nxt); else orc_two_to_one(cur, path + 4 * i, nxt);
        memcpy(cur, nxt, sizeof cur); index >>= 1;
    }
    return memcmp(cur, cap + 4 * index, sizeof cur) == 0;
}
static gl2_t challenger_get_ext(orc_challenger *ch) { gl_t a = orc_challenger_get(ch), b = orc_challenger_get(ch); return gl2_make(a, b); }

int orc_verify(const orc_circuit *c, const uint8_t *proof, size_t len) {
    if (len != orc_proof_size(c)) return 1;
    const unsigned d = (unsigned)c->degree_bits, rb = (unsigned)c->rate_bits, ch_h = (unsigned)c->cap_height, L = d + rb;
    const size_t n = (size_t)1 << d, lde_n = n << rb, R = c->num_routed, NW = c->num_wires, nch = c->num_challenges;
    const size_t npp = c->num_pp, nchunks = npp + 1, chunk = c->qdf, ncs = c->num_selectors + c->num_constants + R;
    const size_t sig0 = c->num_selectors + c->num_constants, cap_words = ((size_t)1 << ch_h) * 4, nq = nch * c->qdf;
    int rc = 0;
    rbuf b = {proof, len, 0, 0};
    gl_t *wires_cap = malloc(8 * cap_words), *zs_cap = malloc(8 * cap_words), *q_cap = malloc(8 * cap_words);
    r_vec(&b, wires_cap, cap_words); r_vec(&b, zs_cap, cap_words); r_vec(&b, q_cap, cap_words);
    gl2_t *o_cs = malloc(sizeof(gl2_t) * ncs), *o_w = malloc(sizeof(gl2_t) * NW), *o_zs = malloc(sizeof(gl2_t) * nch),
          *o_zn = malloc(sizeof(gl2_t) * nch), *o_pp = malloc(sizeof(gl2_t) * (nch * npp + 1)), *o_q = malloc(sizeof(gl2_t) * nq);
    for (size_t i = 0; i < ncs; i++) o_cs[i] = r_ext(&b);
    for (size_t i = 0; i < NW; i++) o_w[i] = r_ext(&b);
    for (size_t i = 0; i < nch; i++) o_zs[i] = r_ext(&b);
    for (size_t i = 0; i < nch; i++) o_zn[i] = r_ext(&b);
    for (size_t i = 0; i < nch * npp; i++) o_pp[i] = r_ext(&b);
    for (size_t i = 0; i < nq; i++) o_q[i] = r_ext(&b);
    gl_t *fri_caps = malloc(8 * cap_words * (c->n_arity + 1));
    for (size_t r = 0; r < c->n_arity; r++) r_vec(&b, fri_caps + r * cap_words, cap_words);
    size_t queries_pos = b.pos;
    /* skip the query rounds to reach final_poly / pow / public inputs */
    const size_t salt = c->zk ? 4 : 0;
    size_t widths[4] = {ncs, NW + salt, nch * (1 + npp) + salt, nq + salt};
    const size_t polys[4] = {ncs, NW, nch * (1 + npp), nq};
    {
        size_t q = 0, lvl = L;
        for (int o = 0; o < 4; o++) q += widths[o] * 8 + 1 + (L - ch_h) * 32;
        for (size_t r = 0; r < c->n_arity; r++) { lvl -= c->arity[r]; q += ((size_t)1 << c->arity[r]) * 16 + 1 + (lvl - ch_h) * 32; }
        b.pos += q * c->num_queries;
    }
    size_t fin_bits = d; for (size_t r = 0; r < c->n_arity; r++) fin_bits -= c->arity[r];
    size_t final_len = (size_t)1 << fin_bits;
    gl2_t *final_poly = malloc(sizeof(gl2_t) * final_len);
    for (size_t i = 0; i < final_len; i++) final_poly[i] = r_ext(&b);
    gl_t pow_witness = r_u64(&b);
    gl_t *pis = malloc(8 * (c->num_pis + 1));
    r_vec(&b, pis, c->num_pis);
    if (b.bad || b.pos != len) { rc = 1; goto done; }
    for (size_t i = 0; i < c->num_pis; i++) if (pis[i] >= GL_P) { rc = 8; goto done; }

    /* ---- challenges ---- */
    gl_t pih[4];
    orc_hash_no_pad(pis, c->num_pis, pih);
    orc_challenger ch; orc_challenger_init(&ch);
    orc_challenger_observe(&ch, c->digest, 4);
    orc_challenger_observe(&ch, pih, 4);
    orc_challenger_observe(&ch, wires_cap, cap_words);
    gl_t betas[4], gammas[4], alphas[4];
    orc_challenger_get_n(&ch, betas, nch); orc_challenger_get_n(&ch, gammas, nch);
    orc_challenger_observe(&ch, zs_cap, cap_words);
    orc_challenger_get_n(&ch, alphas, nch);
    orc_challenger_observe(&ch, q_cap, cap_words);
    gl2_t zeta = challenger_get_ext(&ch);
    orc_challenger_observe(&ch, (gl_t *)o_cs, 2 * ncs); orc_challenger_observe(&ch, (gl_t *)o_w, 2 * NW);
    orc_challenger_observe(&ch, (gl_t *)o_zs, 2 * nch); orc_challenger_observe(&ch, (gl_t *)o_pp, 2 * nch * npp);
    orc_challenger_observe(&ch, (gl_t *)o_q, 2 * nq); orc_challenger_observe(&ch, (gl_t *)o_zn, 2 * nch);
    gl2_t fri_alpha = challenger_get_ext(&ch);
    gl2_t fri_betas[16];
    for (size_t r = 0; r < c->n_arity; r++) { orc_challenger_observe(&ch, fri_caps + r * cap_words, cap_words); fri_betas[r] = challenger_get_ext(&ch); }
    orc_challenger_observe(&ch, (gl_t *)final_poly, 2 * final_len);
    orc_challenger_observe(&ch, &pow_witness, 1);
    gl_t pow_resp = orc_challenger_get(&ch);
    if (c->pow_bits && (pow_resp >> (64 - c->pow_bits)) != 0) { rc = 2; goto done; }

    /* ---- quotient identity at zeta ---- */
    {
        gl2_t zeta_n = zeta; for (unsigned i = 0; i < d; i++) zeta_n = gl2_mul(zeta_n, zeta_n);
        gl2_t zh = gl2_sub(zeta_n, gl2_from(1));
        gl2_t l0 = gl2_mul(zh, gl2_inv(gl2_scale(gl2_sub(zeta, gl2_from(1)), (gl_t)n)));
        size_t nterms = nch + nch * nchunks + c->num_gate_constraints;
        gl2_t *terms = calloc(nterms, sizeof(gl2_t));
        size_t t = 0;
        for (size_t k = 0; k < nch; k++) terms[t++] = gl2_mul(l0, gl2_sub(o_zs[k], gl2_from(1)));
        for (size_t k = 0; k < nch; k++) {
            for (size_t cc = 0; cc < nchunks; cc++) {
                gl2_t prev = cc == 0 ? o_zs[k] : o_pp[k * npp + cc - 1];
                gl2_t next = cc == nchunks - 1 ? o_zn[k] : o_pp[k * npp + cc];
                gl2_t pn = gl2_from(1), pd = gl2_from(1);
                for (size_t j = cc * chunk; j < (cc + 1) * chunk && j < R; j++) {
                    gl2_t num = gl2_add(gl2_add(o_w[j], gl2_scale(zeta, gl_mul(betas[k], c->k_is[j]))), gl2_from(gammas[k]));
                    gl2_t den = gl2_add(gl2_add(o_w[j], gl2_scale(o_cs[sig0 + j], betas[k])), gl2_from(gammas[k]));
                    pn = gl2_mul(pn, num); pd = gl2_mul(pd, den);
                }
                terms[t++] = gl2_sub(gl2_mul(prev, pn), gl2_mul(next, pd));
            }
        }
        orc_eval_gates_ext(c, o_cs, o_w, pih, terms + t);
        for (size_t k = 0; k < nch && !rc; k++) {
            gl2_t acc = gl2_from(0);
            for (size_t j = nterms; j-- > 0;) acc = gl2_add(gl2_scale(acc, alphas[k]), terms[j]);
            gl2_t qv = gl2_from(0);
            for (size_t j = c->qdf; j-- > 0;) qv = gl2_add(gl2_mul(qv, zeta_n), o_q[k * c->qdf + j]);
            if (!gl2_eq(acc, gl2_mul(zh, qv))) rc = 3;
        }
        free(terms);
        if (rc) goto done;
    }

    /* ---- FRI ---- */
    {
        /* reduced openings per batch: sum_j values[j] * alpha^j, batch 0 in oracle order, batch 1 = zs_next */
        gl2_t red0 = gl2_from(0), red1 = gl2_from(0);
        {
            const gl2_t *parts[5] = {o_cs, o_w, o_zs, o_pp, o_q}; size_t lens[5] = {ncs, NW, nch, nch * npp, nq};
            for (int p = 5; p-- > 0;) for (size_t j = lens[p]; j-- > 0;) red0 = gl2_add(gl2_mul(red0, fri_alpha), parts[p][j]);
            for (size_t j = nch; j-- > 0;) red1 = gl2_add(gl2_mul(red1, fri_alpha), o_zn[j]);
        }
        size_t n0 = ncs + NW + nch * (1 + npp) + nq;
        gl2_t g_zeta = gl2_scale(zeta, gl_root_of_unity(d));
        const gl_t *caps0[4] = {c->cs.cap, wires_cap, zs_cap, q_cap};
        rbuf q = {proof, len, queries_pos, 0};
        gl_t *row = malloc(8 * (n0 + 64 + 16)), path[64 * 4];
        for (size_t qi = 0; qi < c->num_queries && !rc; qi++) {
            size_t x_index = (size_t)(orc_challenger_get(&ch) % lde_n);
            /* initial trees */
            gl_t *rows[4]; size_t off = 0;
            for (int o = 0; o < 4; o++) {
                rows[o] = row + off; r_vec(&q, rows[o], widths[o]); off += widths[o];
                size_t plen = r_u8(&q); if (plen > 60) { rc = 4; break; }
                r_vec(&q, path, plen * 4);
                if (!verify_path(rows[o], widths[o], x_index, path, plen, caps0[o], ch_h, L)) { rc = 4; break; }
            }
            if (rc) break;
            gl_t subgroup_x = gl_mul(GL_MULT_GEN, gl_pow(gl_root_of_unity(L), bitrev32((uint32_t)x_index, L)));
            /* fri_combine_initial */
            gl2_t e0 = gl2_from(0), e1 = gl2_from(0);
            /* unsalted evaluations, oracle by oracle in reverse (Horner in alpha) */
            for (int o = 3; o >= 0; o--) for (size_t j = polys[o]; j-- > 0;) e0 = gl2_add(gl2_mul(e0, fri_alpha), gl2_from(rows[o][j]));
            for (size_t j = nch; j-- > 0;) e1 = gl2_add(gl2_mul(e1, fri_alpha), gl2_from(rows[2][j]));
            gl2_t sx = gl2_from(subgroup_x);
            gl2_t sum = gl2_mul(gl2_sub(e0, red0), gl2_inv(gl2_sub(sx, zeta)));
            sum = gl2_mul(sum, gl2_pow(fri_alpha, nch));
            sum = gl2_add(sum, gl2_mul(gl2_sub(e1, red1), gl2_inv(gl2_sub(sx, g_zeta))));
            gl2_t old_eval = sum;
            size_t lvl = L;
            for (size_t r = 0; r < c->n_arity; r++) {
                unsigned ab = (unsigned)c->arity[r]; size_t arity = (size_t)1 << ab;
                gl_t ev[32 * 2];
                r_vec(&q, ev, 2 * arity);
                size_t plen = r_u8(&q); if (plen > 60) { rc = 6; break; }
                r_vec(&q, path, plen * 4);
                size_t coset_index = x_index >> ab, within = x_index & (arity - 1);
                if (!gl2_eq(gl2_make(ev[2 * within], ev[2 * within + 1]), old_eval)) { rc = 5; break; }
                lvl -= ab;
                if (!verify_path(ev, 2 * arity, coset_index, path, plen, fri_caps + r * cap_words, ch_h, lvl)) { rc = 6; break; }
                /* compute_evaluation: interpolate the coset and evaluate at beta */
                gl_t g = gl_root_of_unity(ab);
                size_t rev_within = bitrev32((uint32_t)within, ab);
                gl_t coset_start = gl_mul(subgroup_x, gl_pow(g, arity - rev_within));
                gl2_t pts_y[32]; gl_t pts_x[32];
                for (size_t i = 0; i < arity; i++) {
                    size_t src = bitrev32((uint32_t)i, ab);
                    pts_y[i] = gl2_make(ev[2 * src], ev[2 * src + 1]);
                    pts_x[i] = gl_mul(coset_start, gl_pow(g, i));
                }
                gl2_t beta = fri_betas[r], acc = gl2_from(0);
                for (size_t i = 0; i < arity; i++) {          /* Lagrange form */
                    gl2_t numr = gl2_from(1); gl_t den = 1;
                    for (size_t j = 0; j < arity; j++) if (j != i) {
                        numr = gl2_mul(numr, gl2_sub(beta, gl2_from(pts_x[j])));
                        den = gl_mul(den, gl_sub(pts_x[i], pts_x[j]));
                    }
                    acc = gl2_add(acc, gl2_mul(pts_y[i], gl2_scale(numr, gl_inv(den))));
                }
                old_eval = acc;
                subgroup_x = gl_pow(subgroup_x, arity);
                x_index = coset_index;
            }
            if (rc) break;
            gl2_t fe = gl2_from(0), sxe = gl2_from(subgroup_x);
            for (size_t i = final_len; i-- > 0;) fe = gl2_add(gl2_mul(fe, sxe), final_poly[i]);
            if (!gl2_eq(fe, old_eval)) rc = 7;
        }
        if (q.bad && !rc) rc = 1;
        free(row);
    }
done:
    free(wires_cap); free(zs_cap); free(q_cap); free(o_cs); free(o_w); free(o_zs); free(o_zn); free(o_pp); free(o_q);
    free(fri_caps); free(final_poly); free(pis);
    return rc;
}
