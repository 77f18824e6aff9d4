/*
 * oracle/prove.c — CPU restatement of qp-plonky2 `plonk::prover::prove` + `fri::prover` (stages s4..s12).
 * See oracle/plonk.h. TEST INFRASTRUCTURE ONLY.
 *
 * Stage order and transcript order follow SURVEY.md Appendix A.3-A.5:
 *   wires commit -> betas, gammas -> partial products / Z commit -> alphas -> quotient commit -> zeta ->
 *   openings -> FRI (alpha, per-round cap + beta, final poly, proof of work, query indices).
 */
#include "plonk.h"
#include "challenger.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <omp.h>

#define QPCP_MAGIC 0x0000003150435051ULL
#define MAXC 4

/* ------------------------------------------------------------------ trace */
typedef struct { char name[32]; uint64_t *data; size_t len; } trace_item;
static trace_item g_trace[64];
static int g_ntrace = 0;
static __thread int g_trace_off = 0;     /* set on the threads of orc_prove_many: several proofs at once keep no stage trace */
static void trace_clear(void) { if (g_trace_off) return; for (int i = 0; i < g_ntrace; i++) free(g_trace[i].data); g_ntrace = 0; }
static void trace_put(const char *name, const void *data, size_t words) {
    static double t_last = 0; static int timing = -1;
    if (g_trace_off) return;
    if (timing < 0) timing = getenv("ORC_TIMING") != NULL;
    if (timing) { double t = omp_get_wtime(); fprintf(stderr, "[orc] %-24s +%.3f s\n", name, t_last ? t - t_last : 0.0); t_last = t; }
    if (g_ntrace >= 64) return;
    trace_item *t = &g_trace[g_ntrace++];
    snprintf(t->name, sizeof t->name, "%s", name);
    t->data = (uint64_t *)malloc(words * 8 + 8); t->len = words;
    memcpy(t->data, data, words * 8);
}
size_t orc_trace_len(const char *name) { for (int i = 0; i < g_ntrace; i++) if (!strcmp(g_trace[i].name, name)) return g_trace[i].len; return 0; }
int orc_trace_get(const char *name, uint64_t *out) {
    for (int i = 0; i < g_ntrace; i++) if (!strcmp(g_trace[i].name, name)) { memcpy(out, g_trace[i].data, g_trace[i].len * 8); return 0; }
    return -1;
}

/* ------------------------------------------------------------------ batches */
static void batch_free(orc_batch *b) { free(b->coeffs); free(b->leaves); free(b->digests); free(b->cap); memset(b, 0, sizeof *b); }

/* PolynomialBatch::from_coeffs (no blinding): LDE each column on the coset g<w>, transpose, bit-reverse rows, Merkle */
/* salt for leaf `leaf` of a blinded oracle. The reference draws salts from thread_rng (upstream PolynomialBatch::from_coeffs
 * with blinding), so there is nothing to restate; for byte parity under an injected seed this follows the product's stream
 * (csrc/prover_kernels.hip salt_kernel): ChaCha20 (RFC 8439 block function, 64-bit block counter), key = (seed lo, seed hi,
 * "QPGP", "SALT", 0, 0, 0, 0), nonce = (oracle_index, column), block = leaf >> 1 holding four 64-bit candidates per leaf,
 * the first one below p taken. */
#define ROTL32(x, k) (((x) << (k)) | ((x) >> (32 - (k))))
static void salt_chacha20_block(const uint32_t key[8], uint64_t counter, uint32_t n0, uint32_t n1, uint32_t out[16]) {
    uint32_t x[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key[0], key[1], key[2], key[3], key[4], key[5], key[6], key[7],
                      (uint32_t)counter, (uint32_t)(counter >> 32), n0, n1}, w[16];
    memcpy(w, x, sizeof w);
#define QR(a, b, c, d) w[a] += w[b]; w[d] = ROTL32(w[d] ^ w[a], 16); w[c] += w[d]; w[b] = ROTL32(w[b] ^ w[c], 12); \
                       w[a] += w[b]; w[d] = ROTL32(w[d] ^ w[a], 8);  w[c] += w[d]; w[b] = ROTL32(w[b] ^ w[c], 7);
    for (int r = 0; r < 10; r++) {
        QR(0, 4, 8, 12) QR(1, 5, 9, 13) QR(2, 6, 10, 14) QR(3, 7, 11, 15)
        QR(0, 5, 10, 15) QR(1, 6, 11, 12) QR(2, 7, 8, 13) QR(3, 4, 9, 14)
    }
#undef QR
    for (int i = 0; i < 16; i++) out[i] = w[i] + x[i];
}
gl_t orc_salt_value(uint64_t seed, unsigned oracle_index, unsigned column, uint64_t leaf) {
    const uint32_t key[8] = {(uint32_t)seed, (uint32_t)(seed >> 32), 0x51504750u, 0x53414c54u, 0, 0, 0, 0};
    uint32_t blk[16];
    salt_chacha20_block(key, leaf >> 1, oracle_index, column, blk);
    const unsigned h = (unsigned)(leaf & 1) * 8;
    for (int k = 0; k < 4; k++) {
        const uint64_t cand = ((uint64_t)blk[h + 2 * k + 1] << 32) | blk[h + 2 * k];
        if (cand < GL_P) return cand;
    }
    return (((uint64_t)blk[h + 7] << 32) | blk[h + 6]) - GL_P;
}
/* per proving thread (orc_prove_many runs one proof per thread); read into locals before any parallel loop */
static __thread uint64_t g_seed = 0;
static __thread int g_blind = 0;
static __thread unsigned g_oracle_index = 0;

static void batch_from_coeffs(orc_batch *b, gl_t *coeffs /* owned */, size_t ncols, unsigned log_n, unsigned rate_bits, unsigned cap_height) {
    memset(b, 0, sizeof *b);
    b->ncols = ncols; b->log_n = log_n; b->rate_bits = rate_bits; b->cap_height = cap_height;
    b->n = (size_t)1 << log_n; b->lde_n = b->n << rate_bits;
    b->coeffs = coeffs;
    unsigned L = log_n + rate_bits;
    gl_t *lde = (gl_t *)malloc(sizeof(gl_t) * ncols * b->lde_n);
#pragma omp parallel for schedule(dynamic)
    for (long c = 0; c < (long)ncols; c++) {
        gl_t *col = lde + (size_t)c * b->lde_n;
        memcpy(col, coeffs + (size_t)c * b->n, b->n * sizeof(gl_t));
        memset(col + b->n, 0, (b->lde_n - b->n) * sizeof(gl_t));
        orc_coset_fft(col, L, GL_MULT_GEN);
    }
    const size_t salt = g_blind ? 4 : 0, W = ncols + salt;
    const unsigned oi = g_oracle_index;
    const uint64_t seed = g_seed;
    b->width = W;
    b->leaves = (gl_t *)malloc(sizeof(gl_t) * W * b->lde_n);
#pragma omp parallel for schedule(static)
    for (long j = 0; j < (long)b->lde_n; j++) {
        size_t src = bitrev32((uint32_t)j, L);
        for (size_t c = 0; c < ncols; c++) b->leaves[(size_t)j * W + c] = lde[c * b->lde_n + src];
        for (size_t c = 0; c < salt; c++) b->leaves[(size_t)j * W + ncols + c] = orc_salt_value(seed, oi, (unsigned)c, (uint64_t)j);
    }
    free(lde);
    b->digests = (gl_t *)malloc(sizeof(gl_t) * 4 * 2 * b->lde_n);
    b->cap = (gl_t *)malloc(sizeof(gl_t) * 4 * ((size_t)1 << cap_height));
    orc_merkle_build(b->leaves, b->lde_n, W, cap_height, b->digests, b->cap);
}
/* PolynomialBatch::from_values: ifft then from_coeffs. values is copied. */
static void batch_from_values(orc_batch *b, const gl_t *values, size_t ncols, unsigned log_n, unsigned rate_bits, unsigned cap_height) {
    size_t n = (size_t)1 << log_n;
    gl_t *coeffs = (gl_t *)malloc(sizeof(gl_t) * ncols * n);
    memcpy(coeffs, values, sizeof(gl_t) * ncols * n);
#pragma omp parallel for schedule(dynamic)
    for (long c = 0; c < (long)ncols; c++) orc_ifft(coeffs + (size_t)c * n, log_n);
    batch_from_coeffs(b, coeffs, ncols, log_n, rate_bits, cap_height);
}
/* row of the LDE at natural point index i (plonky2 get_lde_values) */
static const gl_t *batch_lde_row(const orc_batch *b, size_t i) { return b->leaves + (size_t)bitrev32((uint32_t)i, b->log_n + b->rate_bits) * b->width; }

/* ------------------------------------------------------------------ circuit */
orc_circuit *orc_circuit_load(const uint64_t *w, size_t nw) {
    if (nw < 18 || w[0] != QPCP_MAGIC) return NULL;
    orc_circuit *c = (orc_circuit *)calloc(1, sizeof *c);
    size_t p = 1;
    c->degree_bits = w[p++]; c->num_wires = w[p++]; c->num_routed = w[p++]; c->num_constants = w[p++];
    c->num_selectors = w[p++]; c->num_challenges = w[p++]; c->qdf = w[p++]; c->num_pp = w[p++]; c->num_pis = w[p++];
    c->rate_bits = w[p++]; c->cap_height = w[p++]; c->pow_bits = w[p++]; c->num_queries = w[p++]; c->zk = w[p++];
    c->num_gate_constraints = w[p++]; c->n_gates = w[p++]; c->n_arity = w[p++];
    if (c->n_arity > 16 || c->num_challenges > MAXC || c->cap_height > 8 || c->rate_bits > 8 || c->qdf > ((uint64_t)1 << c->rate_bits)) { free(c); return NULL; }
    for (size_t i = 0; i < c->n_arity; i++) c->arity[i] = w[p++];
    c->gates = (orc_gate *)malloc(sizeof(orc_gate) * c->n_gates);
    memcpy(c->gates, w + p, sizeof(orc_gate) * c->n_gates); p += 8 * c->n_gates;
    for (size_t i = 0; i < c->n_gates; i++)   /* scratch bounds of the recursion gates (gates_recursion.inc) */
        if (c->gates[i].type >= OG_REDUCING && (c->gates[i].num_constraints > 256 || (c->gates[i].type == OG_RANDOM_ACCESS && c->gates[i].param0 > 6) ||
                                               (c->gates[i].type == OG_COSET_INTERP && (c->gates[i].param0 > 6 || c->gates[i].param1 < 2)))) { free(c->gates); free(c); return NULL; }
    c->k_is = (gl_t *)malloc(sizeof(gl_t) * c->num_routed);
    memcpy(c->k_is, w + p, sizeof(gl_t) * c->num_routed); p += c->num_routed;
    memcpy(c->digest, w + p, 32); p += 4;
    size_t n = (size_t)1 << c->degree_bits, ncs = c->num_selectors + c->num_constants + c->num_routed;
    { static const uint64_t dflt[10] = {0, 12, 24, 25, 29, 65, 87, 0, 0, 135}; memcpy(c->p2_layout, dflt, sizeof dflt); }
    /* optional trailers are of no concern to the prover: witness hints (stage s1; magic "HINT1", 8 words per entry) and
     * the public-input cells (magic "PUBI1", one word per entry); they are only checked for being well formed */
    {
        size_t q = p + ncs * n;
        int ok = q <= nw;
        while (ok && q != nw) {
            if (q + 2 > nw) { ok = 0; break; }
            const uint64_t magic = w[q], cnt = w[q + 1];
            const uint64_t per = magic == 0x00000031544E4948ULL ? 8 : (magic == 0x0000003149425550ULL || magic == 0x000000314C473250ULL) ? 1 : 0;
            if (!per || cnt > (nw - q - 2) / per) { ok = 0; break; }
            if (magic == 0x000000314C473250ULL) {     /* "P2GL1": the Poseidon2 gate's wire layout, ten words */
                if (cnt != 10) { ok = 0; break; }
                memcpy(c->p2_layout, w + q + 2, sizeof c->p2_layout);
            }
            q += 2 + per * cnt;
        }
        for (size_t i = 0; ok && i < c->n_gates; i++)
            if (c->gates[i].type == OG_POSEIDON2) {    /* the layout must stay inside the wires (scratch bounds of eval_gates_*) */
                const uint64_t *l = c->p2_layout, nwir = c->num_wires;
                const uint64_t f0 = l[7] ? 48 : 36;
                if (l[0] + 12 > nwir || l[1] + 12 > nwir || (l[2] != 0xFFFFFFFFULL && (l[2] >= nwir || l[3] + 4 > nwir)) || l[4] + f0 > nwir || l[5] + 22 > nwir ||
                    l[6] + 48 > nwir || l[7] > 1 || l[8] != 0 || c->gates[i].num_constraints != orc_p2_gate_num_constraints(l)) ok = 0;
            }
        if (!ok) { free(c->gates); free(c->k_is); free(c); return NULL; }
    }
    c->cs_values = (gl_t *)malloc(sizeof(gl_t) * ncs * n);
    memcpy(c->cs_values, w + p, sizeof(gl_t) * ncs * n);
    g_blind = 0;
    batch_from_values(&c->cs, c->cs_values, ncs, (unsigned)c->degree_bits, (unsigned)c->rate_bits, (unsigned)c->cap_height);
    return c;
}
void orc_circuit_free(orc_circuit *c) {
    if (!c) return;
    batch_free(&c->cs); free(c->cs_values); free(c->gates); free(c->k_is); free(c);
}

/* ------------------------------------------------------------------ gate constraints (base field, one point) */
/* compute_filter: prod_{j in group, j != row} (j - s) [* (UNUSED - s) when several selectors] */
static gl_t gate_filter(const orc_circuit *c, size_t gi, gl_t s) {
    const orc_gate *g = &c->gates[gi];
    gl_t f = 1;
    for (uint64_t j = g->group_start; j < g->group_end; j++) if (j != gi) f = gl_mul(f, gl_sub(j, s));
    if (c->num_selectors > 1) f = gl_mul(f, gl_sub(0xFFFFFFFFULL, s));
    return f;
}
/*
 * PoseidonGate (plonky2::gates::poseidon): wires 0..11 input, 12..23 output, 24 swap, 25..28 delta, 29..64 S-box inputs
 * of full rounds 1..3, 65..86 S-box inputs of the 22 partial rounds, 87..134 S-box inputs of the last 4 full rounds;
 * 123 constraints of degree 7. The partial rounds run in upstream's fast basis (partial_first_constant_layer,
 * mds_partial_layer_init, mds_partial_layer_fast); its FAST_PARTIAL_* tables are re-derived below from the MDS matrix and
 * the round constants (published HADES optimisation) and pinned by tests/golden/poseidon_fast_partial.json.
 */
void orc_poseidon_round_constants(gl_t *out);
static gl_t PRC[360]; static int prc_ready = 0;
static const gl_t PMDS[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
static void pg_mds(gl_t s[12]) {
    gl_t o[12];
    for (int r = 0; r < 12; r++) {
        u128 acc = 0;
        for (int i = 0; i < 12; i++) acc += (u128)s[(i + r) % 12] * PMDS[i];
        if (r == 0) acc += (u128)s[0] * 8;
        o[r] = gl_reduce128(acc);
    }
    memcpy(s, o, sizeof o);
}
/* ---- FAST_PARTIAL tables: FIRST[12], RCP[22], VS[22][11], WH[22][11], INIT[11][11] (row c gives element 1+c) ---- */
static gl_t FP_FIRST[12], FP_RC[22], FP_VS[22][11], FP_WH[22][11], FP_INIT[11][11];
static void mat_inv(gl_t *a, gl_t *inv, int n) {          /* Gauss-Jordan, a destroyed */
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) inv[i * n + j] = i == j;
    for (int c = 0; c < n; c++) {
        int piv = c; while (a[piv * n + c] == 0) piv++;
        for (int j = 0; j < n; j++) { gl_t t = a[c * n + j]; a[c * n + j] = a[piv * n + j]; a[piv * n + j] = t; t = inv[c * n + j]; inv[c * n + j] = inv[piv * n + j]; inv[piv * n + j] = t; }
        gl_t f = gl_inv(a[c * n + c]);
        for (int j = 0; j < n; j++) { a[c * n + j] = gl_mul(a[c * n + j], f); inv[c * n + j] = gl_mul(inv[c * n + j], f); }
        for (int r = 0; r < n; r++) if (r != c && a[r * n + c]) {
            gl_t g = a[r * n + c];
            for (int j = 0; j < n; j++) { a[r * n + j] = gl_sub(a[r * n + j], gl_mul(g, a[c * n + j])); inv[r * n + j] = gl_sub(inv[r * n + j], gl_mul(g, inv[c * n + j])); }
        }
    }
}
static void derive_fast_partial(void) {
    gl_t M[144], Mt[144], Minv[144], C[22][12];
    for (int r = 0; r < 12; r++) for (int c = 0; c < 12; c++) M[r * 12 + c] = PMDS[(c - r + 12) % 12] + (r == 0 && c == 0 ? 8 : 0);
    memcpy(Mt, M, sizeof M); mat_inv(Mt, Minv, 12);
    for (int k = 0; k < 22; k++) memcpy(C[k], PRC + (4 + k) * 12, sizeof C[k]);
    for (int k = 20; k >= 0; k--) {       /* constants of round k+1 move behind the S-box of round k */
        gl_t w[12];
        for (int i = 0; i < 12; i++) { gl_t acc = 0; for (int j = 0; j < 12; j++) acc = gl_add(acc, gl_mul(Minv[i * 12 + j], C[k + 1][j])); w[i] = acc; }
        for (int i = 1; i < 12; i++) C[k][i] = gl_add(C[k][i], w[i]);
        FP_RC[k] = w[0];
    }
    FP_RC[21] = 0;
    memcpy(FP_FIRST, C[0], sizeof FP_FIRST);
    gl_t Mmul[144]; memcpy(Mmul, M, sizeof M);
    gl_t B[121], Bt[121], Binv[121];
    for (int i = 21; i >= 0; i--) {       /* Mmul = M'' * diag(1, B) */
        gl_t row[11];
        for (int r = 0; r < 11; r++) { FP_VS[i][r] = Mmul[(r + 1) * 12]; row[r] = Mmul[r + 1]; for (int c = 0; c < 11; c++) B[r * 11 + c] = Mmul[(r + 1) * 12 + c + 1]; }
        for (int r = 0; r < 11; r++) for (int c = 0; c < 11; c++) Bt[r * 11 + c] = B[c * 11 + r];
        mat_inv(Bt, Binv, 11);
        for (int r = 0; r < 11; r++) { gl_t acc = 0; for (int c = 0; c < 11; c++) acc = gl_add(acc, gl_mul(Binv[r * 11 + c], row[c])); FP_WH[i][r] = acc; }
        /* previous round's matrix: diag(1, B) * M */
        gl_t nm[144];
        for (int c = 0; c < 12; c++) nm[c] = M[c];
        for (int r = 0; r < 11; r++) for (int c = 0; c < 12; c++) { gl_t acc = 0; for (int k = 0; k < 11; k++) acc = gl_add(acc, gl_mul(B[r * 11 + k], M[(k + 1) * 12 + c])); nm[(r + 1) * 12 + c] = acc; }
        memcpy(Mmul, nm, sizeof nm);
    }
    for (int c = 0; c < 11; c++) for (int r = 0; r < 11; r++) FP_INIT[c][r] = B[c * 11 + r];
}
static void pg_init(void) {
    if (prc_ready) return;
    orc_poseidon_round_constants(PRC);
    derive_fast_partial();
    prc_ready = 1;
}
/* exports for the tests: flattened like the product's table */
size_t orc_poseidon_fast_partial(gl_t *out) {
    pg_init();
    size_t k = 0;
    for (int i = 0; i < 12; i++) out[k++] = FP_FIRST[i];
    for (int i = 0; i < 22; i++) out[k++] = FP_RC[i];
    for (int i = 0; i < 22; i++) for (int j = 0; j < 11; j++) out[k++] = FP_VS[i][j];
    for (int i = 0; i < 22; i++) for (int j = 0; j < 11; j++) out[k++] = FP_WH[i][j];
    for (int i = 0; i < 11; i++) for (int j = 0; j < 11; j++) out[k++] = FP_INIT[i][j];
    return k;
}
static inline gl_t pg_sbox(gl_t x) { gl_t x2 = gl_sqr(x), x4 = gl_sqr(x2); return gl_mul(gl_mul(x, x2), x4); }
/* The permutation the way CPU implementations of plonky2 organise it (poseidon.rs: full rounds, then
 * partial_first_constant_layer + mds_partial_layer_init, 22 x [S-box on lane 0, add the scalar constant, mds_partial_layer_fast],
 * then full rounds): the same map as the textbook schedule in poseidon.c (tests/test_oracle_poseidon.py holds them against each
 * other and against the golden vectors), about four times cheaper, which is what makes bench.py's cpu_baseline quotable. Dot
 * products accumulate the 128-bit products in two 128-bit sums (low and high halves) and reduce once. */
typedef struct { u128 lo, hi; } acc256;
static inline void acc_mul(acc256 *a, gl_t x, gl_t y) { const u128 p = (u128)x * y; a->lo += (uint64_t)p; a->hi += (uint64_t)(p >> 64); }
static inline gl_t acc_reduce(const acc256 *a) { return gl_add(gl_reduce128(a->lo), gl_mul(gl_reduce128(a->hi), GL_EPS)); }   /* lo + hi * 2^64 */
static inline void fast_mds(gl_t s[12]) {
    uint64_t lo[24], hi[24];
    for (int i = 0; i < 12; i++) { lo[i] = lo[i + 12] = (uint32_t)s[i]; hi[i] = hi[i + 12] = s[i] >> 32; }
    for (int r = 0; r < 12; r++) {
        uint64_t al = 0, ah = 0;
        for (int i = 0; i < 12; i++) { al += lo[i + r] * PMDS[i]; ah += hi[i + r] * PMDS[i]; }
        if (r == 0) { al += lo[0] * 8; ah += hi[0] * 8; }
        s[r] = gl_reduce128((u128)al + ((u128)ah << 32));
    }
}
void orc_poseidon_permute_fast(gl_t s[12]) {
    pg_init();
    int rc = 0;
    for (int r = 0; r < 4; r++, rc++) {
        for (int i = 0; i < 12; i++) s[i] = pg_sbox(gl_add(s[i], PRC[rc * 12 + i]));
        fast_mds(s);
    }
    for (int i = 0; i < 12; i++) s[i] = gl_add(s[i], FP_FIRST[i]);
    {
        gl_t t[11];
        for (int c = 0; c < 11; c++) { acc256 a = {0, 0}; for (int r = 0; r < 11; r++) acc_mul(&a, FP_INIT[c][r], s[1 + r]); t[c] = acc_reduce(&a); }
        memcpy(s + 1, t, sizeof t);
    }
    for (int r = 0; r < 22; r++) {
        const gl_t s0 = gl_add(pg_sbox(s[0]), FP_RC[r]);
        acc256 a = {0, 0};
        acc_mul(&a, s0, 25);
        for (int i = 0; i < 11; i++) acc_mul(&a, FP_WH[r][i], s[1 + i]);
        for (int i = 0; i < 11; i++) s[1 + i] = gl_add(s[1 + i], gl_mul(s0, FP_VS[r][i]));
        s[0] = acc_reduce(&a);
    }
    rc += 22;
    for (int r = 0; r < 4; r++, rc++) {
        for (int i = 0; i < 12; i++) s[i] = pg_sbox(gl_add(s[i], PRC[rc * 12 + i]));
        fast_mds(s);
    }
}
/* emits the 123 constraints into out[] */
static void poseidon_gate_base(const gl_t *w, gl_t *out) {
    pg_init();
    size_t k = 0;
    gl_t swap = w[24], st[12];
    out[k++] = gl_mul(swap, gl_sub(swap, 1));
    for (int i = 0; i < 4; i++) out[k++] = gl_sub(gl_mul(swap, gl_sub(w[i + 4], w[i])), w[25 + i]);
    for (int i = 0; i < 4; i++) { st[i] = gl_add(w[i], w[25 + i]); st[i + 4] = gl_sub(w[i + 4], w[25 + i]); }
    for (int i = 8; i < 12; i++) st[i] = w[i];
    int rc = 0;
    for (int r = 0; r < 4; r++, rc++) {
        for (int i = 0; i < 12; i++) st[i] = gl_add(st[i], PRC[rc * 12 + i]);
        if (r) for (int i = 0; i < 12; i++) { gl_t in = w[29 + 12 * (r - 1) + i]; out[k++] = gl_sub(st[i], in); st[i] = in; }
        for (int i = 0; i < 12; i++) st[i] = pg_sbox(st[i]);
        pg_mds(st);
    }
    for (int i = 0; i < 12; i++) st[i] = gl_add(st[i], FP_FIRST[i]);      /* partial_first_constant_layer */
    { gl_t t[11]; for (int c = 0; c < 11; c++) { gl_t acc = 0; for (int r = 0; r < 11; r++) acc = gl_add(acc, gl_mul(FP_INIT[c][r], st[1 + r])); t[c] = acc; }
      memcpy(st + 1, t, sizeof t); }                                       /* mds_partial_layer_init */
    for (int r = 0; r < 22; r++) {
        gl_t in = w[65 + r]; out[k++] = gl_sub(st[0], in);
        gl_t s0 = gl_add(pg_sbox(in), FP_RC[r]);
        gl_t d = gl_mul(s0, 25);
        for (int i = 0; i < 11; i++) d = gl_add(d, gl_mul(FP_WH[r][i], st[1 + i]));
        for (int i = 0; i < 11; i++) st[1 + i] = gl_add(st[1 + i], gl_mul(s0, FP_VS[r][i]));
        st[0] = d;                                                         /* mds_partial_layer_fast */
    }
    rc += 22;
    for (int r = 0; r < 4; r++, rc++) {
        for (int i = 0; i < 12; i++) st[i] = gl_add(st[i], PRC[rc * 12 + i]);
        for (int i = 0; i < 12; i++) { gl_t in = w[87 + 12 * r + i]; out[k++] = gl_sub(st[i], in); st[i] = in; }
        for (int i = 0; i < 12; i++) st[i] = pg_sbox(st[i]);
        pg_mds(st);
    }
    for (int i = 0; i < 12; i++) out[k++] = gl_sub(st[i], w[12 + i]);
}
static void pg_mds_ext(gl2_t s[12]) {
    gl_t a[12], b[12];
    for (int i = 0; i < 12; i++) { a[i] = s[i].c[0]; b[i] = s[i].c[1]; }
    pg_mds(a); pg_mds(b);
    for (int i = 0; i < 12; i++) s[i] = gl2_make(a[i], b[i]);
}
static inline gl2_t pg_sbox_ext(gl2_t x) { gl2_t x2 = gl2_mul(x, x), x4 = gl2_mul(x2, x2); return gl2_mul(gl2_mul(x, x2), x4); }
static void poseidon_gate_ext(const gl2_t *w, gl2_t *out) {
    pg_init();
    size_t k = 0;
    gl2_t swap = w[24], st[12];
    out[k++] = gl2_mul(swap, gl2_sub(swap, gl2_from(1)));
    for (int i = 0; i < 4; i++) out[k++] = gl2_sub(gl2_mul(swap, gl2_sub(w[i + 4], w[i])), w[25 + i]);
    for (int i = 0; i < 4; i++) { st[i] = gl2_add(w[i], w[25 + i]); st[i + 4] = gl2_sub(w[i + 4], w[25 + i]); }
    for (int i = 8; i < 12; i++) st[i] = w[i];
    int rc = 0;
    for (int r = 0; r < 4; r++, rc++) {
        for (int i = 0; i < 12; i++) st[i] = gl2_add(st[i], gl2_from(PRC[rc * 12 + i]));
        if (r) for (int i = 0; i < 12; i++) { gl2_t in = w[29 + 12 * (r - 1) + i]; out[k++] = gl2_sub(st[i], in); st[i] = in; }
        for (int i = 0; i < 12; i++) st[i] = pg_sbox_ext(st[i]);
        pg_mds_ext(st);
    }
    for (int i = 0; i < 12; i++) st[i] = gl2_add(st[i], gl2_from(FP_FIRST[i]));
    { gl2_t t[11]; for (int c = 0; c < 11; c++) { gl2_t acc = gl2_from(0); for (int r = 0; r < 11; r++) acc = gl2_add(acc, gl2_scale(st[1 + r], FP_INIT[c][r])); t[c] = acc; }
      memcpy(st + 1, t, sizeof t); }
    for (int r = 0; r < 22; r++) {
        gl2_t in = w[65 + r]; out[k++] = gl2_sub(st[0], in);
        gl2_t s0 = gl2_add(pg_sbox_ext(in), gl2_from(FP_RC[r]));
        gl2_t d = gl2_scale(s0, 25);
        for (int i = 0; i < 11; i++) d = gl2_add(d, gl2_scale(st[1 + i], FP_WH[r][i]));
        for (int i = 0; i < 11; i++) st[1 + i] = gl2_add(st[1 + i], gl2_scale(s0, FP_VS[r][i]));
        st[0] = d;
    }
    rc += 22;
    for (int r = 0; r < 4; r++, rc++) {
        for (int i = 0; i < 12; i++) st[i] = gl2_add(st[i], gl2_from(PRC[rc * 12 + i]));
        for (int i = 0; i < 12; i++) { gl2_t in = w[87 + 12 * r + i]; out[k++] = gl2_sub(st[i], in); st[i] = in; }
        for (int i = 0; i < 12; i++) st[i] = pg_sbox_ext(st[i]);
        pg_mds_ext(st);
    }
    for (int i = 0; i < 12; i++) out[k++] = gl2_sub(st[i], w[12 + i]);
}

/* the recursion gate set, once per scalar type */
/* CosetInterpolationGate tables: the subgroup of order 2^bits and barycentric_weights = 1 / prod_{j != i} (x_i - x_j),
   by the definition (the GPU side uses the closed form x_i / n; the parity tests therefore check that identity too) */
static void coset_interp_tables(unsigned bits, gl_t *dom, gl_t *wt) {
    static gl_t c_dom[7][64], c_wt[7][64];
    static int ready[7];
    int ok;
    #pragma omp atomic read
    ok = ready[bits];
    if (!ok) {
        #pragma omp critical(coset_interp_tables)
        if (!ready[bits]) {
            const size_t np = (size_t)1 << bits;
            gl_t om = 7277203076849721926ULL, x = 1;
            for (unsigned i = bits; i < 32; i++) om = gl_mul(om, om);
            for (size_t i = 0; i < np; i++) { c_dom[bits][i] = x; x = gl_mul(x, om); }
            for (size_t i = 0; i < np; i++) {
                gl_t pr = 1;
                for (size_t j = 0; j < np; j++) if (j != i) pr = gl_mul(pr, gl_sub(c_dom[bits][i], c_dom[bits][j]));
                c_wt[bits][i] = gl_inv(pr);
            }
            #pragma omp atomic write
            ready[bits] = 1;
        }
    }
    memcpy(dom, c_dom[bits], sizeof(gl_t) << bits); memcpy(wt, c_wt[bits], sizeof(gl_t) << bits);
}
#define GN(name) name##_base
#define T gl_t
#define T_ADD gl_add
#define T_SUB gl_sub
#define T_MUL gl_mul
#define T_FROM(x) ((gl_t)(x))
#include "gates_recursion.inc"
#undef GN
#undef T
#undef T_ADD
#undef T_SUB
#undef T_MUL
#undef T_FROM
#define GN(name) name##_ext
#define T gl2_t
#define T_ADD gl2_add
#define T_SUB gl2_sub
#define T_MUL gl2_mul
#define T_FROM(x) gl2_from((gl_t)(x))
#include "gates_recursion.inc"
#undef GN
#undef T
#undef T_ADD
#undef T_SUB
#undef T_MUL
#undef T_FROM
#define MAX_GATE_CONSTRAINTS 256

/* adds filter * constraint_k into acc[k]. consts = local constants after the selector prefix. */
static void eval_gates_base(const orc_circuit *c, const gl_t *cs_row, const gl_t *wires, const gl_t pih[4], gl_t *acc) {
    const gl_t *consts = cs_row + c->num_selectors;
    for (size_t gi = 0; gi < c->n_gates; gi++) {
        const orc_gate *g = &c->gates[gi];
        if (g->num_constraints == 0) continue;
        gl_t f = gate_filter(c, gi, cs_row[g->selector_index]);
        switch (g->type) {
        case OG_CONSTANT:
            for (uint64_t i = 0; i < g->param0; i++) acc[i] = gl_add(acc[i], gl_mul(f, gl_sub(consts[i], wires[i])));
            break;
        case OG_PUBLIC_INPUT:
            for (int i = 0; i < 4; i++) acc[i] = gl_add(acc[i], gl_mul(f, gl_sub(wires[i], pih[i])));
            break;
        case OG_ARITHMETIC:
            for (uint64_t i = 0; i < g->param0; i++) {
                gl_t m0 = wires[4 * i], m1 = wires[4 * i + 1], ad = wires[4 * i + 2], out = wires[4 * i + 3];
                gl_t computed = gl_add(gl_mul(gl_mul(m0, m1), consts[0]), gl_mul(ad, consts[1]));
                acc[i] = gl_add(acc[i], gl_mul(f, gl_sub(out, computed)));
            }
            break;
        case OG_POSEIDON: {
            gl_t cst[123];
            poseidon_gate_base(wires, cst);
            for (int i = 0; i < 123; i++) acc[i] = gl_add(acc[i], gl_mul(f, cst[i]));
            break;
        }
        case OG_POSEIDON2: {
            gl_t cst[128];
            const size_t nc = orc_p2_gate_base(c->p2_layout, wires, cst);
            for (size_t i = 0; i < nc; i++) acc[i] = gl_add(acc[i], gl_mul(f, cst[i]));
            break;
        }
        case OG_ARITHMETIC_EXT:   /* ArithmeticExtensionGate<2>: out - (c0 m0 m1 + c1 addend) over F[x]/(x^2-7), 8 wires per op */
            for (uint64_t i = 0; i < g->param0; i++) {
                const gl_t *w = wires + 8 * i;
                gl2_t m0 = gl2_make(w[0], w[1]), m1 = gl2_make(w[2], w[3]), ad = gl2_make(w[4], w[5]), out = gl2_make(w[6], w[7]);
                gl2_t d = gl2_sub(out, gl2_add(gl2_scale(gl2_mul(m0, m1), consts[0]), gl2_scale(ad, consts[1])));
                acc[2 * i] = gl_add(acc[2 * i], gl_mul(f, d.c[0])); acc[2 * i + 1] = gl_add(acc[2 * i + 1], gl_mul(f, d.c[1]));
            }
            break;
        case OG_MUL_EXT:          /* MulExtensionGate<2>: out - c0 m0 m1, 6 wires per op */
            for (uint64_t i = 0; i < g->param0; i++) {
                const gl_t *w = wires + 6 * i;
                gl2_t d = gl2_sub(gl2_make(w[4], w[5]), gl2_scale(gl2_mul(gl2_make(w[0], w[1]), gl2_make(w[2], w[3])), consts[0]));
                acc[2 * i] = gl_add(acc[2 * i], gl_mul(f, d.c[0])); acc[2 * i + 1] = gl_add(acc[2 * i + 1], gl_mul(f, d.c[1]));
            }
            break;
        case OG_BASE_SUM: {   /* BaseSumGate<2>: wire 0 = sum, wires 1..num_limbs = bits */
            gl_t s2 = 0;
            for (uint64_t i = g->param0; i-- > 0;) s2 = gl_add(gl_add(s2, s2), wires[1 + i]);
            acc[0] = gl_add(acc[0], gl_mul(f, gl_sub(s2, wires[0])));
            for (uint64_t i = 0; i < g->param0; i++) acc[1 + i] = gl_add(acc[1 + i], gl_mul(f, gl_mul(wires[1 + i], gl_sub(wires[1 + i], 1))));
            break;
        }
        case OG_REDUCING: case OG_REDUCING_EXT: case OG_RANDOM_ACCESS: case OG_EXPONENTIATION: case OG_POSEIDON_MDS: case OG_COSET_INTERP: {
            gl_t cst[MAX_GATE_CONSTRAINTS];
            recursion_gate_base(g, consts, wires, cst);
            for (uint64_t i = 0; i < g->num_constraints; i++) acc[i] = gl_add(acc[i], gl_mul(f, cst[i]));
            break;
        }
        default: break;
        }
    }
}
/* same over the extension (verifier side, at zeta) */
static gl2_t gate_filter_ext(const orc_circuit *c, size_t gi, gl2_t s) {
    const orc_gate *g = &c->gates[gi];
    gl2_t f = gl2_from(1);
    for (uint64_t j = g->group_start; j < g->group_end; j++) if (j != gi) f = gl2_mul(f, gl2_sub(gl2_from(j), s));
    if (c->num_selectors > 1) f = gl2_mul(f, gl2_sub(gl2_from(0xFFFFFFFFULL), s));
    return f;
}
void orc_eval_gates_ext(const orc_circuit *c, const gl2_t *cs_row, const gl2_t *wires, const gl_t pih[4], gl2_t *acc) {
    const gl2_t *consts = cs_row + c->num_selectors;
    for (size_t gi = 0; gi < c->n_gates; gi++) {
        const orc_gate *g = &c->gates[gi];
        if (g->num_constraints == 0) continue;
        gl2_t f = gate_filter_ext(c, gi, cs_row[g->selector_index]);
        switch (g->type) {
        case OG_CONSTANT:
            for (uint64_t i = 0; i < g->param0; i++) acc[i] = gl2_add(acc[i], gl2_mul(f, gl2_sub(consts[i], wires[i])));
            break;
        case OG_PUBLIC_INPUT:
            for (int i = 0; i < 4; i++) acc[i] = gl2_add(acc[i], gl2_mul(f, gl2_sub(wires[i], gl2_from(pih[i]))));
            break;
        case OG_ARITHMETIC:
            for (uint64_t i = 0; i < g->param0; i++) {
                gl2_t computed = gl2_add(gl2_mul(gl2_mul(wires[4 * i], wires[4 * i + 1]), consts[0]), gl2_mul(wires[4 * i + 2], consts[1]));
                acc[i] = gl2_add(acc[i], gl2_mul(f, gl2_sub(wires[4 * i + 3], computed)));
            }
            break;
        case OG_POSEIDON: {
            gl2_t cst[123];
            poseidon_gate_ext(wires, cst);
            for (int i = 0; i < 123; i++) acc[i] = gl2_add(acc[i], gl2_mul(f, cst[i]));
            break;
        }
        case OG_POSEIDON2: {
            gl2_t cst[128];
            const size_t nc = orc_p2_gate_ext(c->p2_layout, wires, cst);
            for (size_t i = 0; i < nc; i++) acc[i] = gl2_add(acc[i], gl2_mul(f, cst[i]));
            break;
        }
        case OG_ARITHMETIC_EXT:   /* wires are extension values: the algebra F2[X]/(X^2-7) with coefficients in F2 */
            for (uint64_t i = 0; i < g->param0; i++) {
                const gl2_t *w = wires + 8 * i;
                gl2_t p0 = gl2_add(gl2_mul(w[0], w[2]), gl2_scale(gl2_mul(w[1], w[3]), 7));
                gl2_t p1 = gl2_add(gl2_mul(w[0], w[3]), gl2_mul(w[1], w[2]));
                gl2_t d0 = gl2_sub(w[6], gl2_add(gl2_mul(p0, consts[0]), gl2_mul(w[4], consts[1])));
                gl2_t d1 = gl2_sub(w[7], gl2_add(gl2_mul(p1, consts[0]), gl2_mul(w[5], consts[1])));
                acc[2 * i] = gl2_add(acc[2 * i], gl2_mul(f, d0)); acc[2 * i + 1] = gl2_add(acc[2 * i + 1], gl2_mul(f, d1));
            }
            break;
        case OG_MUL_EXT:
            for (uint64_t i = 0; i < g->param0; i++) {
                const gl2_t *w = wires + 6 * i;
                gl2_t p0 = gl2_add(gl2_mul(w[0], w[2]), gl2_scale(gl2_mul(w[1], w[3]), 7));
                gl2_t p1 = gl2_add(gl2_mul(w[0], w[3]), gl2_mul(w[1], w[2]));
                gl2_t d0 = gl2_sub(w[4], gl2_mul(p0, consts[0])), d1 = gl2_sub(w[5], gl2_mul(p1, consts[0]));
                acc[2 * i] = gl2_add(acc[2 * i], gl2_mul(f, d0)); acc[2 * i + 1] = gl2_add(acc[2 * i + 1], gl2_mul(f, d1));
            }
            break;
        case OG_BASE_SUM: {
            gl2_t s2 = gl2_from(0);
            for (uint64_t i = g->param0; i-- > 0;) s2 = gl2_add(gl2_add(s2, s2), wires[1 + i]);
            acc[0] = gl2_add(acc[0], gl2_mul(f, gl2_sub(s2, wires[0])));
            for (uint64_t i = 0; i < g->param0; i++) acc[1 + i] = gl2_add(acc[1 + i], gl2_mul(f, gl2_mul(wires[1 + i], gl2_sub(wires[1 + i], gl2_from(1)))));
            break;
        }
        case OG_REDUCING: case OG_REDUCING_EXT: case OG_RANDOM_ACCESS: case OG_EXPONENTIATION: case OG_POSEIDON_MDS: case OG_COSET_INTERP: {
            gl2_t cst[MAX_GATE_CONSTRAINTS];
            recursion_gate_ext(g, consts, wires, cst);
            for (uint64_t i = 0; i < g->num_constraints; i++) acc[i] = gl2_add(acc[i], gl2_mul(f, cst[i]));
            break;
        }
        default: break;
        }
    }
}

/* ------------------------------------------------------------------ helpers */
static void batch_inverse(gl_t *x, size_t n, gl_t *scratch) {
    gl_t acc = 1;
    for (size_t i = 0; i < n; i++) { scratch[i] = acc; acc = gl_mul(acc, x[i]); }
    gl_t inv = gl_inv(acc);
    for (size_t i = n; i-- > 0;) { gl_t t = gl_mul(inv, scratch[i]); inv = gl_mul(inv, x[i]); x[i] = t; }
}
static gl2_t challenger_get_ext(orc_challenger *ch) { gl_t a = orc_challenger_get(ch), b = orc_challenger_get(ch); return gl2_make(a, b); }
static gl2_t eval_poly_ext(const gl_t *coeffs, size_t n, gl2_t z) {
    gl2_t acc = gl2_from(0);
    for (size_t i = n; i-- > 0;) acc = gl2_add(gl2_mul(acc, z), gl2_from(coeffs[i]));
    return acc;
}

typedef struct { uint8_t *p; size_t cap, len; int overflow; } wbuf;
static void w_u64(wbuf *b, uint64_t v) { if (b->len + 8 > b->cap) { b->overflow = 1; b->len += 8; return; } for (int k = 0; k < 8; k++) b->p[b->len + k] = (uint8_t)(v >> (8 * k)); b->len += 8; }
static void w_u8(wbuf *b, uint8_t v) { if (b->len + 1 > b->cap) { b->overflow = 1; b->len += 1; return; } b->p[b->len++] = v; }
static void w_vec(wbuf *b, const gl_t *v, size_t n) { for (size_t i = 0; i < n; i++) w_u64(b, v[i]); }
static void w_ext(wbuf *b, gl2_t v) { w_u64(b, v.c[0]); w_u64(b, v.c[1]); }
static void w_path(wbuf *b, const gl_t *digests, size_t n_leaves, unsigned cap_h, size_t idx) {
    gl_t path[64 * 4];
    size_t len = orc_merkle_path(digests, n_leaves, cap_h, idx, path);
    w_u8(b, (uint8_t)len);
    w_vec(b, path, len * 4);
}

size_t orc_proof_size(const orc_circuit *c) {
    size_t n_cs = c->num_selectors + c->num_constants + c->num_routed, nch = c->num_challenges;
    size_t cap = ((size_t)1 << c->cap_height) * 4 * 8;
    size_t openings = (n_cs + c->num_wires + nch * 2 + nch * c->num_pp + nch * c->qdf) * 16;
    size_t L = c->degree_bits + c->rate_bits, sz = 3 * cap + openings;
    size_t salt = c->zk ? 4 : 0;
    size_t widths[4] = {n_cs, c->num_wires + salt, nch * (1 + c->num_pp) + salt, nch * c->qdf + salt};
    size_t q = 0;
    for (int o = 0; o < 4; o++) q += widths[o] * 8 + 1 + (L - c->cap_height) * 32;
    size_t lvl = L, fin = c->degree_bits;
    for (size_t r = 0; r < c->n_arity; r++) {
        sz += cap; lvl -= c->arity[r]; fin -= c->arity[r];
        q += ((size_t)1 << c->arity[r]) * 16 + 1 + (lvl - c->cap_height) * 32;
    }
    sz += c->num_queries * q + ((size_t)1 << fin) * 16 + 8 + c->num_pis * 8;
    return sz;
}

/* ------------------------------------------------------------------ prove */
int orc_prove(const orc_circuit *c, const gl_t *wires, const gl_t *public_inputs, uint8_t *out, size_t cap, size_t *len) {
    return orc_prove_seeded(c, wires, public_inputs, 0, out, cap, len);
}
/* orc_prove on a thread that proves beside others (orc_commit_prove_many): keeps no stage trace */
int orc_prove_many_entry(const orc_circuit *c, const gl_t *wires, const gl_t *public_inputs, uint8_t *out, size_t cap, size_t *len) {
    g_trace_off = 1;
    const int rc = orc_prove_seeded(c, wires, public_inputs, 0, out, cap, len);
    g_trace_off = 0;
    return rc;
}
int orc_prove_seeded(const orc_circuit *c, const gl_t *wires, const gl_t *public_inputs, uint64_t seed, uint8_t *out, size_t cap, size_t *len) {
    trace_clear();
    g_seed = seed;
    const unsigned d = (unsigned)c->degree_bits, rb = (unsigned)c->rate_bits, ch_h = (unsigned)c->cap_height, L = d + rb;
    const size_t n = (size_t)1 << d, lde_n = n << rb, R = c->num_routed, NW = c->num_wires, nch = c->num_challenges;
    const size_t npp = c->num_pp, nchunks = npp + 1, chunk = c->qdf, ncs = c->num_selectors + c->num_constants + R;
    const size_t sig0 = c->num_selectors + c->num_constants, cap_words = ((size_t)1 << ch_h) * 4;

    gl_t pih[4];
    orc_hash_no_pad(public_inputs, c->num_pis, pih);
    trace_put("pi_hash", pih, 4);

    /* s2/s3: wires commitment */
    orc_batch wb;
    g_blind = c->zk ? 1 : 0; g_oracle_index = 1;
    batch_from_values(&wb, wires, NW, d, rb, ch_h);
    trace_put("wires_cap", wb.cap, cap_words);

    orc_challenger ch;
    orc_challenger_init(&ch);
    orc_challenger_observe(&ch, c->digest, 4);
    orc_challenger_observe(&ch, pih, 4);
    orc_challenger_observe(&ch, wb.cap, cap_words);
    gl_t betas[MAXC], gammas[MAXC], alphas[MAXC];
    orc_challenger_get_n(&ch, betas, nch);
    orc_challenger_get_n(&ch, gammas, nch);
    trace_put("betas", betas, nch); trace_put("gammas", gammas, nch);

    /* s5: partial products and Z. zs_pp columns: [Z_0..Z_{nch-1}, pp_0_*, pp_1_*, ...] */
    const size_t nzp = nch * (1 + npp);
    gl_t *zs_pp = (gl_t *)malloc(sizeof(gl_t) * nzp * n);
    gl_t *omega = (gl_t *)malloc(sizeof(gl_t) * n);
    { gl_t w = gl_root_of_unity(d), a = 1; for (size_t i = 0; i < n; i++) { omega[i] = a; a = gl_mul(a, w); } }
    for (size_t k = 0; k < nch; k++) {
        gl_t *qcp = (gl_t *)malloc(sizeof(gl_t) * n * nchunks);  /* quotient chunk products per row */
#pragma omp parallel
        {
            gl_t *den = (gl_t *)malloc(sizeof(gl_t) * R), *num = (gl_t *)malloc(sizeof(gl_t) * R), *scr = (gl_t *)malloc(sizeof(gl_t) * R);
#pragma omp for schedule(static)
            for (long i = 0; i < (long)n; i++) {
                gl_t x = omega[i];
                for (size_t j = 0; j < R; j++) {
                    gl_t wv = wires[j * n + i];
                    num[j] = gl_add(gl_add(wv, gl_mul(betas[k], gl_mul(c->k_is[j], x))), gammas[k]);
                    den[j] = gl_add(gl_add(wv, gl_mul(betas[k], c->cs_values[(sig0 + j) * n + i])), gammas[k]);
                }
                batch_inverse(den, R, scr);
                for (size_t cc = 0; cc < nchunks; cc++) {
                    gl_t p = 1;
                    for (size_t j = cc * chunk; j < (cc + 1) * chunk && j < R; j++) p = gl_mul(p, gl_mul(num[j], den[j]));
                    qcp[(size_t)i * nchunks + cc] = p;
                }
            }
            free(den); free(num); free(scr);
        }
        gl_t z = 1;
        for (size_t i = 0; i < n; i++) {
            zs_pp[k * n + i] = z;   /* Z(x_i) */
            gl_t acc = z;
            for (size_t cc = 0; cc < nchunks; cc++) {
                acc = gl_mul(acc, qcp[i * nchunks + cc]);
                if (cc < npp) zs_pp[(nch + k * npp + cc) * n + i] = acc;
            }
            z = acc;                /* Z(g x_i) */
        }
        free(qcp);
    }
    trace_put("zs_pp_values", zs_pp, nzp * n);
    orc_batch zb;
    g_oracle_index = 2;
    batch_from_values(&zb, zs_pp, nzp, d, rb, ch_h);
    free(zs_pp);
    trace_put("zs_pp_cap", zb.cap, cap_words);
    orc_challenger_observe(&ch, zb.cap, cap_words);
    orc_challenger_get_n(&ch, alphas, nch);
    trace_put("alphas", alphas, nch);

    /* s6: quotient polynomials on the coset g<w_{n qdf}>: every `step`-th point of the LDE (compute_quotient_polys) */
    unsigned qbits = 0; while (((uint64_t)1 << qbits) < c->qdf) qbits++;
    const size_t q_n = n << qbits, step = (size_t)1 << (rb - qbits);
    gl_t *quot = (gl_t *)malloc(sizeof(gl_t) * nch * lde_n);
    {
        gl_t zh_inv[256];
        gl_t gn = gl_pow(GL_MULT_GEN, n), wr = gl_root_of_unity(rb);
        size_t rate = (size_t)1 << rb;
        gl_t zh[256];
        for (size_t i = 0; i < rate; i++) { zh[i] = gl_sub(gl_mul(gn, gl_pow(wr, i)), 1); zh_inv[i] = gl_inv(zh[i]); }
        gl_t wl = gl_root_of_unity(L), n_f = (gl_t)n;
        const size_t nterms = nch + nch * nchunks + c->num_gate_constraints;
#pragma omp parallel
        {
            gl_t *terms = (gl_t *)malloc(sizeof(gl_t) * nterms), *num = (gl_t *)malloc(sizeof(gl_t) * R), *den = (gl_t *)malloc(sizeof(gl_t) * R);
#pragma omp for schedule(static)
            for (long iq = 0; iq < (long)q_n; iq++) {
                const size_t i = (size_t)iq * step;   /* index on the LDE domain */
                gl_t x = gl_mul(GL_MULT_GEN, gl_pow(wl, (uint64_t)i));
                const gl_t *w_row = batch_lde_row(&wb, i), *cs_row = batch_lde_row(&c->cs, i), *z_row = batch_lde_row(&zb, i);
                const gl_t *z_next = batch_lde_row(&zb, (i + rate) % lde_n);
                gl_t l0 = gl_mul(zh[i % rate], gl_inv(gl_mul(n_f, gl_sub(x, 1))));
                size_t t = 0;
                for (size_t k = 0; k < nch; k++) terms[t++] = gl_mul(l0, gl_sub(z_row[k], 1));
                for (size_t k = 0; k < nch; k++) {
                    for (size_t j = 0; j < R; j++) {
                        num[j] = gl_add(gl_add(w_row[j], gl_mul(betas[k], gl_mul(c->k_is[j], x))), gammas[k]);
                        den[j] = gl_add(gl_add(w_row[j], gl_mul(betas[k], cs_row[sig0 + j])), gammas[k]);
                    }
                    for (size_t cc = 0; cc < nchunks; cc++) {
                        gl_t prev = cc == 0 ? z_row[k] : z_row[nch + k * npp + cc - 1];
                        gl_t next = cc == nchunks - 1 ? z_next[k] : z_row[nch + k * npp + cc];
                        gl_t pn = 1, pd = 1;
                        for (size_t j = cc * chunk; j < (cc + 1) * chunk && j < R; j++) { pn = gl_mul(pn, num[j]); pd = gl_mul(pd, den[j]); }
                        terms[t++] = gl_sub(gl_mul(prev, pn), gl_mul(next, pd));
                    }
                }
                for (size_t g = 0; g < c->num_gate_constraints; g++) terms[t + g] = 0;
                eval_gates_base(c, cs_row, w_row, pih, terms + t);
                for (size_t k = 0; k < nch; k++) {
                    gl_t acc = 0;   /* reduce_with_powers: sum terms[j] * alpha^j */
                    for (size_t j = nterms; j-- > 0;) acc = gl_add(gl_mul(acc, alphas[k]), terms[j]);
                    quot[k * q_n + (size_t)iq] = gl_mul(acc, zh_inv[i % rate]);
                }
            }
            free(terms); free(num); free(den);
        }
    }
    trace_put("quotient_values", quot, nch * q_n);
#pragma omp parallel for
    for (long k = 0; k < (long)nch; k++) orc_coset_ifft(quot + (size_t)k * q_n, d + qbits, GL_MULT_GEN);
    /* each quotient poly (8n coefficients) is split into qdf chunks of n: contiguous => nch*qdf columns of n */
    const size_t nq = nch * c->qdf;
    orc_batch qb;
    g_oracle_index = 3;
    batch_from_coeffs(&qb, quot, nq, d, rb, ch_h);   /* quot now owned by qb.coeffs */
    trace_put("quotient_chunk_coeffs", qb.coeffs, nq * n);
    trace_put("quotient_cap", qb.cap, cap_words);
    orc_challenger_observe(&ch, qb.cap, cap_words);
    gl2_t zeta = challenger_get_ext(&ch);
    trace_put("zeta", zeta.c, 2);

    /* s7: openings */
    gl2_t g_zeta = gl2_scale(zeta, gl_root_of_unity(d));
    const orc_batch *oracles[4] = {&c->cs, &wb, &zb, &qb};
    const size_t n_open = ncs + NW + nzp + nq;
    gl2_t *open_zeta = (gl2_t *)malloc(sizeof(gl2_t) * n_open), *open_next = (gl2_t *)malloc(sizeof(gl2_t) * nch);
    {
        size_t off = 0;
        for (int o = 0; o < 4; o++) {
            const orc_batch *b = oracles[o];
#pragma omp parallel for schedule(dynamic)
            for (long p = 0; p < (long)b->ncols; p++) open_zeta[off + p] = eval_poly_ext(b->coeffs + (size_t)p * n, n, zeta);
            off += b->ncols;
        }
        for (size_t k = 0; k < nch; k++) open_next[k] = eval_poly_ext(zb.coeffs + k * n, n, g_zeta);
    }
    /* OpeningSet field order: constants, plonk_sigmas, wires, plonk_zs, plonk_zs_next, partial_products, quotient_polys */
    const gl2_t *o_consts = open_zeta, *o_wires = open_zeta + ncs, *o_zs = o_wires + NW, *o_pp = o_zs + nch, *o_q = o_pp + nch * npp;
    trace_put("openings_zeta", open_zeta, n_open * 2); trace_put("openings_zeta_next", open_next, nch * 2);
    /* observe_openings(to_fri_openings): batch 0 = everything at zeta in oracle order, batch 1 = zs_next */
    orc_challenger_observe(&ch, (const gl_t *)open_zeta, n_open * 2);
    orc_challenger_observe(&ch, (const gl_t *)open_next, nch * 2);

    /* s8: batched opening polynomial */
    gl2_t fri_alpha = challenger_get_ext(&ch);
    trace_put("fri_alpha", fri_alpha.c, 2);
    gl2_t *final_poly = (gl2_t *)calloc(lde_n, sizeof(gl2_t));   /* zero padded to 8n (the LDE of step s8) */
    {
        /* batch 0: all polynomials of all oracles at zeta; batch 1: Z polynomials at g*zeta */
        gl2_t *comp = (gl2_t *)malloc(sizeof(gl2_t) * n), *q0 = (gl2_t *)malloc(sizeof(gl2_t) * n);
        for (int batch = 0; batch < 2; batch++) {
            for (size_t i = 0; i < n; i++) comp[i] = gl2_from(0);
            gl2_t apow = gl2_from(1);
            size_t count = 0;
            if (batch == 0) {
                for (int o = 0; o < 4; o++) for (size_t p = 0; p < oracles[o]->ncols; p++) {
                    const gl_t *f = oracles[o]->coeffs + p * n;
                    for (size_t i = 0; i < n; i++) comp[i] = gl2_add(comp[i], gl2_scale(apow, f[i]));
                    apow = gl2_mul(apow, fri_alpha); count++;
                }
            } else {
                for (size_t p = 0; p < nch; p++) {
                    const gl_t *f = zb.coeffs + p * n;
                    for (size_t i = 0; i < n; i++) comp[i] = gl2_add(comp[i], gl2_scale(apow, f[i]));
                    apow = gl2_mul(apow, fri_alpha); count++;
                }
            }
            gl2_t z = batch == 0 ? zeta : g_zeta;
            /* divide_by_linear: synthetic division, quotient has n-1 coefficients, padded with a zero */
            gl2_t acc = gl2_from(0);
            for (size_t i = n; i-- > 0;) { acc = gl2_add(gl2_mul(acc, z), comp[i]); if (i > 0) q0[i - 1] = acc; }
            q0[n - 1] = gl2_from(0);
            /* alpha.shift_poly(final_poly): multiply by alpha^count, then add the quotient */
            gl2_t shift = gl2_pow(fri_alpha, count);
            for (size_t i = 0; i < n; i++) final_poly[i] = gl2_add(gl2_mul(final_poly[i], shift), q0[i]);
        }
        free(comp); free(q0);
    }
    trace_put("final_poly_coeffs", final_poly, n * 2);

    /* s9: FRI commit phase */
    wbuf fri_caps = {(uint8_t *)malloc(cap_words * 8 * (c->n_arity + 1)), cap_words * 8 * (c->n_arity + 1), 0, 0};
    size_t cur_len = lde_n;                 /* coefficient vector length (zero padded) */
    gl2_t *coeffs = final_poly;
    gl_t shift = GL_MULT_GEN;
    gl_t **tree_digests = (gl_t **)calloc(c->n_arity + 1, sizeof(gl_t *));
    gl_t **tree_leaves = (gl_t **)calloc(c->n_arity + 1, sizeof(gl_t *));
    size_t tree_nleaves[17];
    gl2_t *values = (gl2_t *)malloc(sizeof(gl2_t) * lde_n);
    {   /* values = coset_fft(coeffs, g) componentwise */
        gl_t *a = (gl_t *)malloc(sizeof(gl_t) * lde_n), *b = (gl_t *)malloc(sizeof(gl_t) * lde_n);
        for (size_t i = 0; i < lde_n; i++) { a[i] = coeffs[i].c[0]; b[i] = coeffs[i].c[1]; }
        orc_coset_fft(a, L, shift); orc_coset_fft(b, L, shift);
        for (size_t i = 0; i < lde_n; i++) values[i] = gl2_make(a[i], b[i]);
        free(a); free(b);
    }
    trace_put("fri_values0", values, lde_n * 2);
    for (size_t r = 0; r < c->n_arity; r++) {
        unsigned ab = (unsigned)c->arity[r], logc = 0;
        size_t arity = (size_t)1 << ab;
        while (((size_t)1 << logc) < cur_len) logc++;
        /* reverse_index_bits, chunk into leaves of `arity` extension values */
        size_t nl = cur_len >> ab;
        gl_t *leaves = (gl_t *)malloc(sizeof(gl_t) * cur_len * 2);
        for (size_t j = 0; j < cur_len; j++) { gl2_t v = values[bitrev32((uint32_t)j, logc)]; leaves[2 * j] = v.c[0]; leaves[2 * j + 1] = v.c[1]; }
        gl_t *dig = (gl_t *)malloc(sizeof(gl_t) * 4 * 2 * nl), capv[256 * 4];   /* cap_height <= 8 (reference common/src/circuit.rs:455-470) */
        orc_merkle_build(leaves, nl, 2 * arity, ch_h, dig, capv);
        tree_digests[r] = dig; tree_leaves[r] = leaves; tree_nleaves[r] = nl;
        w_vec(&fri_caps, capv, cap_words);
        orc_challenger_observe(&ch, capv, cap_words);
        gl2_t beta = challenger_get_ext(&ch);
        { char nm[32]; snprintf(nm, sizeof nm, "fri_beta%zu", r); trace_put(nm, beta.c, 2); snprintf(nm, sizeof nm, "fri_cap%zu", r); trace_put(nm, capv, cap_words); }
        /* fold coefficients: new[i] = sum_k beta^k * coeffs[arity*i + k] */
        size_t new_len = cur_len >> ab;
        for (size_t i = 0; i < new_len; i++) {
            gl2_t acc = gl2_from(0);
            for (size_t k = arity; k-- > 0;) acc = gl2_add(gl2_mul(acc, beta), coeffs[arity * i + k]);
            coeffs[i] = acc;
        }
        cur_len = new_len;
        shift = gl_pow(shift, arity);
        unsigned logn2 = logc - ab;
        gl_t *a = (gl_t *)malloc(sizeof(gl_t) * cur_len), *b = (gl_t *)malloc(sizeof(gl_t) * cur_len);
        for (size_t i = 0; i < cur_len; i++) { a[i] = coeffs[i].c[0]; b[i] = coeffs[i].c[1]; }
        orc_coset_fft(a, logn2, shift); orc_coset_fft(b, logn2, shift);
        for (size_t i = 0; i < cur_len; i++) values[i] = gl2_make(a[i], b[i]);
        free(a); free(b);
    }
    size_t final_len = cur_len >> rb;   /* coeffs.truncate(len >> rate_bits) */
    orc_challenger_observe(&ch, (const gl_t *)coeffs, final_len * 2);
    trace_put("fri_final_poly", coeffs, final_len * 2);

    /* s10: proof of work, minimum nonce */
    gl_t pow_witness = 0;
    {
        unsigned min_lz = (unsigned)c->pow_bits;
        /* windows of candidates searched in parallel; the smallest hit of the first window that has one is the minimum */
        const gl_t window = 1u << 14;
        for (gl_t base = 0; min_lz != 0; base += window) {
            gl_t best = ~(gl_t)0;
            #pragma omp parallel for schedule(static) reduction(min : best)
            for (gl_t k = 0; k < window; k++) {
                gl_t resp = orc_challenger_pow_response(&ch, base + k);
                if ((resp >> (64 - min_lz)) == 0 && base + k < best) best = base + k;
            }
            if (best != ~(gl_t)0) { pow_witness = best; break; }
        }
        orc_challenger_observe(&ch, &pow_witness, 1);
        gl_t resp = orc_challenger_get(&ch);
        (void)resp;
    }
    trace_put("pow_witness", &pow_witness, 1);

    /* s12 (first part): serialise */
    wbuf o = {out, cap, 0, 0};
    w_vec(&o, wb.cap, cap_words); w_vec(&o, zb.cap, cap_words); w_vec(&o, qb.cap, cap_words);
    for (size_t i = 0; i < ncs; i++) w_ext(&o, o_consts[i]);           /* constants then plonk_sigmas */
    for (size_t i = 0; i < NW; i++) w_ext(&o, o_wires[i]);
    for (size_t i = 0; i < nch; i++) w_ext(&o, o_zs[i]);
    for (size_t i = 0; i < nch; i++) w_ext(&o, open_next[i]);
    for (size_t i = 0; i < nch * npp; i++) w_ext(&o, o_pp[i]);
    for (size_t i = 0; i < nq; i++) w_ext(&o, o_q[i]);
    /* lookup_zs, lookup_zs_next: empty */
    for (size_t i = 0; i < fri_caps.len && !o.overflow; i++) w_u8(&o, fri_caps.p[i]);
    if (o.overflow) o.len += 0;

    /* s11: query rounds */
    uint64_t *qidx = (uint64_t *)malloc(8 * c->num_queries);
    for (size_t q = 0; q < c->num_queries; q++) {
        gl_t xch = orc_challenger_get(&ch);
        size_t x_index = (size_t)(xch % lde_n);
        qidx[q] = x_index;
        for (int oi = 0; oi < 4; oi++) {
            const orc_batch *b = oracles[oi];
            w_vec(&o, b->leaves + x_index * b->width, b->width);
            w_path(&o, b->digests, b->lde_n, ch_h, x_index);
        }
        for (size_t r = 0; r < c->n_arity; r++) {
            unsigned ab = (unsigned)c->arity[r];
            size_t arity = (size_t)1 << ab, leaf = x_index >> ab;
            w_vec(&o, tree_leaves[r] + leaf * 2 * arity, 2 * arity);
            w_path(&o, tree_digests[r], tree_nleaves[r], ch_h, leaf);
            x_index = leaf;
        }
    }
    trace_put("query_indices", qidx, c->num_queries);
    free(qidx);
    for (size_t i = 0; i < final_len; i++) w_ext(&o, coeffs[i]);
    w_u64(&o, pow_witness);
    w_vec(&o, public_inputs, c->num_pis);
    *len = o.len;

    for (size_t r = 0; r < c->n_arity; r++) { free(tree_digests[r]); free(tree_leaves[r]); }
    free(tree_digests); free(tree_leaves); free(values); free(final_poly); free(fri_caps.p);
    free(open_zeta); free(open_next); free(omega);
    batch_free(&wb); batch_free(&zb); batch_free(&qb);
    return o.overflow ? -1 : 0;
}
