/*
 * oracle/witness.c — CPU restatement of stage s1, `iop::generator::generate_partial_witness` of qp-plonky2 1.5.5 (reached
 * from the reference through ProverCircuitData::prove, wormhole/prover/src/lib.rs:171-175), over a circuit pack.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/gl.h). Restated from the published upstream algorithm, not from the device code:
 * a PartitionWitness (one slot per copy class; `set_target` on a slot that already holds another value is the
 * "partition set twice with different values" error the reference's negative tests expect,
 * wormhole/tests/src/circuit/block_header_tests.rs:34-95), the generators of the gates a row selects plus the pack's
 * free-standing ones (hint trailer), each run once all the targets it watches are set, to a fixpoint. Where the product
 * resolves the dependency order once per circuit into levels and lets one producer per copy class be the source, this file
 * does what plonky2 does: sweep the pending generators until nothing moves, every generator writing through set_target.
 *
 * Gate generators restated: ConstantGate, PublicInputGate's hash wires (the host value the prover binds), ArithmeticGate,
 * BaseSumGate<2> (BaseSplitGenerator), PoseidonGate, the fork's Poseidon2 gate (layout from the pack, LAYOUT UNPINNED),
 * ArithmeticExtensionGate, MulExtensionGate, RandomAccessGate, PoseidonMdsGate; hints: all seven opcodes of circuit.hpp.
 * A pack that selects any other gate returns ORC_WIT_UNSUPPORTED.
 */
#include <stdlib.h>
#include <string.h>
#include "plonk.h"
#include "poseidon2.h"

enum { ORC_WIT_OK = 0, ORC_WIT_CONFLICT = 1, ORC_WIT_INCOMPLETE = 2, ORC_WIT_UNSUPPORTED = 3, ORC_WIT_BAD_PACK = 4,
       ORC_WIT_ZERO_INVERSE = 5 /* a generator divided by zero (plonky2's Field::inverse panics: "Tried to invert zero") */ };

void orc_poseidon_round_constants(gl_t *out);
int orc_prove_many_entry(const orc_circuit *c, const gl_t *wires, const gl_t *public_inputs, uint8_t *out, size_t cap, size_t *len);   /* prove.c: orc_prove without the stage trace */

typedef struct {
    const orc_circuit *c;
    size_t n, nw, r;
    uint32_t *cls;          /* [nw * n] flat cell (col * n + row) -> slot */
    gl_t *val; uint8_t *set; /* per slot */
    int conflict; uint64_t conflict_cell; int derive_pis;
    int zero_inverse; uint64_t zero_cell;      /* the first generator that inverted zero, named by the cell it read */
} pw_t;

static inline size_t flat(const pw_t *p, size_t row, size_t col) { return col * p->n + row; }
static int pw_get(const pw_t *p, size_t row, size_t col, gl_t *out) {
    const uint32_t s = p->cls[flat(p, row, col)];
    if (!p->set[s]) return 0;
    *out = p->val[s];
    return 1;
}
static void pw_set(pw_t *p, size_t row, size_t col, gl_t v) {
    const uint32_t s = p->cls[flat(p, row, col)];
    v = gl_from_u64(v);
    if (p->set[s]) {
        if (p->val[s] != v && !p->conflict) { p->conflict = 1; p->conflict_cell = row * p->nw + col; }
        return;
    }
    p->set[s] = 1; p->val[s] = v;
}

/* ---- copy classes from sigma: sigma(row, col) = k_is[col'] * w^row' ---- */
typedef struct { gl_t key; uint32_t row; } row_ent;
static int row_cmp(const void *a, const void *b) { const gl_t x = ((const row_ent *)a)->key, y = ((const row_ent *)b)->key; return x < y ? -1 : x > y; }
static uint32_t uf_find(uint32_t *par, uint32_t x) { while (par[x] != x) { par[x] = par[par[x]]; x = par[x]; } return x; }

static int build_classes(pw_t *p) {
    const orc_circuit *c = p->c;
    const size_t n = p->n, R = p->r, NW = p->nw, sig0 = c->num_selectors + c->num_constants;
    row_ent *rows = (row_ent *)malloc(sizeof(row_ent) * n);
    { gl_t w = gl_root_of_unity((unsigned)c->degree_bits), a = 1; for (size_t i = 0; i < n; i++) { rows[i].key = a; rows[i].row = (uint32_t)i; a = gl_mul(a, w); } }
    qsort(rows, n, sizeof(row_ent), row_cmp);
    gl_t *kn = (gl_t *)malloc(sizeof(gl_t) * R), *kinv = (gl_t *)malloc(sizeof(gl_t) * R);
    for (size_t j = 0; j < R; j++) { kn[j] = gl_exp_pow2(c->k_is[j], (unsigned)c->degree_bits); kinv[j] = gl_inv(c->k_is[j]); }
    uint32_t *par = (uint32_t *)malloc(sizeof(uint32_t) * NW * n);
    for (size_t i = 0; i < NW * n; i++) par[i] = (uint32_t)i;
    int ok = 1;
    for (size_t col = 0; col < R && ok; col++)
        for (size_t row = 0; row < n; row++) {
            const gl_t s = c->cs_values[(sig0 + col) * n + row], sn = gl_exp_pow2(s, (unsigned)c->degree_bits);
            size_t tc = 0;
            while (tc < R && kn[tc] != sn) tc++;
            if (tc == R) { ok = 0; break; }
            row_ent key = {gl_mul(s, kinv[tc]), 0};
            const row_ent *hit = (const row_ent *)bsearch(&key, rows, n, sizeof(row_ent), row_cmp);
            if (!hit) { ok = 0; break; }
            const uint32_t a = uf_find(par, (uint32_t)(col * n + row)), b = uf_find(par, (uint32_t)(tc * n + hit->row));
            if (a != b) par[a > b ? a : b] = a > b ? b : a;
        }
    for (size_t i = 0; i < NW * n; i++) p->cls[i] = uf_find(par, (uint32_t)i);
    free(par); free(rows); free(kn); free(kinv);
    return ok;
}

/* ---- the two hash gates' generators ---- */
static inline gl_t sbox7(gl_t x) { gl_t x2 = gl_sqr(x), x4 = gl_sqr(x2), x3 = gl_mul(x, x2); return gl_mul(x3, x4); }
static const gl_t MDS_C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
static void mds(gl_t s[12]) {
    gl_t o[12];
    for (int r = 0; r < 12; r++) {
        u128 acc = 0;
        for (int i = 0; i < 12; i++) acc += (u128)s[(i + r) % 12] * MDS_C[i];
        if (r == 0) acc += (u128)s[0] * 8;
        o[r] = gl_reduce128(acc);
    }
    memcpy(s, o, sizeof o);
}
/* PoseidonGenerator (gates/poseidon.rs): swap, deltas, the S-box inputs of full rounds 1..3, of the 22 partial rounds (the
 * value of state[0] entering the S-box is the same in the textbook and the fast-basis schedules), of the last four full
 * rounds, and the outputs */
static void gen_poseidon(pw_t *p, size_t row, const gl_t in[12], gl_t swap) {
    static gl_t RC[360]; static int ready = 0;
    if (!ready) { orc_poseidon_round_constants(RC); ready = 1; }
    gl_t st[12];
    memcpy(st, in, sizeof st);
    for (int i = 0; i < 4; i++) {
        const gl_t delta = swap ? gl_sub(in[i + 4], in[i]) : 0;
        pw_set(p, row, 25 + i, delta);
        st[i] = gl_add(in[i], delta); st[i + 4] = gl_sub(in[i + 4], delta);
    }
    int rc = 0;
    for (int r = 0; r < 4; r++, rc++) {
        for (int i = 0; i < 12; i++) st[i] = gl_add(st[i], RC[rc * 12 + i]);
        if (r) for (int i = 0; i < 12; i++) pw_set(p, row, 29 + 12 * (r - 1) + i, st[i]);
        for (int i = 0; i < 12; i++) st[i] = sbox7(st[i]);
        mds(st);
    }
    for (int r = 0; r < 22; r++, rc++) {
        for (int i = 0; i < 12; i++) st[i] = gl_add(st[i], RC[rc * 12 + i]);
        pw_set(p, row, 65 + r, st[0]);
        st[0] = sbox7(st[0]);
        mds(st);
    }
    for (int r = 0; r < 4; r++, rc++) {
        for (int i = 0; i < 12; i++) st[i] = gl_add(st[i], RC[rc * 12 + i]);
        for (int i = 0; i < 12; i++) pw_set(p, row, 87 + 12 * r + i, st[i]);
        for (int i = 0; i < 12; i++) st[i] = sbox7(st[i]);
        mds(st);
    }
    for (int i = 0; i < 12; i++) pw_set(p, row, 12 + i, st[i]);
}
static void p2_ext(const orc_p2_params *q, gl_t s[12]) {
    gl_t t[12];
    for (int b = 0; b < 3; b++)
        for (int i = 0; i < 4; i++) {
            u128 acc = 0;
            for (int j = 0; j < 4; j++) acc += (u128)q->m4[i][j] * s[4 * b + j];
            t[4 * b + i] = gl_reduce128(acc);
        }
    for (int i = 0; i < 4; i++) {
        const gl_t sum = gl_add(gl_add(t[i], t[4 + i]), t[8 + i]);
        for (int b = 0; b < 3; b++) s[4 * b + i] = gl_add(t[4 * b + i], sum);
    }
}
static void p2_int(const orc_p2_params *q, gl_t s[12]) {
    gl_t sum = 0;
    for (int i = 0; i < 12; i++) sum = gl_add(sum, s[i]);
    for (int i = 0; i < 12; i++) s[i] = gl_add(gl_mul(s[i], q->diag_m1[i]), sum);
}
/* generator of the fork's Poseidon2 gate under the pack's wire layout (ten words, circuit.hpp "P2GL1") */
static void gen_poseidon2(pw_t *p, size_t row, const uint64_t *lay, const gl_t in[12], gl_t swap) {
    static orc_p2_params Q; static int ready = 0;
    if (!ready) { orc_p2_qp_params(&Q); ready = 1; }
    gl_t st[12];
    memcpy(st, in, sizeof st);
    if (lay[2] != 0xFFFFFFFFULL)
        for (int i = 0; i < 4; i++) {
            const gl_t delta = swap ? gl_sub(in[i + 4], in[i]) : 0;
            pw_set(p, row, lay[3] + i, delta);
            st[i] = gl_add(in[i], delta); st[i + 4] = gl_sub(in[i + 4], delta);
        }
    p2_ext(&Q, st);
    size_t rec = lay[4];
    for (int r = 0; r < 4; r++) {
        for (int i = 0; i < 12; i++) st[i] = gl_add(st[i], Q.rc_ext[r][i]);
        if (r || lay[7]) { for (int i = 0; i < 12; i++) pw_set(p, row, rec + i, st[i]); rec += 12; }
        for (int i = 0; i < 12; i++) st[i] = sbox7(st[i]);
        p2_ext(&Q, st);
    }
    for (int r = 0; r < 22; r++) {
        st[0] = gl_add(st[0], Q.rc_int[r]);
        pw_set(p, row, lay[5] + r, st[0]);
        st[0] = sbox7(st[0]);
        p2_int(&Q, st);
    }
    for (int r = 0; r < 4; r++) {
        for (int i = 0; i < 12; i++) st[i] = gl_add(st[i], Q.rc_ext[4 + r][i]);
        for (int i = 0; i < 12; i++) pw_set(p, row, lay[6] + 12 * r + i, st[i]);
        for (int i = 0; i < 12; i++) st[i] = sbox7(st[i]);
        p2_ext(&Q, st);
    }
    for (int i = 0; i < 12; i++) pw_set(p, row, lay[1] + i, st[i]);
}

/* one generator instance: a gate row's operation `op`, or hint `op` (gate = -1). Returns 1 when it ran, 0 when a watched
 * target is still unset */
typedef struct { int32_t gate; uint32_t row, op; } gen_t;

static int run_hint(pw_t *p, const uint64_t *h) {
    const size_t NW = p->nw;
#define HR(x) ((size_t)((x) / NW))
#define HC(x) ((size_t)((x) % NW))
    gl_t a, b, c2, d;
    switch (h[0]) {
    case 1: if (!pw_get(p, HR(h[2]), HC(h[2]), &a)) return 0; pw_set(p, HR(h[1]), HC(h[1]), a); return 1;
    case 2:
        if (!pw_get(p, HR(h[1]), HC(h[1]), &a) || !pw_get(p, HR(h[2]), HC(h[2]), &b)) return 0;
        pw_set(p, HR(h[3]), HC(h[3]), a == b ? 1 : 0);
        pw_set(p, HR(h[4]), HC(h[4]), a == b ? 0 : gl_inv(gl_sub(a, b)));
        return 1;
    case 3: if (!pw_get(p, HR(h[1]), HC(h[1]), &a)) return 0; pw_set(p, HR(h[2]), HC(h[2]), (a >> h[3]) & ((1ULL << h[4]) - 1)); return 1;
    case 4: {
        if (!pw_get(p, HR(h[1]), HC(h[1]), &a) || !pw_get(p, HR(h[2]), HC(h[2]), &b) || !pw_get(p, HR(h[3]), HC(h[3]), &c2) || !pw_get(p, HR(h[4]), HC(h[4]), &d)) return 0;
        if (c2 == 0 && d == 0 && !p->zero_inverse) { p->zero_inverse = 1; p->zero_cell = h[3]; }
        const gl2_t q = gl2_mul(gl2_make(a, b), gl2_inv(gl2_make(c2, d)));
        pw_set(p, HR(h[5]), HC(h[5]), q.c[0]); pw_set(p, HR(h[6]), HC(h[6]), q.c[1]);
        return 1;
    }
    case 5: pw_set(p, HR(h[1]), HC(h[1]), h[2]); return 1;
    case 6: if (!pw_get(p, HR(h[1]), HC(h[1]), &a)) return 0; pw_set(p, HR(h[2]), HC(h[2]), a == 0 ? 1 : gl_inv(a)); return 1;
    case 7: if (!pw_get(p, HR(h[1]), HC(h[1]), &a)) return 0; pw_set(p, HR(h[2]), HC(h[2]), a & ((1ULL << h[4]) - 1)); pw_set(p, HR(h[3]), HC(h[3]), a >> h[4]); return 1;
    default: return 1;
    }
#undef HR
#undef HC
}

static int run_gate(pw_t *p, const gen_t *g, const gl_t pih[4]) {
    const orc_circuit *c = p->c;
    const orc_gate *gt = &c->gates[g->gate];
    const size_t row = g->row, n = p->n;
    const gl_t *consts = c->cs_values + c->num_selectors * n + row;      /* constant i at consts[i * n] */
    gl_t in[64];
    switch (gt->type) {
    case OG_CONSTANT: pw_set(p, row, g->op, consts[(size_t)g->op * n]); return 1;
    case OG_PUBLIC_INPUT: if (!p->derive_pis) for (int i = 0; i < 4; i++) pw_set(p, row, i, pih[i]); return 1;
    case OG_ARITHMETIC: {
        const size_t b = 4 * (size_t)g->op;
        for (int i = 0; i < 3; i++) if (!pw_get(p, row, b + i, &in[i])) return 0;
        pw_set(p, row, b + 3, gl_add(gl_mul(gl_mul(in[0], in[1]), consts[0]), gl_mul(in[2], consts[n])));
        return 1;
    }
    case OG_ARITHMETIC_EXT: {
        const size_t b = 8 * (size_t)g->op;
        for (int i = 0; i < 6; i++) if (!pw_get(p, row, b + i, &in[i])) return 0;
        const gl2_t o = gl2_add(gl2_scale(gl2_mul(gl2_make(in[0], in[1]), gl2_make(in[2], in[3])), consts[0]), gl2_scale(gl2_make(in[4], in[5]), consts[n]));
        pw_set(p, row, b + 6, o.c[0]); pw_set(p, row, b + 7, o.c[1]);
        return 1;
    }
    case OG_MUL_EXT: {
        const size_t b = 6 * (size_t)g->op;
        for (int i = 0; i < 4; i++) if (!pw_get(p, row, b + i, &in[i])) return 0;
        const gl2_t o = gl2_scale(gl2_mul(gl2_make(in[0], in[1]), gl2_make(in[2], in[3])), consts[0]);
        pw_set(p, row, b + 4, o.c[0]); pw_set(p, row, b + 5, o.c[1]);
        return 1;
    }
    case OG_BASE_SUM: {
        if (!pw_get(p, row, 0, &in[0])) return 0;
        for (uint64_t i = 0; i < gt->param0; i++) pw_set(p, row, 1 + i, (in[0] >> i) & 1);
        return 1;
    }
    case OG_POSEIDON: {
        gl_t swap;
        for (int i = 0; i < 12; i++) if (!pw_get(p, row, i, &in[i])) return 0;
        if (!pw_get(p, row, 24, &swap)) return 0;
        gen_poseidon(p, row, in, swap);
        return 1;
    }
    case OG_POSEIDON2: {
        const uint64_t *lay = c->p2_layout;
        gl_t swap = 0;
        for (int i = 0; i < 12; i++) if (!pw_get(p, row, lay[0] + i, &in[i])) return 0;
        if (lay[2] != 0xFFFFFFFFULL && !pw_get(p, row, lay[2], &swap)) return 0;
        gen_poseidon2(p, row, lay, in, swap);
        return 1;
    }
    case OG_RANDOM_ACCESS: {
        const size_t bits = gt->param0, copies = gt->param1, extra = gt->reserved, vec = (size_t)1 << bits, routed = (2 + vec) * copies + extra;
        if (g->op == copies) { for (size_t i = 0; i < extra; i++) pw_set(p, row, (2 + vec) * copies + i, consts[i * n]); return 1; }
        const size_t b0 = (2 + vec) * g->op;
        gl_t idx, v;
        if (!pw_get(p, row, b0, &idx)) return 0;
        idx &= vec - 1;
        if (!pw_get(p, row, b0 + 2 + idx, &v)) return 0;
        for (size_t i = 0; i < vec; i++) { gl_t t; if (!pw_get(p, row, b0 + 2 + i, &t)) return 0; }
        pw_set(p, row, b0 + 1, v);
        for (size_t i = 0; i < bits; i++) pw_set(p, row, routed + g->op * bits + i, (idx >> i) & 1);
        return 1;
    }
    case OG_REDUCING: case OG_REDUCING_EXT: {
        /* ReducingGenerator (gates/reducing.rs, reducing_extension.rs): output 0..2, alpha 2..4, old_acc 4..6, coefficients from 6,
         * the running accumulators after them; acc <- acc * alpha + coefficient, the last one is the output */
        const int ext = gt->type == OG_REDUCING_EXT;
        const size_t nc = gt->param0, accs = 6 + (ext ? 2 * nc : nc);
        gl_t al[2], ac[2], co[2] = {0, 0};
        for (int e = 0; e < 2; e++) if (!pw_get(p, row, 2 + e, &al[e]) || !pw_get(p, row, 4 + e, &ac[e])) return 0;
        for (size_t i = 0; i < (ext ? 2 * nc : nc); i++) { gl_t t; if (!pw_get(p, row, 6 + i, &t)) return 0; }
        gl2_t acc = gl2_make(ac[0], ac[1]);
        const gl2_t alpha = gl2_make(al[0], al[1]);
        for (size_t i = 0; i < nc; i++) {
            if (ext) { pw_get(p, row, 6 + 2 * i, &co[0]); pw_get(p, row, 7 + 2 * i, &co[1]); } else { pw_get(p, row, 6 + i, &co[0]); co[1] = 0; }
            acc = gl2_add(gl2_mul(acc, alpha), gl2_make(co[0], co[1]));
            if (i + 1 < nc) { pw_set(p, row, accs + 2 * i, acc.c[0]); pw_set(p, row, accs + 2 * i + 1, acc.c[1]); }
            else { pw_set(p, row, 0, acc.c[0]); pw_set(p, row, 1, acc.c[1]); }
        }
        return 1;
    }
    case OG_COSET_INTERP: {
        /* InterpolationGenerator (gates/coset_interpolation.rs): shifted point = point / shift; barycentric interpolation over the subgroup
         * of order 2^bits in chunks of `degree` points, the running evaluation and product written at every chunk boundary */
        const size_t bits = gt->param0, deg = gt->param1, np = (size_t)1 << bits, ni = (np - 2) / (deg - 1);
        const size_t s_ep = 1 + 2 * np, s_ev = s_ep + 2, s_int = s_ev + 2;
        gl_t shift, e[2], v[64];
        if (!pw_get(p, row, 0, &shift) || !pw_get(p, row, s_ep, &e[0]) || !pw_get(p, row, s_ep + 1, &e[1])) return 0;
        for (size_t i = 0; i < 2 * np; i++) if (!pw_get(p, row, 1 + i, &v[i])) return 0;
        if (shift == 0 && !p->zero_inverse) { p->zero_inverse = 1; p->zero_cell = row * p->nw; }
        const gl2_t sp = gl2_scale(gl2_make(e[0], e[1]), gl_inv(shift));
        pw_set(p, row, s_int + 4 * ni, sp.c[0]); pw_set(p, row, s_int + 4 * ni + 1, sp.c[1]);
        const gl_t omega = gl_root_of_unity((unsigned)bits), inv_n = gl_inv((gl_t)np);
        gl2_t ev = gl2_make(0, 0), pr = gl2_make(1, 0);
        gl_t x = 1;
        size_t lo = 0, hi = deg;
        for (size_t cidx = 0; cidx <= ni; cidx++) {
            for (size_t q = lo; q < hi; q++) {
                const gl2_t term = gl2_make(gl_sub(sp.c[0], x), sp.c[1]);
                const gl2_t t = gl2_scale(gl2_mul(gl2_make(v[2 * q], v[2 * q + 1]), pr), gl_mul(x, inv_n));
                ev = gl2_add(gl2_mul(ev, term), t);
                pr = gl2_mul(pr, term);
                x = gl_mul(x, omega);
            }
            if (cidx == ni) break;
            pw_set(p, row, s_int + 2 * cidx, ev.c[0]); pw_set(p, row, s_int + 2 * cidx + 1, ev.c[1]);
            pw_set(p, row, s_int + 2 * (ni + cidx), pr.c[0]); pw_set(p, row, s_int + 2 * (ni + cidx) + 1, pr.c[1]);
            lo = 1 + (deg - 1) * (cidx + 1); hi = lo + deg - 1 < np ? lo + deg - 1 : np;
        }
        pw_set(p, row, s_ev, ev.c[0]); pw_set(p, row, s_ev + 1, ev.c[1]);
        return 1;
    }
    case OG_POSEIDON_MDS: {
        for (int i = 0; i < 24; i++) if (!pw_get(p, row, i, &in[i])) return 0;
        for (int comp = 0; comp < 2; comp++) {
            gl_t st[12];
            for (int i = 0; i < 12; i++) st[i] = in[2 * i + comp];
            mds(st);
            for (int i = 0; i < 12; i++) pw_set(p, row, 24 + 2 * i + comp, st[i]);
        }
        return 1;
    }
    default: return 1;
    }
}
static uint32_t gate_instances(const orc_gate *g) {
    switch (g->type) {
    case OG_NOOP: return 0;
    case OG_CONSTANT: case OG_ARITHMETIC: case OG_ARITHMETIC_EXT: case OG_MUL_EXT: return (uint32_t)g->param0;
    case OG_RANDOM_ACCESS: return (uint32_t)g->param1 + (g->reserved ? 1 : 0);
    default: return 1;
    }
}

/* ---- a circuit prepared for witness generation: what plonky2 holds in ProverOnlyCircuitData (the partition of the
 * targets, the generator list); built once per circuit, outside any timed region ---- */
typedef struct {
    orc_circuit *c;
    uint64_t *words; size_t n_words;          /* own copy of the pack (the trailers are read in place) */
    const uint64_t *hints, *pi_cells; size_t n_hints, n_pi;
    uint32_t *cls;
    gen_t *gens; size_t n_gens;
} orc_witness_plan;

void orc_witness_plan_free(orc_witness_plan *w) {
    if (!w) return;
    free(w->cls); free(w->gens); free(w->words);
    orc_circuit_free(w->c);
    free(w);
}
const orc_circuit *orc_witness_plan_circuit(const orc_witness_plan *w) { return w->c; }

/* returns NULL for a pack the oracle rejects; *rc_out says why (ORC_WIT_BAD_PACK / ORC_WIT_UNSUPPORTED) */
orc_witness_plan *orc_witness_plan_create(const uint64_t *words_in, size_t n_words, int *rc_out) {
    int rc = ORC_WIT_OK;
    orc_witness_plan *w = (orc_witness_plan *)calloc(1, sizeof *w);
    w->words = (uint64_t *)malloc(n_words * 8); w->n_words = n_words;
    memcpy(w->words, words_in, n_words * 8);
    const uint64_t *words = w->words;
    orc_circuit *c = w->c = orc_circuit_load(words, n_words);
    if (!c) { if (rc_out) *rc_out = ORC_WIT_BAD_PACK; free(w->words); free(w); return NULL; }
    const size_t n = (size_t)1 << c->degree_bits, NW = c->num_wires, R = c->num_routed, ncs = c->num_selectors + c->num_constants + c->num_routed;
    {   /* trailers: hints and public-input cells */
        size_t q = 18 + c->n_arity + 8 * c->n_gates + R + 4 + ncs * n;
        while (q + 2 <= n_words) {
            const uint64_t magic = words[q], cnt = words[q + 1];
            if (magic == 0x00000031544E4948ULL) { w->hints = words + q + 2; w->n_hints = cnt; q += 2 + 8 * cnt; }
            else if (magic == 0x0000003149425550ULL) { w->pi_cells = words + q + 2; w->n_pi = cnt; q += 2 + cnt; }
            else q += 2 + cnt;
        }
    }
    for (size_t i = 0; i < c->n_gates; i++) {
        const uint64_t t = c->gates[i].type;
        if (t == OG_EXPONENTIATION) rc = ORC_WIT_UNSUPPORTED;
    }
    pw_t p;
    memset(&p, 0, sizeof p);
    p.c = c; p.n = n; p.nw = NW; p.r = R;
    p.cls = w->cls = (uint32_t *)malloc(sizeof(uint32_t) * NW * n);
    if (rc == ORC_WIT_OK && !build_classes(&p)) rc = ORC_WIT_BAD_PACK;
    if (rc == ORC_WIT_OK) {
        /* the generator list: per row the selected gate's instances, then the hints */
        size_t cap = w->n_hints;
        int32_t *gate_of_row = (int32_t *)malloc(sizeof(int32_t) * n);
        for (size_t r = 0; r < n; r++) {
            gate_of_row[r] = -1;
            for (size_t gi = 0; gi < c->n_gates; gi++)
                if (c->cs_values[c->gates[gi].selector_index * n + r] == gi) { gate_of_row[r] = (int32_t)gi; break; }
            if (gate_of_row[r] < 0) { rc = ORC_WIT_BAD_PACK; break; }
            cap += gate_instances(&c->gates[gate_of_row[r]]);
        }
        if (rc == ORC_WIT_OK) {
            w->gens = (gen_t *)malloc(sizeof(gen_t) * (cap ? cap : 1));
            for (size_t r = 0; r < n; r++)
                for (uint32_t op = 0, k = gate_instances(&c->gates[gate_of_row[r]]); op < k; op++) w->gens[w->n_gens++] = (gen_t){gate_of_row[r], (uint32_t)r, op};
            for (size_t h = 0; h < w->n_hints; h++) w->gens[w->n_gens++] = (gen_t){-1, 0, (uint32_t)h};
        }
        free(gate_of_row);
    }
    if (rc_out) *rc_out = rc;
    if (rc != ORC_WIT_OK) { orc_witness_plan_free(w); return NULL; }
    return w;
}

/* cells / values: the PartialWitness (cell = row * num_wires + wire); public_inputs go to the pack's public-input cells
 * ("PUBI1"). wires_out: num_wires x n column-major, unset targets 0. conflict_cell_out: the target that was set twice. */
int orc_witness_generate(const orc_witness_plan *w, const uint64_t *cells, const gl_t *values, size_t count, const gl_t *public_inputs,
                         gl_t *wires_out, uint64_t *conflict_cell_out) {
    const orc_circuit *c = w->c;
    const size_t n = (size_t)1 << c->degree_bits, NW = c->num_wires;
    int rc = ORC_WIT_OK;
    pw_t p;
    memset(&p, 0, sizeof p);
    p.c = c; p.n = n; p.nw = NW; p.r = c->num_routed; p.cls = w->cls;
    p.val = (gl_t *)calloc(NW * n, sizeof(gl_t));
    p.set = (uint8_t *)calloc(NW * n, 1);
    gl_t pih[4] = {0, 0, 0, 0};
    /* public_inputs == NULL: plonky2's own order of things — the public inputs are whatever the generators make of the
     * public-input targets; the PublicInputGate's wires get the hash through their copy constraints (tests read the public
     * inputs back from the trace) */
    p.derive_pis = public_inputs == NULL;
    if (c->num_pis && public_inputs) orc_hash_no_pad(public_inputs, c->num_pis, pih);
    for (size_t i = 0; public_inputs && i < w->n_pi && i < c->num_pis; i++) pw_set(&p, w->pi_cells[i] / NW, w->pi_cells[i] % NW, public_inputs[i]);
    for (size_t i = 0; i < count; i++) {
        if (cells[i] >= NW * n) { rc = ORC_WIT_BAD_PACK; break; }
        pw_set(&p, cells[i] / NW, cells[i] % NW, values[i]);
    }
    if (rc == ORC_WIT_OK) {
        gen_t *pending = (gen_t *)malloc(sizeof(gen_t) * (w->n_gens ? w->n_gens : 1));
        size_t np = w->n_gens;
        memcpy(pending, w->gens, sizeof(gen_t) * np);
        /* generate_partial_witness: run whatever is ready until nothing is */
        for (;;) {
            size_t kept = 0;
            for (size_t i = 0; i < np; i++) {
                const int ran = pending[i].gate < 0 ? run_hint(&p, w->hints + 8 * (size_t)pending[i].op) : run_gate(&p, &pending[i], pih);
                if (!ran) pending[kept++] = pending[i];
            }
            if (kept == np || kept == 0) { np = kept; break; }
            np = kept;
        }
        if (p.zero_inverse) rc = ORC_WIT_ZERO_INVERSE;     /* first: the value such a generator writes (inverse taken as 0) is what later conflicts come from */
        else if (p.conflict) rc = ORC_WIT_CONFLICT;
        else if (np) rc = ORC_WIT_INCOMPLETE;    /* generators left waiting: plonky2 would fail on the first unset target */
        free(pending);
    }
    if (conflict_cell_out) *conflict_cell_out = p.zero_inverse ? p.zero_cell : p.conflict ? p.conflict_cell : ~0ULL;
    if (wires_out) for (size_t i = 0; i < NW * n; i++) { const uint32_t s = p.cls[i]; wires_out[i] = p.set[s] ? p.val[s] : 0; }
    /* the slots held the spend secret */
    memset(p.val, 0, NW * n * sizeof(gl_t));
    free(p.val); free(p.set);
    return rc;
}

/* one-shot form: plan, generate, free */
int orc_generate_witness(const uint64_t *words, size_t n_words, const uint64_t *cells, const gl_t *values, size_t count,
                         const gl_t *public_inputs, gl_t *wires_out, uint64_t *conflict_cell_out) {
    int rc = ORC_WIT_OK;
    orc_witness_plan *w = orc_witness_plan_create(words, n_words, &rc);
    if (!w) { if (conflict_cell_out) *conflict_cell_out = ~0ULL; return rc; }
    rc = orc_witness_generate(w, cells, values, count, public_inputs, wires_out, conflict_cell_out);
    orc_witness_plan_free(w);
    return rc;
}

/* `commit().prove()` for `nproofs` PartialWitnesses over one cell list, ONE PROOF PER THREAD (bench.py's cpu_baseline): every
 * thread generates its witness and proves it serially — the inner loops' OpenMP regions are nested and run on the one thread —
 * which is how independent proofs use a many-core host best (no barrier per loop, no shared cache lines). values:
 * [nproofs][count], public_inputs: [nproofs][num_pis], outs: nproofs buffers of cap bytes. Returns the number of failures. */
int orc_commit_prove_many(const orc_witness_plan *w, size_t nproofs, const uint64_t *cells, size_t count, const gl_t *values,
                          const gl_t *public_inputs, uint8_t *outs, size_t cap) {
    const orc_circuit *c = w->c;
    const size_t words = (size_t)c->num_wires << c->degree_bits;
    int failures = 0;
#pragma omp parallel for schedule(dynamic) reduction(+ : failures)
    for (long i = 0; i < (long)nproofs; i++) {
        gl_t *wires = (gl_t *)malloc(words * sizeof(gl_t));
        size_t len = 0;
        const gl_t *pis = public_inputs + (size_t)i * c->num_pis;
        int rc = orc_witness_generate(w, cells, values + (size_t)i * count, count, pis, wires, NULL);
        if (rc == ORC_WIT_OK) rc = orc_prove_many_entry(c, wires, pis, outs + (size_t)i * cap, cap, &len);
        if (rc != 0 || len != cap) failures++;
        memset(wires, 0, words * sizeof(gl_t));
        free(wires);
    }
    return failures;
}
