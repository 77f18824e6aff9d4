/*
 * oracle/poseidon2_gate.c — CPU restatement of the qp fork's Poseidon2 GATE (the gate behind `hash_n_to_hash_no_pad_p2`;
 * reference call sites wormhole/circuit/src/zk_merkle_proof.rs:482,504,606, nullifier.rs:298-299,
 * unspendable_account.rs:229-231, block_header/mod.rs:66), prover side (base field, one trace point) and verifier side
 * (quadratic extension, at zeta).
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/gl.h).
 *
 * What is pinned and what is not: the PERMUTATION the gate constrains is qp-poseidon-core 3.1.0's Poseidon2
 * (orc_p2_qp_params, pinned by the reference's seven known-answer vectors, tests/test_oracle_poseidon.py). The gate's WIRE
 * LAYOUT and constraint order live in un-vendored qp-plonky2 1.5.5 and cannot be read offline: LAYOUT UNPINNED. It is
 * therefore data — the ten words of the circuit pack's "P2GL1" trailer (qp-zk-circuits_amd/csrc/circuit.hpp) — and the
 * default restates upstream plonky2's PoseidonGate with Poseidon2's round structure: inputs, outputs, swap bit, four deltas,
 * then one wire per S-box input that is not an affine function of wires already fixed (full rounds 1..3, the 22 internal
 * rounds' lane 0, full rounds 4..7): 135 wires, 123 constraints of degree 7 — the figures the reference's config comments
 * quote for its hash gate (common/src/circuit.rs:428-431,447-449).
 *
 * Constraint order (layout word 8 = 0): [swap (swap - 1)], [swap (rhs_i - lhs_i) - delta_i, i < 4], then round by round
 * `state after the round's constant addition - recorded S-box input`, and last `final state - output wire`.
 */
#include "poseidon2.h"
#include <string.h>

enum { L_IN = 0, L_OUT, L_SWAP, L_DELTA, L_FULL0, L_PARTIAL, L_FULL1, L_FIRST_ROUND, L_ORDER, L_END };
#define P2_NO_SWAP 0xFFFFFFFFULL

static orc_p2_params G;
static int g_ready = 0;
static const orc_p2_params *params(void) {
    if (!g_ready) { orc_p2_qp_params(&G); g_ready = 1; }
    return &G;
}

size_t orc_p2_gate_num_constraints(const uint64_t lay[10]) {
    return (lay[L_SWAP] != P2_NO_SWAP ? 5 : 0) + 12 * (lay[L_FIRST_ROUND] ? 4 : 3) + 22 + 48 + 12;
}

/* ---- base field ---- */
static inline gl_t sbox(gl_t x) { gl_t x2 = gl_sqr(x), x3 = gl_mul(x2, x), x4 = gl_sqr(x2); return gl_mul(x3, x4); }
static void external_b(const orc_p2_params *p, gl_t s[12]) {
    gl_t t[12];
    for (int b = 0; b < 3; b++)
        for (int i = 0; i < 4; i++) {
            gl_t acc = 0;
            for (int j = 0; j < 4; j++) acc = gl_add(acc, gl_mul(p->m4[i][j], s[4 * b + j]));
            t[4 * b + i] = acc;
        }
    for (int i = 0; i < 4; i++) {
        const gl_t cs = gl_add(gl_add(t[i], t[4 + i]), t[8 + i]);
        for (int b = 0; b < 3; b++) s[4 * b + i] = gl_add(t[4 * b + i], cs);
    }
}
static void internal_b(const orc_p2_params *p, gl_t s[12]) {
    gl_t tot = 0;
    for (int i = 0; i < 12; i++) tot = gl_add(tot, s[i]);
    for (int i = 0; i < 12; i++) s[i] = gl_add(gl_mul(s[i], p->diag_m1[i]), tot);
}
size_t orc_p2_gate_base(const uint64_t lay[10], const gl_t *w, gl_t *out) {
    const orc_p2_params *p = params();
    size_t k = 0;
    gl_t st[12];
    memcpy(st, w + lay[L_IN], sizeof st);
    if (lay[L_SWAP] != P2_NO_SWAP) {
        const gl_t swap = w[lay[L_SWAP]];
        out[k++] = gl_mul(swap, gl_sub(swap, 1));
        for (int i = 0; i < 4; i++) {
            const gl_t lhs = st[i], rhs = st[i + 4], delta = w[lay[L_DELTA] + i];
            out[k++] = gl_sub(gl_mul(swap, gl_sub(rhs, lhs)), delta);
            st[i] = gl_add(lhs, delta); st[i + 4] = gl_sub(rhs, delta);
        }
    }
    external_b(p, st);
    size_t rec = lay[L_FULL0];
    for (int r = 0; r < 4; r++) {
        for (int i = 0; i < 12; i++) st[i] = gl_add(st[i], p->rc_ext[r][i]);
        if (r > 0 || lay[L_FIRST_ROUND]) { for (int i = 0; i < 12; i++) { out[k++] = gl_sub(st[i], w[rec + i]); st[i] = w[rec + i]; } rec += 12; }
        for (int i = 0; i < 12; i++) st[i] = sbox(st[i]);
        external_b(p, st);
    }
    for (int r = 0; r < 22; r++) {
        st[0] = gl_add(st[0], p->rc_int[r]);
        out[k++] = gl_sub(st[0], w[lay[L_PARTIAL] + r]);
        st[0] = sbox(w[lay[L_PARTIAL] + r]);
        internal_b(p, st);
    }
    for (int r = 0; r < 4; r++) {
        const gl_t *rw = w + lay[L_FULL1] + 12 * r;
        for (int i = 0; i < 12; i++) st[i] = gl_add(st[i], p->rc_ext[4 + r][i]);
        for (int i = 0; i < 12; i++) { out[k++] = gl_sub(st[i], rw[i]); st[i] = rw[i]; }
        for (int i = 0; i < 12; i++) st[i] = sbox(st[i]);
        external_b(p, st);
    }
    for (int i = 0; i < 12; i++) out[k++] = gl_sub(st[i], w[lay[L_OUT] + i]);
    return k;
}

/* ---- quadratic extension (the verifier evaluates the same constraints at zeta) ---- */
static inline gl2_t sbox_e(gl2_t x) { gl2_t x2 = gl2_mul(x, x), x3 = gl2_mul(x2, x), x4 = gl2_mul(x2, x2); return gl2_mul(x3, x4); }
static void external_e(const orc_p2_params *p, gl2_t s[12]) {
    gl2_t t[12];
    for (int b = 0; b < 3; b++)
        for (int i = 0; i < 4; i++) {
            gl2_t acc = gl2_from(0);
            for (int j = 0; j < 4; j++) acc = gl2_add(acc, gl2_scale(s[4 * b + j], p->m4[i][j]));
            t[4 * b + i] = acc;
        }
    for (int i = 0; i < 4; i++) {
        const gl2_t cs = gl2_add(gl2_add(t[i], t[4 + i]), t[8 + i]);
        for (int b = 0; b < 3; b++) s[4 * b + i] = gl2_add(t[4 * b + i], cs);
    }
}
static void internal_e(const orc_p2_params *p, gl2_t s[12]) {
    gl2_t tot = gl2_from(0);
    for (int i = 0; i < 12; i++) tot = gl2_add(tot, s[i]);
    for (int i = 0; i < 12; i++) s[i] = gl2_add(gl2_scale(s[i], p->diag_m1[i]), tot);
}
size_t orc_p2_gate_ext(const uint64_t lay[10], const gl2_t *w, gl2_t *out) {
    const orc_p2_params *p = params();
    size_t k = 0;
    gl2_t st[12];
    memcpy(st, w + lay[L_IN], sizeof st);
    if (lay[L_SWAP] != P2_NO_SWAP) {
        const gl2_t swap = w[lay[L_SWAP]];
        out[k++] = gl2_mul(swap, gl2_sub(swap, gl2_from(1)));
        for (int i = 0; i < 4; i++) {
            const gl2_t lhs = st[i], rhs = st[i + 4], delta = w[lay[L_DELTA] + i];
            out[k++] = gl2_sub(gl2_mul(swap, gl2_sub(rhs, lhs)), delta);
            st[i] = gl2_add(lhs, delta); st[i + 4] = gl2_sub(rhs, delta);
        }
    }
    external_e(p, st);
    size_t rec = lay[L_FULL0];
    for (int r = 0; r < 4; r++) {
        for (int i = 0; i < 12; i++) st[i] = gl2_add(st[i], gl2_from(p->rc_ext[r][i]));
        if (r > 0 || lay[L_FIRST_ROUND]) { for (int i = 0; i < 12; i++) { out[k++] = gl2_sub(st[i], w[rec + i]); st[i] = w[rec + i]; } rec += 12; }
        for (int i = 0; i < 12; i++) st[i] = sbox_e(st[i]);
        external_e(p, st);
    }
    for (int r = 0; r < 22; r++) {
        st[0] = gl2_add(st[0], gl2_from(p->rc_int[r]));
        out[k++] = gl2_sub(st[0], w[lay[L_PARTIAL] + r]);
        st[0] = sbox_e(w[lay[L_PARTIAL] + r]);
        internal_e(p, st);
    }
    for (int r = 0; r < 4; r++) {
        const gl2_t *rw = w + lay[L_FULL1] + 12 * r;
        for (int i = 0; i < 12; i++) st[i] = gl2_add(st[i], gl2_from(p->rc_ext[4 + r][i]));
        for (int i = 0; i < 12; i++) { out[k++] = gl2_sub(st[i], rw[i]); st[i] = rw[i]; }
        for (int i = 0; i < 12; i++) st[i] = sbox_e(st[i]);
        external_e(p, st);
    }
    for (int i = 0; i < 12; i++) out[k++] = gl2_sub(st[i], w[lay[L_OUT] + i]);
    return k;
}
