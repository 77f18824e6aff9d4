// verify_math.hpp — the arithmetic of plonky2's verifier at ONE point, generic in the element type: every gate's unfiltered
// constraints in upstream order (gates/*.rs eval_unfiltered), the filtered sum over the circuit's gates, the permutation
// argument's terms and the reduction with the alphas (plonk/vanishing_poly.rs eval_vanishing_poly). Instantiated twice:
//   X = gl::e2                       the host verifier (verifier.cpp): field elements, the result is compared;
//   X = wrapper_circuit.cpp's XT     the same expressions over ExtensionTargets of the circuit builder — the in-circuit verifier
//                                    (plonk/vanishing_poly.rs eval_vanishing_poly_circuit, gates/*.rs eval_unfiltered_circuit), so
//                                    that the circuit enforces exactly what the host verifier checks.
// X needs: X + X, X - X, X * X, scale(X, u64), sadd(X a, u64 s, X c) = s a + c, madd(X a, X b, X c) = a b + c, and konst<X>(u64).
#pragma once
#include <algorithm>
#include <vector>
#include "circuit.hpp"
#include "gl64.hpp"
#include "poseidon.hpp"

namespace gl {
inline e2 operator+(e2 x, e2 y) { return e2_add(x, y); }
inline e2 operator-(e2 x, e2 y) { return e2_sub(x, y); }
inline e2 operator*(e2 x, e2 y) { return e2_mul(x, y); }
inline e2 scale(e2 x, u64 s) { return e2_scale(x, s); }
inline e2 sadd(e2 a, u64 s, e2 c) { return e2_add(e2_scale(a, s), c); }
inline e2 madd(e2 a, e2 b, e2 c) { return e2_add(e2_mul(a, b), c); }
}  // namespace gl

namespace vmath {
using gl::u64;

template <class X> X konst(u64 c);
template <> inline gl::e2 konst<gl::e2>(u64 c) { return gl::e2_from(c); }

// the extension ALGEBRA over the extension (wire pairs of the *Extension gates at zeta): c0 + c1 X, X^2 = 7, coefficients in X
template <class X> struct AlgT { X c0, c1; };
template <class X> AlgT<X> alg_mul(AlgT<X> a, AlgT<X> b) { return {madd(a.c0, b.c0, scale(a.c1 * b.c1, 7)), madd(a.c0, b.c1, a.c1 * b.c0)}; }

// ---- gate constraints at one point of the extension field ----
template <class X> X sbox7(X x) { const X x2 = x * x, x4 = x2 * x2; return (x * x2) * x4; }
template <class X> void mds_ext(X (&s)[12]) {   // the MDS matrix has base-field entries: out[r] = sum_i circ[i] s[(i + r) % 12] + diag[r] s[r]
    static const u64 CIRC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    X t[12];
    for (int r = 0; r < 12; r++) {
        X acc = r == 0 ? scale(s[0], CIRC[0] + 8) : scale(s[r], CIRC[0]);
        for (int i = 1; i < 12; i++) acc = sadd(s[(i + r) % 12], CIRC[i], acc);
        t[r] = acc;
    }
    for (int r = 0; r < 12; r++) s[r] = t[r];
}

// PoseidonGate (plonky2::gates::poseidon): wires 0..11 input, 12..23 output, 24 swap, 25..28 delta, 29..64 S-box inputs of
// full rounds 1..3, 65..86 of the 22 partial rounds, 87..134 of the last four full rounds; 123 constraints.
template <class X> void poseidon_gate(const X *w, X *out) {
    auto E = [](u64 c) { return konst<X>(c); };
    const u64 *rc = poseidon::host_round_constants(), *fp = poseidon::host_fast_partial();
    size_t k = 0;
    const X swap = w[24];
    X st[12];
    out[k++] = swap * (swap - E(1));
    for (int i = 0; i < 4; i++) out[k++] = swap * (w[i + 4] - w[i]) - w[25 + i];
    for (int i = 0; i < 4; i++) { st[i] = w[i] + w[25 + i]; st[i + 4] = w[i + 4] - w[25 + i]; }
    for (int i = 8; i < 12; i++) st[i] = w[i];
    int r_idx = 0;
    for (int r = 0; r < 4; r++, r_idx++) {
        for (int i = 0; i < 12; i++) st[i] = st[i] + E(rc[r_idx * 12 + i]);
        if (r) for (int i = 0; i < 12; i++) { const X in = w[29 + 12 * (r - 1) + i]; out[k++] = st[i] - in; st[i] = in; }
        for (int i = 0; i < 12; i++) st[i] = sbox7(st[i]);
        mds_ext(st);
    }
    for (int i = 0; i < 12; i++) st[i] = st[i] + E(fp[poseidon::FP_FIRST + i]);                       // partial_first_constant_layer
    {
        X t[11];
        for (int c = 0; c < 11; c++) { X acc = E(0); for (int r = 0; r < 11; r++) acc = sadd(st[1 + r], fp[poseidon::FP_INIT + c * 11 + r], acc); t[c] = acc; }
        for (int c = 0; c < 11; c++) st[1 + c] = t[c];                                                // mds_partial_layer_init
    }
    for (int r = 0; r < 22; r++) {
        const X in = w[65 + r];
        out[k++] = st[0] - in;
        const X s0 = sbox7(in) + E(fp[poseidon::FP_RC + r]);
        X d = scale(s0, poseidon::MDS_00);
        for (int i = 0; i < 11; i++) d = sadd(st[1 + i], fp[poseidon::FP_WHATS + r * 11 + i], d);
        for (int i = 0; i < 11; i++) st[1 + i] = sadd(s0, fp[poseidon::FP_VS + r * 11 + i], st[1 + i]);
        st[0] = d;                                                                                    // mds_partial_layer_fast
    }
    r_idx += 22;
    for (int r = 0; r < 4; r++, r_idx++) {
        for (int i = 0; i < 12; i++) st[i] = st[i] + E(rc[r_idx * 12 + i]);
        for (int i = 0; i < 12; i++) { const X in = w[87 + 12 * r + i]; out[k++] = st[i] - in; st[i] = in; }
        for (int i = 0; i < 12; i++) st[i] = sbox7(st[i]);
        mds_ext(st);
    }
    for (int i = 0; i < 12; i++) out[k++] = st[i] - w[12 + i];
}

// The qp fork's Poseidon2 gate (type 14) at one extension point, verifier side: wires as the pack's layout table places them
// (circuit.hpp P2GateLayout; default = upstream PoseidonGate's layout carried over, LAYOUT UNPINNED), permutation =
// qp-poseidon-core's Poseidon2 (pinned by the reference's known-answer vectors). The linear layers have base-field entries, so
// they act on an extension element coefficient-wise (scale); only the S-boxes multiply extension elements.
template <class X> void p2_external(X (&s)[12], const poseidon2::Params &P) {
    auto E = [](u64 c) { return konst<X>(c); };
    X t[12];
    for (int b = 0; b < 3; b++)
        for (int i = 0; i < 4; i++) {
            X acc = E(0);
            for (int j = 0; j < 4; j++) acc = sadd(s[4 * b + j], P.m4[4 * i + j], acc);
            t[4 * b + i] = acc;
        }
    for (int i = 0; i < 4; i++) {
        const X colsum = t[i] + t[4 + i] + t[8 + i];
        for (int b = 0; b < 3; b++) s[4 * b + i] = t[4 * b + i] + colsum;
    }
}
template <class X> void p2_internal(X (&s)[12], const poseidon2::Params &P) {
    auto E = [](u64 c) { return konst<X>(c); };
    X total = E(0);
    for (int i = 0; i < 12; i++) total = total + s[i];
    for (int i = 0; i < 12; i++) s[i] = sadd(s[i], P.diag_m1[i], total);
}
template <class X> size_t poseidon2_gate(const P2GateLayout &lay, const X *w, X *out) {
    auto E = [](u64 c) { return konst<X>(c); };
    const poseidon2::Params &P = poseidon2::qp_params();
    size_t k = 0;
    X st[12];
    for (int i = 0; i < 12; i++) st[i] = w[lay.w_input + i];
    if (lay.has_swap()) {
        const X swap = w[lay.w_swap];
        out[k++] = swap * (swap - E(1));
        for (int i = 0; i < 4; i++) {
            const X delta = w[lay.w_delta + i];
            out[k++] = swap * (st[i + 4] - st[i]) - delta;
            st[i] = st[i] + delta; st[i + 4] = st[i + 4] - delta;
        }
    }
    p2_external(st, P);
    uint32_t rec = lay.w_full0;
    for (int r = 0; r < 8; r++) {
        if (r == 4) {   // the 22 internal rounds sit between the two halves
            for (int q = 0; q < 22; q++) {
                const X in = w[lay.w_partial + q];
                out[k++] = st[0] + E(P.rc_int[q]) - in;
                st[0] = sbox7(in);
                p2_internal(st, P);
            }
            rec = lay.w_full1;
        }
        for (int i = 0; i < 12; i++) st[i] = st[i] + E(P.rc_ext[r * 12 + i]);
        if (r != 0 || lay.first_round_wires) {
            for (int i = 0; i < 12; i++) { const X in = w[rec + i]; out[k++] = st[i] - in; st[i] = in; }
            rec += 12;
        }
        for (int i = 0; i < 12; i++) st[i] = sbox7(st[i]);
        p2_external(st, P);
    }
    for (int i = 0; i < 12; i++) out[k++] = st[i] - w[lay.w_output + i];
    return k;
}

// the unfiltered constraints of gate g in upstream order; returns how many were written
template <class X> size_t gate_constraints(const GateInfo &g, const P2GateLayout &p2_layout, const X *consts, const X *w, const X pih[4], std::vector<X> &out) {
    auto E = [](u64 c) { return konst<X>(c); };
    using Alg = AlgT<X>;
    size_t k = 0;
    out.assign((size_t)g.num_constraints + 8, E(0));
    switch (g.type) {
        case GATE_NOOP: break;
        case GATE_CONSTANT:
            for (u64 i = 0; i < g.param0; i++) out[k++] = consts[i] - w[i];
            break;
        case GATE_PUBLIC_INPUT:
            for (int i = 0; i < 4; i++) out[k++] = w[i] - pih[i];
            break;
        case GATE_ARITHMETIC:           // per op: multiplicand_0, multiplicand_1, addend, output
            for (u64 i = 0; i < g.param0; i++) out[k++] = w[4 * i + 3] - ((w[4 * i] * w[4 * i + 1]) * consts[0] + w[4 * i + 2] * consts[1]);
            break;
        case GATE_POSEIDON:
            poseidon_gate(w, out.data());
            k = 123;
            break;
        case GATE_POSEIDON2:
            k = poseidon2_gate(p2_layout, w, out.data());
            break;
        case GATE_BASE_SUM: {           // wire 0 = sum, wires 1..num_limbs = bits (little endian)
            X s = E(0);
            for (u64 i = g.param0; i-- > 0;) s = (s + s) + w[1 + i];
            out[k++] = s - w[0];
            for (u64 i = 0; i < g.param0; i++) out[k++] = w[1 + i] * (w[1 + i] - E(1));
            break;
        }
        case GATE_ARITHMETIC_EXT:       // 8 wires per op: two multiplicands, addend, output, each an algebra element
            for (u64 i = 0; i < g.param0; i++) {
                const X *o = w + 8 * i;
                const Alg p = alg_mul(Alg{o[0], o[1]}, Alg{o[2], o[3]});
                out[k++] = o[6] - (p.c0 * consts[0] + o[4] * consts[1]);
                out[k++] = o[7] - (p.c1 * consts[0] + o[5] * consts[1]);
            }
            break;
        case GATE_MUL_EXT:              // 6 wires per op
            for (u64 i = 0; i < g.param0; i++) {
                const X *o = w + 6 * i;
                const Alg p = alg_mul(Alg{o[0], o[1]}, Alg{o[2], o[3]});
                out[k++] = o[4] - p.c0 * consts[0];
                out[k++] = o[5] - p.c1 * consts[0];
            }
            break;
        case GATE_REDUCING:             // output 0..2, alpha 2..4, old_acc 4..6, coefficients from 6 (base field), accumulators after
        case GATE_REDUCING_EXT: {       // the same with extension coefficients (two wires each)
            const bool ext = g.type == GATE_REDUCING_EXT;
            const u64 n = g.param0, accs = 6 + (ext ? 2 * n : n);
            const Alg alpha = {w[2], w[3]};
            Alg acc = {w[4], w[5]};
            for (u64 i = 0; i < n; i++) {
                Alg t = alg_mul(acc, alpha);
                const Alg next = i == n - 1 ? Alg{w[0], w[1]} : Alg{w[accs + 2 * i], w[accs + 2 * i + 1]};
                if (ext) { t.c0 = t.c0 + w[6 + 2 * i]; t.c1 = t.c1 + w[7 + 2 * i]; } else t.c0 = t.c0 + w[6 + i];
                out[k++] = t.c0 - next.c0; out[k++] = t.c1 - next.c1;
                acc = next;
            }
            break;
        }
        case GATE_RANDOM_ACCESS: {      // per copy: access_index, claimed_element, 2^bits items; the bit wires follow the routed ones
            const u64 bits = g.param0, copies = g.param1, extra = g.param2, vec = 1ull << bits;
            const u64 routed = (2 + vec) * copies + extra;
            std::vector<X> items(vec);
            for (u64 c = 0; c < copies; c++) {
                const X *cw = w + (2 + vec) * c, *bw = w + routed + c * bits;
                for (u64 i = 0; i < vec; i++) items[i] = cw[2 + i];
                for (u64 i = 0; i < bits; i++) out[k++] = bw[i] * (bw[i] - E(1));
                X idx = E(0);
                for (u64 i = bits; i-- > 0;) idx = (idx + idx) + bw[i];
                out[k++] = idx - cw[0];
                u64 len = vec;
                for (u64 b = 0; b < bits; b++) {
                    for (u64 i = 0; i < len / 2; i++) items[i] = items[2 * i] + bw[b] * (items[2 * i + 1] - items[2 * i]);
                    len >>= 1;
                }
                out[k++] = items[0] - cw[1];
            }
            for (u64 i = 0; i < extra; i++) out[k++] = consts[i] - w[(2 + vec) * copies + i];
            break;
        }
        case GATE_EXPONENTIATION: {     // base 0, power bits 1..1+n (little endian), output 1+n, intermediate values after
            const u64 n = g.param0;
            for (u64 i = 0; i < n; i++) {
                const X prev = i == 0 ? E(1) : w[2 + n + i - 1] * w[2 + n + i - 1];
                const X bit = w[1 + (n - 1 - i)];
                out[k++] = prev * (bit * w[0] + (E(1) - bit)) - w[2 + n + i];
            }
            out[k++] = w[1 + n] - w[2 + n + n - 1];
            break;
        }
        case GATE_POSEIDON_MDS: {       // 12 algebra elements in (wires 0..24), 12 out (24..48): out - MDS * in
            static const u64 CIRC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
            for (int r = 0; r < 12; r++)
                for (int comp = 0; comp < 2; comp++) {
                    X s = r == 0 ? scale(w[comp], 8) : E(0);
                    for (int i = 0; i < 12; i++) s = s + scale(w[2 * ((i + r) % 12) + comp], CIRC[i]);
                    out[k++] = w[24 + 2 * r + comp] - s;
                }
            break;
        }
        case GATE_COSET_INTERPOLATION: {   // shift, 2^bits values (algebra), evaluation point, value, intermediates, shifted point
            const u64 bits = g.param0, degree = g.param1, np = 1ull << bits, ni = (np - 2) / (degree - 1);
            const u64 s_ep = 1 + 2 * np, s_ev = s_ep + 2, s_int = s_ev + 2;
            // barycentric weights of the subgroup of order np: 1 / prod_{j != i} (x_i - x_j) = x_i / np
            std::vector<u64> dom(np), wt(np);
            { const u64 om = gl::root_of_unity((unsigned)bits), ninv = gl::inv(np); u64 x = 1; for (u64 i = 0; i < np; i++) { dom[i] = x; wt[i] = gl::mul(x, ninv); x = gl::mul(x, om); } }
            const X shift = w[0];
            const Alg ep = {w[s_ep], w[s_ep + 1]}, sp = {w[s_int + 4 * ni], w[s_int + 4 * ni + 1]};
            out[k++] = ep.c0 - sp.c0 * shift; out[k++] = ep.c1 - sp.c1 * shift;
            Alg ev = {E(0), E(0)}, pr = {E(1), E(0)};
            u64 lo = 0, hi = degree;
            for (u64 c = 0; c <= ni; c++) {
                for (u64 q = lo; q < hi; q++) {      // partial_interpolate_ext_algebra
                    Alg term = sp;
                    term.c0 = term.c0 - E(dom[q]);
                    const Alg t = alg_mul(Alg{w[1 + 2 * q], w[2 + 2 * q]}, pr);
                    ev = alg_mul(ev, term);
                    ev.c0 = ev.c0 + scale(t.c0, wt[q]); ev.c1 = ev.c1 + scale(t.c1, wt[q]);
                    pr = alg_mul(pr, term);
                }
                if (c == ni) break;
                const Alg ie = {w[s_int + 2 * c], w[s_int + 2 * c + 1]}, ip = {w[s_int + 2 * (ni + c)], w[s_int + 2 * (ni + c) + 1]};
                out[k++] = ie.c0 - ev.c0; out[k++] = ie.c1 - ev.c1;
                out[k++] = ip.c0 - pr.c0; out[k++] = ip.c1 - pr.c1;
                ev = ie; pr = ip;
                lo = 1 + (degree - 1) * (c + 1); hi = std::min<u64>(lo + degree - 1, np);
            }
            out[k++] = w[s_ev] - ev.c0; out[k++] = w[s_ev + 1] - ev.c1;
            break;
        }
        default: break;
    }
    return k;
}


// eval_vanishing_poly at zeta: the Z(1) = 1 terms, the partial-product checks and the filtered gate constraints, reduced with
// each alpha (Horner from the last term); out[k] for challenge k. l0 = L_0(zeta), the caller's (it needs a division).
// betas / gammas / alphas: num_challenges elements each, lifted to X by the caller. Returns "" or what is inconsistent in the pack.
template <class X>
std::string vanishing_at_zeta(const CircuitPack &c, const X &zeta, const X &l0, const X *o_cs, const X *o_w, const X *o_zs, const X *o_zn, const X *o_pp,
                              const X *betas, const X *gammas, const X *alphas, const X pih[4], std::vector<X> &out) {
    const size_t R = c.num_routed_wires, nch = c.num_challenges, npp = c.num_partial_products, nchunks = npp + 1, chunk = c.quotient_degree_factor;
    const size_t sig0 = c.num_selectors + c.num_constants;
    const X one = konst<X>(1);
    std::vector<X> terms;
    terms.reserve(nch + nch * nchunks + c.num_gate_constraints);
    for (size_t k = 0; k < nch; k++) terms.push_back(l0 * (o_zs[k] - one));
    for (size_t k = 0; k < nch; k++) {
        const X zeta_beta = zeta * betas[k];
        for (size_t cc = 0; cc < nchunks; cc++) {
            const X prev = cc == 0 ? o_zs[k] : o_pp[k * npp + cc - 1];
            const X next = cc == nchunks - 1 ? o_zn[k] : o_pp[k * npp + cc];
            X pn = prev, pd = next;
            for (size_t j = cc * chunk; j < (cc + 1) * chunk && j < R; j++) {
                pn = pn * (sadd(zeta_beta, c.k_is[j], o_w[j]) + gammas[k]);
                pd = pd * (madd(o_cs[sig0 + j], betas[k], o_w[j]) + gammas[k]);
            }
            terms.push_back(pn - pd);
        }
    }
    std::vector<X> gate_terms(c.num_gate_constraints, konst<X>(0)), cst;
    const X *consts = o_cs + c.num_selectors;
    for (size_t gi = 0; gi < c.gates.size(); gi++) {
        const GateInfo &g = c.gates[gi];
        if (g.num_constraints == 0) continue;
        const X s = o_cs[g.selector_index];
        X f = one;            // compute_filter: prod_{j in group, j != gate} (j - s), times (UNUSED - s) with several selectors
        for (u64 j = g.group_start; j < g.group_end; j++) if (j != gi) f = f * (konst<X>(j) - s);
        if (c.num_selectors > 1) f = f * (konst<X>(0xFFFFFFFFull) - s);
        const size_t cnt = gate_constraints(g, c.p2_layout, consts, o_w, pih, cst);
        if (cnt != g.num_constraints || cnt > gate_terms.size())
            return "gate " + std::to_string(gi) + ": the pack declares " + std::to_string(g.num_constraints) + " constraints, the gate has " + std::to_string(cnt);
        for (size_t i = 0; i < cnt; i++) gate_terms[i] = madd(f, cst[i], gate_terms[i]);
    }
    terms.insert(terms.end(), gate_terms.begin(), gate_terms.end());
    out.clear();
    for (size_t k = 0; k < nch; k++) {
        X acc = konst<X>(0);
        for (size_t j = terms.size(); j-- > 0;) acc = madd(acc, alphas[k], terms[j]);
        out.push_back(acc);
    }
    return "";
}

}  // namespace vmath
