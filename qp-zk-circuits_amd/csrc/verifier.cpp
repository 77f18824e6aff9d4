// verifier.cpp — host-side proof verification over a circuit pack (include/qpgpu_verify.h): qp-plonky2's
// `VerifierCircuitData::verify` as the reference applies it at every hand-over of a proof (leaf proofs entering a private
// batch, private-batch proofs entering a public batch, self-verification after proving). Pure host code: extension-field
// arithmetic from gl64.hpp, the context-independent hasher, the pack parser. The gate constraints are the verifier-side
// (extension field, one point) counterparts of the prover-side kernels in prover_kernels.hip, written separately from them.
#include "../../include/qpgpu.h"
#include "../../include/qpgpu_verify.h"
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <atomic>
#include <thread>
#include <string>
#include <vector>
#include "circuit.hpp"
#include "gl64.hpp"
#include "poseidon.hpp"

using gl::e2;
using gl::u64;

namespace {

int fail(char *err, int code, const char *fmt, ...) {
    if (err) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(err, QPGPU_VERIFY_ERR_CAP, fmt, ap);
        va_end(ap);
    }
    return code;
}

// ---- extension-field shorthands (F[x]/(x^2 - 7)) ----
inline e2 E(u64 a) { return gl::e2_from(a); }
inline e2 operator+(e2 x, e2 y) { return gl::e2_add(x, y); }
inline e2 operator-(e2 x, e2 y) { return gl::e2_sub(x, y); }
inline e2 operator*(e2 x, e2 y) { return gl::e2_mul(x, y); }
inline e2 scale(e2 x, u64 s) { return gl::e2_scale(x, s); }
inline bool same(e2 x, e2 y) { x = gl::e2_canon(x); y = gl::e2_canon(y); return x.a == y.a && x.b == y.b; }
// the extension ALGEBRA over the extension (wire pairs of the *Extension gates at zeta): c0 + c1 X, X^2 = 7, coefficients in e2
struct Alg { e2 c0, c1; };
inline Alg alg_mul(Alg a, Alg b) { return {a.c0 * b.c0 + scale(a.c1 * b.c1, 7), a.c0 * b.c1 + a.c1 * b.c0}; }

// ---- hashing under the proof system's permutation ----
struct Hash {
    const hasher::Config *h;
    void no_pad(const u64 *in, size_t n, u64 out[4]) const {       // hash_n_to_hash_no_pad: overwrite absorption, rate 8
        u64 st[12] = {0};
        for (size_t i = 0; i < n; i += 8) {
            const size_t len = std::min<size_t>(8, n - i);
            for (size_t k = 0; k < len; k++) st[k] = in[i + k];
            h->permute(st);
        }
        std::memcpy(out, st, 32);
    }
    void leaf(const u64 *row, size_t width, u64 out[4]) const {     // hash_or_noop
        if (width <= 4) { for (size_t i = 0; i < 4; i++) out[i] = i < width ? row[i] : 0; return; }
        no_pad(row, width, out);
    }
    void two_to_one(const u64 *l, const u64 *r, u64 out[4]) const {
        u64 st[12] = {l[0], l[1], l[2], l[3], r[0], r[1], r[2], r[3], 0, 0, 0, 0};
        h->permute(st);
        std::memcpy(out, st, 32);
    }
};

struct Transcript {     // plonky2::iop::challenger::Challenger
    const hasher::Config *h;
    u64 state[12] = {0};
    u64 in[8]; int n_in = 0;
    u64 out[8]; int n_out = 0;
    void duplex() {
        for (int i = 0; i < n_in; i++) state[i] = in[i];
        n_in = 0;
        h->permute(state);
        std::memcpy(out, state, sizeof out);
        n_out = 8;
    }
    void observe(const u64 *x, size_t n) { for (size_t i = 0; i < n; i++) { n_out = 0; in[n_in++] = x[i]; if (n_in == 8) duplex(); } }
    void observe(const std::vector<e2> &v) { for (const e2 &x : v) { u64 t[2] = {x.a, x.b}; observe(t, 2); } }
    u64 get() { if (n_in > 0 || n_out == 0) duplex(); return out[--n_out]; }
    e2 get_ext() { const u64 a = get(), b = get(); return gl::e2_make(a, b); }
};

inline uint32_t bitrev(uint32_t x, unsigned bits) {
    uint32_t r = 0;
    for (unsigned i = 0; i < bits; i++) r |= ((x >> i) & 1u) << (bits - 1 - i);
    return r;
}

// in-place radix-2 transform of 2^log_n values by powers of `root` (natural order in and out)
void ntt(std::vector<u64> &a, unsigned log_n, u64 root) {
    const size_t n = (size_t)1 << log_n;
    for (size_t i = 0; i < n; i++) { const size_t j = bitrev((uint32_t)i, log_n); if (j > i) std::swap(a[i], a[j]); }
    for (unsigned s = 1; s <= log_n; s++) {
        const size_t m = (size_t)1 << s, half = m >> 1;
        const u64 wm = gl::pow(root, n >> s);
        for (size_t k = 0; k < n; k += m) {
            u64 w = 1;
            for (size_t j = 0; j < half; j++) {
                const u64 t = gl::mul(w, a[k + j + half]), u = a[k + j];
                a[k + j] = gl::add(u, t); a[k + j + half] = gl::sub(u, t);
                w = gl::mul(w, wm);
            }
        }
    }
}

// ---- gate constraints at one point of the extension field ----
e2 sbox7(e2 x) { const e2 x2 = x * x, x4 = x2 * x2; return (x * x2) * x4; }
void mds_ext(e2 (&s)[12]) {   // the MDS matrix has base-field entries: it acts on the two coordinates separately
    u64 a[12], b[12];
    for (int i = 0; i < 12; i++) { a[i] = gl::canon(s[i].a); b[i] = gl::canon(s[i].b); }
    poseidon::mds_layer(a); poseidon::mds_layer(b);
    for (int i = 0; i < 12; i++) s[i] = gl::e2_make(a[i], b[i]);
}

// PoseidonGate (plonky2::gates::poseidon): wires 0..11 input, 12..23 output, 24 swap, 25..28 delta, 29..64 S-box inputs of
// full rounds 1..3, 65..86 of the 22 partial rounds, 87..134 of the last four full rounds; 123 constraints.
void poseidon_gate(const e2 *w, e2 *out) {
    const u64 *rc = poseidon::host_round_constants(), *fp = poseidon::host_fast_partial();
    size_t k = 0;
    const e2 swap = w[24];
    e2 st[12];
    out[k++] = swap * (swap - E(1));
    for (int i = 0; i < 4; i++) out[k++] = swap * (w[i + 4] - w[i]) - w[25 + i];
    for (int i = 0; i < 4; i++) { st[i] = w[i] + w[25 + i]; st[i + 4] = w[i + 4] - w[25 + i]; }
    for (int i = 8; i < 12; i++) st[i] = w[i];
    int r_idx = 0;
    for (int r = 0; r < 4; r++, r_idx++) {
        for (int i = 0; i < 12; i++) st[i] = st[i] + E(rc[r_idx * 12 + i]);
        if (r) for (int i = 0; i < 12; i++) { const e2 in = w[29 + 12 * (r - 1) + i]; out[k++] = st[i] - in; st[i] = in; }
        for (int i = 0; i < 12; i++) st[i] = sbox7(st[i]);
        mds_ext(st);
    }
    for (int i = 0; i < 12; i++) st[i] = st[i] + E(fp[poseidon::FP_FIRST + i]);                       // partial_first_constant_layer
    {
        e2 t[11];
        for (int c = 0; c < 11; c++) { e2 acc = E(0); for (int r = 0; r < 11; r++) acc = acc + scale(st[1 + r], fp[poseidon::FP_INIT + c * 11 + r]); t[c] = acc; }
        for (int c = 0; c < 11; c++) st[1 + c] = t[c];                                                // mds_partial_layer_init
    }
    for (int r = 0; r < 22; r++) {
        const e2 in = w[65 + r];
        out[k++] = st[0] - in;
        const e2 s0 = sbox7(in) + E(fp[poseidon::FP_RC + r]);
        e2 d = scale(s0, poseidon::MDS_00);
        for (int i = 0; i < 11; i++) d = d + scale(st[1 + i], fp[poseidon::FP_WHATS + r * 11 + i]);
        for (int i = 0; i < 11; i++) st[1 + i] = st[1 + i] + scale(s0, fp[poseidon::FP_VS + r * 11 + i]);
        st[0] = d;                                                                                    // mds_partial_layer_fast
    }
    r_idx += 22;
    for (int r = 0; r < 4; r++, r_idx++) {
        for (int i = 0; i < 12; i++) st[i] = st[i] + E(rc[r_idx * 12 + i]);
        for (int i = 0; i < 12; i++) { const e2 in = w[87 + 12 * r + i]; out[k++] = st[i] - in; st[i] = in; }
        for (int i = 0; i < 12; i++) st[i] = sbox7(st[i]);
        mds_ext(st);
    }
    for (int i = 0; i < 12; i++) out[k++] = st[i] - w[12 + i];
}

// The qp fork's Poseidon2 gate (type 14) at one extension point, verifier side: wires as the pack's layout table places them
// (circuit.hpp P2GateLayout; default = upstream PoseidonGate's layout carried over, LAYOUT UNPINNED), permutation =
// qp-poseidon-core's Poseidon2 (pinned by the reference's known-answer vectors). The linear layers have base-field entries, so
// they act on an extension element coefficient-wise (scale); only the S-boxes multiply extension elements.
void p2_external(e2 (&s)[12], const poseidon2::Params &P) {
    e2 t[12];
    for (int b = 0; b < 3; b++)
        for (int i = 0; i < 4; i++) {
            e2 acc = E(0);
            for (int j = 0; j < 4; j++) acc = acc + scale(s[4 * b + j], P.m4[4 * i + j]);
            t[4 * b + i] = acc;
        }
    for (int i = 0; i < 4; i++) {
        const e2 colsum = t[i] + t[4 + i] + t[8 + i];
        for (int b = 0; b < 3; b++) s[4 * b + i] = t[4 * b + i] + colsum;
    }
}
void p2_internal(e2 (&s)[12], const poseidon2::Params &P) {
    e2 total = E(0);
    for (int i = 0; i < 12; i++) total = total + s[i];
    for (int i = 0; i < 12; i++) s[i] = scale(s[i], P.diag_m1[i]) + total;
}
size_t poseidon2_gate(const P2GateLayout &lay, const e2 *w, e2 *out) {
    const poseidon2::Params &P = poseidon2::qp_params();
    size_t k = 0;
    e2 st[12];
    for (int i = 0; i < 12; i++) st[i] = w[lay.w_input + i];
    if (lay.has_swap()) {
        const e2 swap = w[lay.w_swap];
        out[k++] = swap * (swap - E(1));
        for (int i = 0; i < 4; i++) {
            const e2 delta = w[lay.w_delta + i];
            out[k++] = swap * (st[i + 4] - st[i]) - delta;
            st[i] = st[i] + delta; st[i + 4] = st[i + 4] - delta;
        }
    }
    p2_external(st, P);
    uint32_t rec = lay.w_full0;
    for (int r = 0; r < 8; r++) {
        if (r == 4) {   // the 22 internal rounds sit between the two halves
            for (int q = 0; q < 22; q++) {
                const e2 in = w[lay.w_partial + q];
                out[k++] = st[0] + E(P.rc_int[q]) - in;
                st[0] = sbox7(in);
                p2_internal(st, P);
            }
            rec = lay.w_full1;
        }
        for (int i = 0; i < 12; i++) st[i] = st[i] + E(P.rc_ext[r * 12 + i]);
        if (r != 0 || lay.first_round_wires) {
            for (int i = 0; i < 12; i++) { const e2 in = w[rec + i]; out[k++] = st[i] - in; st[i] = in; }
            rec += 12;
        }
        for (int i = 0; i < 12; i++) st[i] = sbox7(st[i]);
        p2_external(st, P);
    }
    for (int i = 0; i < 12; i++) out[k++] = st[i] - w[lay.w_output + i];
    return k;
}

// the unfiltered constraints of gate g in upstream order; returns how many were written
size_t gate_constraints(const GateInfo &g, const P2GateLayout &p2_layout, const e2 *consts, const e2 *w, const u64 pih[4], std::vector<e2> &out) {
    size_t k = 0;
    out.assign((size_t)g.num_constraints + 8, E(0));
    switch (g.type) {
        case GATE_NOOP: break;
        case GATE_CONSTANT:
            for (u64 i = 0; i < g.param0; i++) out[k++] = consts[i] - w[i];
            break;
        case GATE_PUBLIC_INPUT:
            for (int i = 0; i < 4; i++) out[k++] = w[i] - E(pih[i]);
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
            e2 s = E(0);
            for (u64 i = g.param0; i-- > 0;) s = (s + s) + w[1 + i];
            out[k++] = s - w[0];
            for (u64 i = 0; i < g.param0; i++) out[k++] = w[1 + i] * (w[1 + i] - E(1));
            break;
        }
        case GATE_ARITHMETIC_EXT:       // 8 wires per op: two multiplicands, addend, output, each an algebra element
            for (u64 i = 0; i < g.param0; i++) {
                const e2 *o = w + 8 * i;
                const Alg p = alg_mul({o[0], o[1]}, {o[2], o[3]});
                out[k++] = o[6] - (p.c0 * consts[0] + o[4] * consts[1]);
                out[k++] = o[7] - (p.c1 * consts[0] + o[5] * consts[1]);
            }
            break;
        case GATE_MUL_EXT:              // 6 wires per op
            for (u64 i = 0; i < g.param0; i++) {
                const e2 *o = w + 6 * i;
                const Alg p = alg_mul({o[0], o[1]}, {o[2], o[3]});
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
            std::vector<e2> items(vec);
            for (u64 c = 0; c < copies; c++) {
                const e2 *cw = w + (2 + vec) * c, *bw = w + routed + c * bits;
                for (u64 i = 0; i < vec; i++) items[i] = cw[2 + i];
                for (u64 i = 0; i < bits; i++) out[k++] = bw[i] * (bw[i] - E(1));
                e2 idx = E(0);
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
                const e2 prev = i == 0 ? E(1) : w[2 + n + i - 1] * w[2 + n + i - 1];
                const e2 bit = w[1 + (n - 1 - i)];
                out[k++] = prev * (bit * w[0] + (E(1) - bit)) - w[2 + n + i];
            }
            out[k++] = w[1 + n] - w[2 + n + n - 1];
            break;
        }
        case GATE_POSEIDON_MDS: {       // 12 algebra elements in (wires 0..24), 12 out (24..48): out - MDS * in
            static const u64 CIRC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
            for (int r = 0; r < 12; r++)
                for (int comp = 0; comp < 2; comp++) {
                    e2 s = r == 0 ? scale(w[comp], 8) : E(0);
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
            const e2 shift = w[0];
            const Alg ep = {w[s_ep], w[s_ep + 1]}, sp = {w[s_int + 4 * ni], w[s_int + 4 * ni + 1]};
            out[k++] = ep.c0 - sp.c0 * shift; out[k++] = ep.c1 - sp.c1 * shift;
            Alg ev = {E(0), E(0)}, pr = {E(1), E(0)};
            u64 lo = 0, hi = degree;
            for (u64 c = 0; c <= ni; c++) {
                for (u64 q = lo; q < hi; q++) {      // partial_interpolate_ext_algebra
                    Alg term = sp;
                    term.c0 = term.c0 - E(dom[q]);
                    const Alg t = alg_mul({w[1 + 2 * q], w[2 + 2 * q]}, pr);
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

}  // namespace

struct qpgpu_verifier {
    CircuitPack pack;
    hasher::Config hash;
    std::vector<u64> cs_cap;
    size_t proof_size = 0;
};

namespace {

size_t proof_size_of(const CircuitPack &p) {
    const size_t ncs = p.num_cs_cols(), nch = p.num_challenges, cap = ((size_t)1 << p.cap_height) * 32;
    const size_t openings = (ncs + p.num_wires + 2 * nch + nch * p.num_partial_products + nch * p.quotient_degree_factor) * 16;
    const size_t L = p.degree_bits + p.rate_bits, salt = p.zero_knowledge ? 4 : 0;
    const size_t widths[4] = {ncs, p.num_wires + salt, p.num_zs_pp_cols() + salt, p.num_quotient_cols() + salt};
    size_t q = 0, sz = 3 * cap + openings, lvl = L, fin = p.degree_bits;
    for (size_t w : widths) q += w * 8 + 1 + (L - p.cap_height) * 32;
    for (u64 ab : p.arity_bits) {
        sz += cap; lvl -= ab; fin -= ab;
        q += ((size_t)1 << ab) * 16 + 1 + (lvl - p.cap_height) * 32;
    }
    return sz + p.num_query_rounds * q + ((size_t)1 << fin) * 16 + 8 + p.num_public_inputs * 8;
}

// constants/sigmas cap from the pack: per column values -> coefficients -> coset LDE in leaf order; leaves hashed, tree to the cap
void host_cs_cap(const CircuitPack &p, const Hash &H, std::vector<u64> &cap) {
    const unsigned d = (unsigned)p.degree_bits, L = d + (unsigned)p.rate_bits;
    const size_t n = (size_t)1 << d, lde_n = (size_t)1 << L, ncs = p.num_cs_cols();
    std::vector<u64> lde(ncs * lde_n), col(n), ext(lde_n);
    const u64 w_inv = gl::inv(gl::root_of_unity(d)), n_inv = gl::inv(n), w_lde = gl::root_of_unity(L);
    for (size_t c = 0; c < ncs; c++) {
        std::copy(p.constants_sigmas.begin() + c * n, p.constants_sigmas.begin() + (c + 1) * n, col.begin());
        ntt(col, d, w_inv);
        u64 shift = 1;
        std::fill(ext.begin(), ext.end(), 0);
        for (size_t i = 0; i < n; i++) { ext[i] = gl::mul(gl::mul(col[i], n_inv), shift); shift = gl::mul(shift, gl::MULT_GEN); }
        ntt(ext, L, w_lde);
        for (size_t j = 0; j < lde_n; j++) lde[c * lde_n + j] = gl::canon(ext[bitrev((uint32_t)j, L)]);   // leaf j = point g w^rev(j)
    }
    std::vector<u64> level(lde_n * 4), row(ncs);
    for (size_t j = 0; j < lde_n; j++) {
        for (size_t c = 0; c < ncs; c++) row[c] = lde[c * lde_n + j];
        H.leaf(row.data(), ncs, &level[4 * j]);
    }
    size_t cnt = lde_n;
    while (cnt > ((size_t)1 << p.cap_height)) {
        for (size_t i = 0; i < cnt / 2; i++) { u64 o[4]; H.two_to_one(&level[8 * i], &level[8 * i + 4], o); std::memcpy(&level[4 * i], o, 32); }
        cnt >>= 1;
    }
    cap.assign(level.begin(), level.begin() + 4 * cnt);
}

struct Reader {
    const uint8_t *p; size_t len, pos = 0; bool bad = false, noncanonical = false;
    u64 word() {
        if (pos + 8 > len) { bad = true; return 0; }
        u64 v; std::memcpy(&v, p + pos, 8); pos += 8;
        if (v >= gl::P) noncanonical = true;      // Field::from_canonical_u64 on read: a proof carries canonical elements only
        return v;
    }
    uint8_t byte() { if (pos + 1 > len) { bad = true; return 0; } return p[pos++]; }
    void vec(u64 *out, size_t n) { for (size_t i = 0; i < n; i++) out[i] = word(); }
    e2 ext() { const u64 a = word(), b = word(); return gl::e2_make(a, b); }
    void exts(std::vector<e2> &v, size_t n) { v.resize(n); for (size_t i = 0; i < n; i++) v[i] = ext(); }
};

bool path_ok(const Hash &H, const u64 *leaf, size_t width, size_t index, const u64 *path, size_t plen, const u64 *cap, unsigned cap_h, size_t log_leaves) {
    if (plen != log_leaves - cap_h) return false;
    u64 cur[4], nxt[4];
    H.leaf(leaf, width, cur);
    for (size_t i = 0; i < plen; i++) {
        if (index & 1) H.two_to_one(path + 4 * i, cur, nxt); else H.two_to_one(cur, path + 4 * i, nxt);
        std::memcpy(cur, nxt, 32);
        index >>= 1;
    }
    return std::memcmp(cur, cap + 4 * index, 32) == 0;
}

}  // namespace

extern "C" {

int qpgpu_verifier_create(const uint64_t *pack_words, size_t n_words, const uint64_t *cs_cap, size_t cap_words, int hasher_kind,
                          const uint64_t *hasher_params, size_t n_params, qpgpu_verifier **out, char *err) {
    if (!pack_words || !out) return fail(err, QPGPU_EINVAL, "null argument");
    qpgpu_verifier *v = new (std::nothrow) qpgpu_verifier();
    if (!v) return fail(err, QPGPU_ENOMEM, "out of memory");
    const std::string why = v->pack.parse(pack_words, n_words);
    if (!why.empty()) { delete v; return fail(err, QPGPU_EINVAL, "circuit pack: %s", why.c_str()); }
    if (hasher_kind == hasher::POSEIDON) v->hash.kind = hasher::POSEIDON;
    else if (hasher_kind == hasher::POSEIDON2) {
        v->hash.kind = hasher::POSEIDON2;
        if (!hasher_params && n_params == 0) v->hash.p2 = poseidon2::qp_params();
        else if (hasher_params && n_params == (size_t)poseidon2::PARAM_WORDS) {
            for (size_t i = 0; i < n_params; i++) if (hasher_params[i] >= gl::P) { delete v; return fail(err, QPGPU_EINVAL, "Poseidon2 parameter %zu is not canonical", i); }
            std::memcpy(v->hash.p2.rc_ext, hasher_params, 96 * 8);
            std::memcpy(v->hash.p2.rc_int, hasher_params + 96, 22 * 8);
            std::memcpy(v->hash.p2.diag_m1, hasher_params + 118, 12 * 8);
            std::memcpy(v->hash.p2.m4, hasher_params + 130, 16 * 8);
        } else { delete v; return fail(err, QPGPU_EINVAL, "bad Poseidon2 parameter block"); }
    } else { delete v; return fail(err, QPGPU_EINVAL, "unknown hasher kind"); }
    const CircuitPack &p = v->pack;
    if (p.num_challenges > 4 || p.arity_bits.size() > 16) { delete v; return fail(err, QPGPU_EINVAL, "circuit outside the supported range"); }
    for (u64 ab : p.arity_bits) if (ab == 0 || ab > 5) { delete v; return fail(err, QPGPU_EINVAL, "FRI arity outside 2..32"); }
    {   // every FRI round's tree must still be at least as tall as the cap (the proof layout subtracts the two)
        u64 lvl = p.degree_bits + p.rate_bits;
        for (u64 ab : p.arity_bits) {
            if (ab > lvl || lvl - ab < p.cap_height) { delete v; return fail(err, QPGPU_EINVAL, "FRI reduction schedule is inconsistent with degree_bits / cap_height"); }
            lvl -= ab;
        }
    }
    const size_t want = ((size_t)1 << p.cap_height) * 4;
    if (cs_cap) {
        if (cap_words != want) { delete v; return fail(err, QPGPU_EINVAL, "constants/sigmas cap has %zu words, the circuit's cap height needs %zu", cap_words, want); }
        v->cs_cap.assign(cs_cap, cs_cap + want);
    } else {
        if (p.degree_bits + p.rate_bits > 20) { delete v; return fail(err, QPGPU_EINVAL, "constants/sigmas cap not given and the circuit is too large to rebuild it on the host"); }
        const Hash H{&v->hash};
        host_cs_cap(p, H, v->cs_cap);
    }
    v->proof_size = proof_size_of(p);
    *out = v;
    return QPGPU_OK;
}

void qpgpu_verifier_free(qpgpu_verifier *v) { delete v; }
size_t qpgpu_verifier_proof_size(const qpgpu_verifier *v) { return v ? v->proof_size : 0; }
int qpgpu_verifier_constants_sigmas_cap(const qpgpu_verifier *v, uint64_t *out, size_t out_words) {
    if (!v || !out || out_words < v->cs_cap.size()) return QPGPU_EINVAL;
    std::memcpy(out, v->cs_cap.data(), v->cs_cap.size() * 8);
    return QPGPU_OK;
}

static int verify_impl(const qpgpu_verifier *v, const uint8_t *proof, size_t len, char *err, uint64_t *indices_out);
int qpgpu_verifier_verify(const qpgpu_verifier *v, const uint8_t *proof, size_t len, char *err) { return verify_impl(v, proof, len, err, nullptr); }
// The FRI query indices of a proof: the transcript replayed up to the proof of work, then num_query_rounds challenges reduced
// mod the LDE size. Everything the transcript absorbs is checked on the way (sizes, canonical elements, quotient identity,
// proof of work); the query rounds themselves are NOT looked at, so the indices of a proof whose opened rows or paths were
// tampered with are still returned (they are what a recursive verifier circuit derives in-circuit, wormhole/aggregator/src/
// common/recursive.rs:91-97).
int qpgpu_verifier_query_indices(const qpgpu_verifier *v, const uint8_t *proof, size_t len, uint64_t *out, size_t cap, char *err) {
    if (!v || !out || cap < v->pack.num_query_rounds) return fail(err, QPGPU_EINVAL, "query_indices: null argument or room for fewer than num_query_rounds indices");
    return verify_impl(v, proof, len, err, out);
}
static int verify_impl(const qpgpu_verifier *v, const uint8_t *proof, size_t len, char *err, uint64_t *indices_out) {
    if (!v || !proof) return fail(err, QPGPU_EINVAL, "null argument");
    const CircuitPack &c = v->pack;
    const Hash H{&v->hash};
    if (len != v->proof_size) return fail(err, QPGPU_EVERIFY, "proof has %zu bytes, this circuit's proofs have %zu", len, v->proof_size);
    const unsigned d = (unsigned)c.degree_bits, rb = (unsigned)c.rate_bits, cap_h = (unsigned)c.cap_height, L = d + rb;
    const size_t n = (size_t)1 << d, lde_n = n << rb, R = c.num_routed_wires, NW = c.num_wires, nch = c.num_challenges;
    const size_t npp = c.num_partial_products, nchunks = npp + 1, chunk = c.quotient_degree_factor, ncs = c.num_cs_cols();
    const size_t sig0 = c.num_selectors + c.num_constants, cap_words = ((size_t)1 << cap_h) * 4, nq = nch * c.quotient_degree_factor;
    const size_t n_rounds = c.arity_bits.size();

    Reader b{proof, len};
    std::vector<u64> wires_cap(cap_words), zs_cap(cap_words), q_cap(cap_words);
    b.vec(wires_cap.data(), cap_words); b.vec(zs_cap.data(), cap_words); b.vec(q_cap.data(), cap_words);
    std::vector<e2> o_cs, o_w, o_zs, o_zn, o_pp, o_q;
    b.exts(o_cs, ncs); b.exts(o_w, NW); b.exts(o_zs, nch); b.exts(o_zn, nch); b.exts(o_pp, nch * npp); b.exts(o_q, nq);
    std::vector<u64> fri_caps(cap_words * n_rounds);
    b.vec(fri_caps.data(), cap_words * n_rounds);
    const size_t queries_pos = b.pos;
    const size_t salt = c.zero_knowledge ? 4 : 0;
    const size_t widths[4] = {ncs, NW + salt, nch * (1 + npp) + salt, nq + salt};
    const size_t polys[4] = {ncs, NW, nch * (1 + npp), nq};
    {
        size_t q = 0, lvl = L;
        for (size_t w : widths) q += w * 8 + 1 + (L - cap_h) * 32;
        for (u64 ab : c.arity_bits) { lvl -= ab; q += ((size_t)1 << ab) * 16 + 1 + (lvl - cap_h) * 32; }
        b.pos += q * c.num_query_rounds;
    }
    size_t fin_bits = d;
    for (u64 ab : c.arity_bits) fin_bits -= ab;
    std::vector<e2> final_poly;
    b.exts(final_poly, (size_t)1 << fin_bits);
    u64 pow_witness = b.word();
    std::vector<u64> pis(c.num_public_inputs + 1);
    b.vec(pis.data(), c.num_public_inputs);
    if (b.bad || b.pos != len) return fail(err, QPGPU_EVERIFY, "proof layout does not match the circuit");
    if (b.noncanonical) return fail(err, QPGPU_EVERIFY, "proof holds a non-canonical field element");

    // ---- challenges: the prover's transcript, replayed ----
    u64 pih[4];
    H.no_pad(pis.data(), c.num_public_inputs, pih);
    Transcript ch{&v->hash};
    ch.observe(c.circuit_digest, 4);
    ch.observe(pih, 4);
    ch.observe(wires_cap.data(), cap_words);
    u64 betas[4], gammas[4], alphas[4];
    for (size_t k = 0; k < nch; k++) betas[k] = ch.get();
    for (size_t k = 0; k < nch; k++) gammas[k] = ch.get();
    ch.observe(zs_cap.data(), cap_words);
    for (size_t k = 0; k < nch; k++) alphas[k] = ch.get();
    ch.observe(q_cap.data(), cap_words);
    const e2 zeta = ch.get_ext();
    ch.observe(o_cs); ch.observe(o_w); ch.observe(o_zs); ch.observe(o_pp); ch.observe(o_q); ch.observe(o_zn);
    const e2 fri_alpha = ch.get_ext();
    std::vector<e2> fri_betas(n_rounds);
    for (size_t r = 0; r < n_rounds; r++) { ch.observe(fri_caps.data() + r * cap_words, cap_words); fri_betas[r] = ch.get_ext(); }
    ch.observe(final_poly);
    ch.observe(&pow_witness, 1);
    const u64 pow_response = ch.get();
    if (c.proof_of_work_bits && (pow_response >> (64 - c.proof_of_work_bits)) != 0)
        return fail(err, QPGPU_EVERIFY, "proof-of-work response has fewer than %llu leading zero bits", (unsigned long long)c.proof_of_work_bits);

    // ---- vanishing polynomial at zeta against Z_H(zeta) * quotient(zeta) ----
    {
        e2 zeta_n = zeta;
        for (unsigned i = 0; i < d; i++) zeta_n = zeta_n * zeta_n;
        const e2 zh = zeta_n - E(1);
        const e2 l0 = zh * gl::e2_inv(scale(zeta - E(1), (u64)n));
        std::vector<e2> terms;
        terms.reserve(nch + nch * nchunks + c.num_gate_constraints);
        for (size_t k = 0; k < nch; k++) terms.push_back(l0 * (o_zs[k] - E(1)));
        for (size_t k = 0; k < nch; k++)
            for (size_t cc = 0; cc < nchunks; cc++) {
                const e2 prev = cc == 0 ? o_zs[k] : o_pp[k * npp + cc - 1];
                const e2 next = cc == nchunks - 1 ? o_zn[k] : o_pp[k * npp + cc];
                e2 pn = E(1), pd = E(1);
                for (size_t j = cc * chunk; j < (cc + 1) * chunk && j < R; j++) {
                    pn = pn * (o_w[j] + scale(zeta, gl::mul(betas[k], c.k_is[j])) + E(gammas[k]));
                    pd = pd * (o_w[j] + scale(o_cs[sig0 + j], betas[k]) + E(gammas[k]));
                }
                terms.push_back(prev * pn - next * pd);
            }
        std::vector<e2> gate_terms(c.num_gate_constraints, E(0)), cst;
        const e2 *consts = o_cs.data() + c.num_selectors;
        for (size_t gi = 0; gi < c.gates.size(); gi++) {
            const GateInfo &g = c.gates[gi];
            if (g.num_constraints == 0) continue;
            e2 f = E(1);          // compute_filter: prod_{j in group, j != gate} (j - s), times (UNUSED - s) with several selectors
            const e2 s = o_cs[g.selector_index];
            for (u64 j = g.group_start; j < g.group_end; j++) if (j != gi) f = f * (E(j) - s);
            if (c.num_selectors > 1) f = f * (E(0xFFFFFFFFull) - s);
            const size_t cnt = gate_constraints(g, c.p2_layout, consts, o_w.data(), pih, cst);
            if (cnt != g.num_constraints || cnt > gate_terms.size())
                return fail(err, QPGPU_EVERIFY, "gate %zu: the pack declares %llu constraints, the gate has %zu", gi, (unsigned long long)g.num_constraints, cnt);
            for (size_t i = 0; i < cnt; i++) gate_terms[i] = gate_terms[i] + f * cst[i];
        }
        terms.insert(terms.end(), gate_terms.begin(), gate_terms.end());
        for (size_t k = 0; k < nch; k++) {
            e2 acc = E(0);
            for (size_t j = terms.size(); j-- > 0;) acc = scale(acc, alphas[k]) + terms[j];
            e2 qv = E(0);
            for (size_t j = c.quotient_degree_factor; j-- > 0;) qv = qv * zeta_n + o_q[k * c.quotient_degree_factor + j];
            if (!same(acc, zh * qv)) return fail(err, QPGPU_EVERIFY, "quotient identity fails at zeta (challenge %zu): the openings do not satisfy the circuit", k);
        }
    }

    // ---- FRI ----
    e2 red0 = E(0), red1 = E(0);     // reduced openings: sum_j opening_j alpha^j, batch 0 in oracle order, batch 1 = Zs at g zeta
    {
        const std::vector<e2> *parts[5] = {&o_cs, &o_w, &o_zs, &o_pp, &o_q};
        for (int p = 5; p-- > 0;) for (size_t j = parts[p]->size(); j-- > 0;) red0 = red0 * fri_alpha + (*parts[p])[j];
        for (size_t j = nch; j-- > 0;) red1 = red1 * fri_alpha + o_zn[j];
    }
    const e2 g_zeta = scale(zeta, gl::root_of_unity(d));
    const e2 alpha_nch = gl::e2_pow(fri_alpha, nch);
    const u64 *caps0[4] = {v->cs_cap.data(), wires_cap.data(), zs_cap.data(), q_cap.data()};
    Reader q{proof, len, queries_pos};
    std::vector<u64> row(ncs + NW + nch * (1 + npp) + nq + 16), path(64 * 4), ev(64);
    const u64 w_lde = gl::root_of_unity(L);
    std::vector<size_t> x_indices(c.num_query_rounds);      // the loop below does not touch the transcript: draw them all first
    for (size_t qi = 0; qi < c.num_query_rounds; qi++) x_indices[qi] = (size_t)(ch.get() % lde_n);
    if (indices_out) { for (size_t qi = 0; qi < c.num_query_rounds; qi++) indices_out[qi] = x_indices[qi]; return QPGPU_OK; }
    for (size_t qi = 0; qi < c.num_query_rounds; qi++) {
        size_t x_index = x_indices[qi];
        const u64 *rows[4];
        size_t off = 0;
        for (int o = 0; o < 4; o++) {
            u64 *r = row.data() + off;
            rows[o] = r;
            q.vec(r, widths[o]); off += widths[o];
            const size_t plen = q.byte();
            if (plen > 60) return fail(err, QPGPU_EVERIFY, "query %zu: Merkle path length of oracle %d out of range", qi, o);
            q.vec(path.data(), plen * 4);
            if (q.bad || !path_ok(H, r, widths[o], x_index, path.data(), plen, caps0[o], cap_h, L))
                return fail(err, QPGPU_EVERIFY, "query %zu: Merkle path of initial oracle %d does not lead to its cap", qi, o);
        }
        u64 subgroup_x = gl::mul(gl::MULT_GEN, gl::pow(w_lde, bitrev((uint32_t)x_index, L)));
        e2 e0 = E(0), e1 = E(0);     // fri_combine_initial: salts are not opened
        for (int o = 3; o >= 0; o--) for (size_t j = polys[o]; j-- > 0;) e0 = e0 * fri_alpha + E(rows[o][j]);
        for (size_t j = nch; j-- > 0;) e1 = e1 * fri_alpha + E(rows[2][j]);
        const e2 sx = E(subgroup_x);
        e2 sum = (e0 - red0) * gl::e2_inv(sx - zeta);
        sum = sum * alpha_nch + (e1 - red1) * gl::e2_inv(sx - g_zeta);
        e2 old_eval = sum;
        size_t lvl = L;
        for (size_t r = 0; r < n_rounds; r++) {
            const unsigned ab = (unsigned)c.arity_bits[r];
            const size_t arity = (size_t)1 << ab;
            q.vec(ev.data(), 2 * arity);
            const size_t plen = q.byte();
            if (plen > 60) return fail(err, QPGPU_EVERIFY, "query %zu: Merkle path length of FRI round %zu out of range", qi, r);
            q.vec(path.data(), plen * 4);
            const size_t coset_index = x_index >> ab, within = x_index & (arity - 1);
            if (!same(gl::e2_make(ev[2 * within], ev[2 * within + 1]), old_eval))
                return fail(err, QPGPU_EVERIFY, "query %zu: FRI round %zu does not continue the previous evaluation", qi, r);
            lvl -= ab;
            if (q.bad || !path_ok(H, ev.data(), 2 * arity, coset_index, path.data(), plen, fri_caps.data() + r * cap_words, cap_h, lvl))
                return fail(err, QPGPU_EVERIFY, "query %zu: Merkle path of FRI round %zu does not lead to its cap", qi, r);
            // compute_evaluation: interpolate the coset's 2^ab evaluations and evaluate at beta
            const u64 g = gl::root_of_unity(ab);
            const size_t rev_within = bitrev((uint32_t)within, ab);
            const u64 coset_start = gl::mul(subgroup_x, gl::pow(g, arity - rev_within));
            u64 px[32]; e2 py[32];
            for (size_t i = 0; i < arity; i++) {
                const size_t src = bitrev((uint32_t)i, ab);
                py[i] = gl::e2_make(ev[2 * src], ev[2 * src + 1]);
                px[i] = gl::mul(coset_start, gl::pow(g, i));
            }
            const e2 beta = fri_betas[r];
            e2 acc = E(0);
            for (size_t i = 0; i < arity; i++) {      // Lagrange form
                e2 num = E(1); u64 den = 1;
                for (size_t j = 0; j < arity; j++) if (j != i) { num = num * (beta - E(px[j])); den = gl::mul(den, gl::sub(px[i], px[j])); }
                acc = acc + py[i] * scale(num, gl::inv(den));
            }
            old_eval = acc;
            subgroup_x = gl::pow(subgroup_x, arity);
            x_index = coset_index;
        }
        e2 fe = E(0);
        const e2 sxe = E(subgroup_x);
        for (size_t i = final_poly.size(); i-- > 0;) fe = fe * sxe + final_poly[i];
        if (!same(fe, old_eval)) return fail(err, QPGPU_EVERIFY, "query %zu: the final polynomial does not match the last FRI round", qi);
    }
    if (q.bad) return fail(err, QPGPU_EVERIFY, "proof layout does not match the circuit");
    return QPGPU_OK;
}

int qpgpu_verifier_verify_many(const qpgpu_verifier *v, const uint8_t *const *proofs, const size_t *lens, size_t count, unsigned threads,
                               int *results, char *err) {
    if (!v || !proofs || !lens || !results) return fail(err, QPGPU_EINVAL, "null argument");
    if (threads == 0) threads = std::max(1u, std::thread::hardware_concurrency());
    threads = (unsigned)std::min<size_t>(threads, std::max<size_t>(count, 1));
    std::atomic<size_t> next{0};
    std::vector<std::string> reasons(count);
    auto work = [&] {
        char local[QPGPU_VERIFY_ERR_CAP];
        for (size_t i = next.fetch_add(1); i < count; i = next.fetch_add(1)) {
            local[0] = 0;
            results[i] = proofs[i] ? qpgpu_verifier_verify(v, proofs[i], lens[i], local) : QPGPU_EINVAL;
            if (results[i]) reasons[i] = local;
        }
    };
    std::vector<std::thread> pool;
    try {
        for (unsigned t = 1; t < threads; t++) pool.emplace_back(work);
    } catch (...) {}                 // no more threads to be had: the ones that started (and this one) drain the queue
    work();
    for (auto &t : pool) t.join();
    for (size_t i = 0; i < count; i++)
        if (results[i]) return fail(err, QPGPU_EVERIFY, "proof %zu: %.170s", i, reasons[i].c_str());
    return QPGPU_OK;
}

}  // extern "C"
