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
#include "verify_math.hpp"

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
using gl::scale;       // + - * scale sadd madd on e2: verify_math.hpp
inline bool same(e2 x, e2 y) { x = gl::e2_canon(x); y = gl::e2_canon(y); return x.a == y.a && x.b == y.b; }

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

// the gate constraints and the vanishing polynomial at one point: verify_math.hpp (shared with the in-circuit verifier)
using vmath::gate_constraints;

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
    const size_t n = (size_t)1 << d, lde_n = n << rb, NW = c.num_wires, nch = c.num_challenges;
    const size_t npp = c.num_partial_products, ncs = c.num_cs_cols();
    const size_t cap_words = ((size_t)1 << cap_h) * 4, nq = nch * c.quotient_degree_factor;
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
        e2 xb[4], xg[4], xa[4], xpih[4];
        for (size_t k = 0; k < nch; k++) { xb[k] = E(betas[k]); xg[k] = E(gammas[k]); xa[k] = E(alphas[k]); }
        for (int i = 0; i < 4; i++) xpih[i] = E(pih[i]);
        std::vector<e2> van;
        const std::string why = vmath::vanishing_at_zeta<e2>(c, zeta, l0, o_cs.data(), o_w.data(), o_zs.data(), o_zn.data(), o_pp.data(), xb, xg, xa, xpih, van);
        if (!why.empty()) return fail(err, QPGPU_EVERIFY, "%s", why.c_str());
        for (size_t k = 0; k < nch; k++) {
            e2 qv = E(0);
            for (size_t j = c.quotient_degree_factor; j-- > 0;) qv = qv * zeta_n + o_q[k * c.quotient_degree_factor + j];
            if (!same(van[k], zh * qv)) return fail(err, QPGPU_EVERIFY, "quotient identity fails at zeta (challenge %zu): the openings do not satisfy the circuit", k);
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
