// proof_targets.cpp — inner proofs as witness assignments of a recursive wrapper circuit (include/qpgpu_batch.h, "inner-proof
// targets"). Host only.
//
// The reference fills the private-batch (and public-batch) wrapper's PartialWitness with N complete inner proofs:
//   fill_private_batch_witness                 wormhole/aggregator/src/private_batch/prover/witness.rs:15-77
//   ensure_proof_shape_matches_targets         wormhole/aggregator/src/common/utils.rs:295-540
//   pw.set_proof_with_pis_target(proof_t, p)   qp-plonky2 1.5.5 iop::witness::WitnessWrite (un-vendored; restated from upstream)
// A proof target is the tree of virtual targets `add_virtual_proof_with_pis(common_data)` creates: one target per field
// element of a proof of the INNER circuit. Here that tree is flattened into "logical targets" 0 .. T-1 in the documented order
// of qpgpu_proof_target_count; the circuit-pack exporter records which wire cell each logical target became
// (integration/qpgpu_backend.rs), exactly as for the leaf circuit's targets (include/qpgpu_leaf.h), and
// qpgpu_leaf_map_targets / qpgpu_generate_witness_partial_dev take it from there.
#include "../../include/qpgpu_batch.h"
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include "circuit.hpp"

namespace {

constexpr uint64_t P = 0xFFFFFFFF00000001ull;

int fail(char *err, int code, const char *fmt, ...) {
    if (err) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(err, QPGPU_BATCH_ERR_CAP, fmt, ap);
        va_end(ap);
    }
    return code;
}

// vector lengths of a ProofWithPublicInputs (or of its target), in the order ensure_proof_shape_matches_targets visits them
struct Shape {
    uint32_t public_inputs = 0, wires_cap = 0, zs_pp_cap = 0, quotient_cap = 0;
    uint32_t openings[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};   // constants, plonk_sigmas, wires, plonk_zs, plonk_zs_next, partial_products, quotient_polys, lookup_zs, lookup_zs_next
    std::vector<uint32_t> commit_caps;                    // digests per FRI commit-phase cap
    struct Round { std::vector<uint32_t> evals, siblings, step_evals, step_siblings; };
    std::vector<Round> rounds;
    uint32_t final_poly = 0;

    std::vector<uint32_t> flatten() const {
        std::vector<uint32_t> f = {public_inputs, wires_cap, zs_pp_cap, quotient_cap};
        f.insert(f.end(), openings, openings + 9);
        f.push_back((uint32_t)commit_caps.size());
        f.insert(f.end(), commit_caps.begin(), commit_caps.end());
        f.push_back((uint32_t)rounds.size());
        for (const Round &r : rounds) {
            f.push_back((uint32_t)r.evals.size());
            for (size_t j = 0; j < r.evals.size(); j++) { f.push_back(r.evals[j]); f.push_back(r.siblings[j]); }
            f.push_back((uint32_t)r.step_evals.size());
            for (size_t j = 0; j < r.step_evals.size(); j++) { f.push_back(r.step_evals[j]); f.push_back(r.step_siblings[j]); }
        }
        f.push_back(final_poly);
        return f;
    }
    // the inverse; false when the words do not describe a shape
    bool unflatten(const uint32_t *w, size_t n) {
        size_t p = 0;
        auto get = [&](uint32_t &v) { if (p >= n) return false; v = w[p++]; return true; };
        if (!get(public_inputs) || !get(wires_cap) || !get(zs_pp_cap) || !get(quotient_cap)) return false;
        for (int i = 0; i < 9; i++) if (!get(openings[i])) return false;
        uint32_t k = 0;
        if (!get(k) || k > 64) return false;
        commit_caps.resize(k);
        for (uint32_t &c : commit_caps) if (!get(c)) return false;
        if (!get(k) || k > 4096) return false;
        rounds.assign(k, Round());
        for (Round &r : rounds) {
            uint32_t m = 0;
            if (!get(m) || m > 64) return false;
            r.evals.resize(m); r.siblings.resize(m);
            for (uint32_t j = 0; j < m; j++) if (!get(r.evals[j]) || !get(r.siblings[j])) return false;
            if (!get(m) || m > 64) return false;
            r.step_evals.resize(m); r.step_siblings.resize(m);
            for (uint32_t j = 0; j < m; j++) if (!get(r.step_evals[j]) || !get(r.step_siblings[j])) return false;
        }
        return get(final_poly) && p == n;
    }
};

// what add_virtual_proof_with_pis(common_data) allocates for a proof of the circuit `c`
Shape target_shape(const CircuitPack &c) {
    Shape s;
    const uint32_t cap = 1u << c.cap_height, nch = (uint32_t)c.num_challenges, salt = c.zero_knowledge ? 4 : 0;
    const uint32_t L = (uint32_t)(c.degree_bits + c.rate_bits);
    s.public_inputs = (uint32_t)c.num_public_inputs;
    s.wires_cap = s.zs_pp_cap = s.quotient_cap = cap;
    s.openings[0] = (uint32_t)(c.num_selectors + c.num_constants); s.openings[1] = (uint32_t)c.num_routed_wires; s.openings[2] = (uint32_t)c.num_wires;
    s.openings[3] = nch; s.openings[4] = nch; s.openings[5] = nch * (uint32_t)c.num_partial_products; s.openings[6] = (uint32_t)c.num_quotient_cols();
    s.openings[7] = s.openings[8] = 0;      // none of the reference's circuits registers a lookup table
    s.commit_caps.assign(c.arity_bits.size(), cap);
    Shape::Round r;
    const uint32_t widths[4] = {(uint32_t)c.num_cs_cols(), (uint32_t)c.num_wires + salt, (uint32_t)c.num_zs_pp_cols() + salt, (uint32_t)c.num_quotient_cols() + salt};
    for (uint32_t w : widths) { r.evals.push_back(w); r.siblings.push_back(L - (uint32_t)c.cap_height); }
    uint32_t lvl = L, fin = (uint32_t)c.degree_bits;
    for (uint64_t ab : c.arity_bits) {
        lvl -= (uint32_t)ab; fin -= (uint32_t)ab;
        r.step_evals.push_back(1u << ab);                   // uncompressed query steps carry every evaluation of the coset
        r.step_siblings.push_back(lvl - (uint32_t)c.cap_height);
    }
    s.rounds.assign(c.num_query_rounds, r);
    s.final_poly = 1u << fin;
    return s;
}

struct Reader {
    const uint8_t *p; size_t len, pos = 0; bool bad = false, noncanonical = false;
    uint64_t word() {
        if (pos + 8 > len) { bad = true; pos = len; return 0; }
        uint64_t v; std::memcpy(&v, p + pos, 8); pos += 8;
        if (v >= P) noncanonical = true;
        return v;
    }
    uint8_t byte() { if (pos + 1 > len) { bad = true; return 0; } return p[pos++]; }
    void words(std::vector<uint64_t> &out, size_t n) { for (size_t i = 0; i < n && !bad; i++) out.push_back(word()); }
};

// a proof's field elements, vector by vector, as ProofWithPublicInputs::from_bytes(bytes, common_data) reads them
// (util::serialization order): the only lengths the bytes carry are the Merkle paths'
struct Parsed {
    Shape shape;
    std::vector<uint64_t> caps[3], openings[7], commit_caps, final_poly, public_inputs;   // openings in BYTE order: constants, sigmas, wires, zs, zs_next, pp, quotient
    struct Round { std::vector<std::vector<uint64_t>> evals, siblings, step_evals, step_siblings; };
    std::vector<Round> rounds;
    uint64_t pow_witness = 0;
};
std::string parse_proof(const CircuitPack &c, const uint8_t *bytes, size_t len, Parsed &out) {
    const Shape t = target_shape(c);
    Reader b{bytes, len};
    out.shape = t;
    for (int i = 0; i < 3; i++) b.words(out.caps[i], (size_t)t.wires_cap * 4);
    const int byte_order[7] = {0, 1, 2, 3, 4, 5, 6};
    for (int i : byte_order) b.words(out.openings[i], (size_t)t.openings[i] * 2);
    b.words(out.commit_caps, (size_t)t.commit_caps.size() * t.wires_cap * 4);
    out.rounds.assign(t.rounds.size(), Parsed::Round());
    for (size_t q = 0; q < t.rounds.size() && !b.bad; q++) {
        Parsed::Round &r = out.rounds[q];
        Shape::Round &sr = out.shape.rounds[q];
        for (size_t j = 0; j < t.rounds[q].evals.size(); j++) {
            r.evals.emplace_back(); b.words(r.evals.back(), t.rounds[q].evals[j]);
            const uint8_t n_sib = b.byte();                   // write_merkle_proof's one-byte length
            sr.siblings[j] = n_sib;
            r.siblings.emplace_back(); b.words(r.siblings.back(), (size_t)n_sib * 4);
        }
        for (size_t j = 0; j < t.rounds[q].step_evals.size(); j++) {
            r.step_evals.emplace_back(); b.words(r.step_evals.back(), (size_t)t.rounds[q].step_evals[j] * 2);
            const uint8_t n_sib = b.byte();
            sr.step_siblings[j] = n_sib;
            r.step_siblings.emplace_back(); b.words(r.step_siblings.back(), (size_t)n_sib * 4);
        }
    }
    b.words(out.final_poly, (size_t)t.final_poly * 2);
    out.pow_witness = b.word();
    // the public inputs take what is left: their count is the one length a caller can get wrong without breaking the layout
    if (!b.bad && (len - b.pos) % 8 == 0) {
        const size_t n = (len - b.pos) / 8;
        b.words(out.public_inputs, n);
        out.shape.public_inputs = (uint32_t)n;
    } else b.bad = true;
    if (b.bad || b.pos != len) return "proof bytes end inside a vector or leave trailing bytes (" + std::to_string(len) + " bytes)";
    if (b.noncanonical) return "proof holds a non-canonical field element";
    return "";
}

std::string pack_of(const uint64_t *words, size_t n, CircuitPack &c) {
    if (!words) return "null circuit pack";
    return c.parse(words, n);
}

size_t count_targets(const Shape &s) {
    size_t t = s.public_inputs + 4ull * (s.wires_cap + s.zs_pp_cap + s.quotient_cap);
    for (int i = 0; i < 9; i++) t += 2ull * s.openings[i];
    t += 1 + 2ull * s.final_poly;
    for (uint32_t c : s.commit_caps) t += 4ull * c;
    for (const Shape::Round &r : s.rounds) {
        for (size_t j = 0; j < r.evals.size(); j++) t += r.evals[j] + 4ull * r.siblings[j];
        for (size_t j = 0; j < r.step_evals.size(); j++) t += 2ull * r.step_evals[j] + 4ull * r.step_siblings[j];
    }
    return t;
}

// ensure_len_matches (common/utils.rs:295-317): the reference's message, word for word
int len_mismatch(char *err, size_t actual, size_t expected, const char *label, size_t slot, const std::string &what) {
    return fail(err, -1, "%s at slot %zu is malformed: %s has length %zu, but the circuit expects %zu", label, slot, what.c_str(), actual, expected);
}
int ensure_shape(const Shape &t, const Shape &p, size_t slot, const char *label, char *err) {
#define LEN(a, e, what) do { if ((size_t)(a) != (size_t)(e)) return len_mismatch(err, (a), (e), label, slot, (what)); } while (0)
    LEN(p.public_inputs, t.public_inputs, "public inputs");
    LEN(p.wires_cap, t.wires_cap, "wires_cap");
    LEN(p.zs_pp_cap, t.zs_pp_cap, "plonk_zs_partial_products_cap");
    LEN(p.quotient_cap, t.quotient_cap, "quotient_polys_cap");
    static const char *names[9] = {"openings.constants", "openings.plonk_sigmas", "openings.wires", "openings.plonk_zs", "openings.plonk_zs_next",
                                   "openings.partial_products", "openings.quotient_polys", "openings.lookup_zs", "openings.lookup_zs_next"};
    for (int i = 0; i < 9; i++) LEN(p.openings[i], t.openings[i], names[i]);
    LEN(p.commit_caps.size(), t.commit_caps.size(), "opening_proof.commit_phase_merkle_caps");
    for (size_t i = 0; i < t.commit_caps.size(); i++) LEN(p.commit_caps[i], t.commit_caps[i], "opening_proof.commit_phase_merkle_caps[" + std::to_string(i) + "]");
    LEN(p.rounds.size(), t.rounds.size(), "opening_proof.query_round_proofs");
    for (size_t i = 0; i < t.rounds.size(); i++) {
        const Shape::Round &pr = p.rounds[i], &tr = t.rounds[i];
        const std::string q = "opening_proof.query_round_proofs[" + std::to_string(i) + "]";
        LEN(pr.evals.size(), tr.evals.size(), q + ".initial_trees_proof.evals_proofs");
        for (size_t j = 0; j < tr.evals.size(); j++) {
            LEN(pr.evals[j], tr.evals[j], q + ".initial_trees_proof.evals_proofs[" + std::to_string(j) + "].evals");
            LEN(pr.siblings[j], tr.siblings[j], q + ".initial_trees_proof.evals_proofs[" + std::to_string(j) + "].siblings");
        }
        LEN(pr.step_evals.size(), tr.step_evals.size(), q + ".steps");
        for (size_t j = 0; j < tr.step_evals.size(); j++) {
            LEN(pr.step_evals[j], tr.step_evals[j], q + ".steps[" + std::to_string(j) + "].evals");
            LEN(pr.step_siblings[j], tr.step_siblings[j], q + ".steps[" + std::to_string(j) + "].merkle_proof.siblings");
        }
    }
    LEN(p.final_poly, t.final_poly, "opening_proof.final_poly");
#undef LEN
    return 0;
}

// the values of one proof in logical-target order (see qpgpu_proof_target_count)
void emit_values(const Parsed &p, std::vector<uint64_t> &v) {
    auto put = [&](const std::vector<uint64_t> &x) { v.insert(v.end(), x.begin(), x.end()); };
    put(p.public_inputs);
    for (int i = 0; i < 3; i++) put(p.caps[i]);
    // OpeningSet::to_fri_openings: the zeta batch (constants, plonk_sigmas, wires, plonk_zs, partial_products, quotient_polys),
    // then the zeta-next batch (plonk_zs_next); byte order has plonk_zs_next in front of partial_products
    const int batch_order[7] = {0, 1, 2, 3, 5, 6, 4};
    for (int i : batch_order) put(p.openings[i]);
    v.push_back(p.pow_witness);
    put(p.final_poly);
    put(p.commit_caps);
    for (const Parsed::Round &r : p.rounds) {
        for (size_t j = 0; j < r.evals.size(); j++) { put(r.evals[j]); put(r.siblings[j]); }
        for (size_t j = 0; j < r.step_evals.size(); j++) { put(r.step_evals[j]); put(r.step_siblings[j]); }
    }
}

}  // namespace

extern "C" {

int qpgpu_proof_target_shape(const uint64_t *inner_pack, size_t n_words, uint32_t *out, size_t cap, size_t *count, char *err) {
    CircuitPack c;
    const std::string why = pack_of(inner_pack, n_words, c);
    if (!why.empty()) return fail(err, -1, "circuit pack: %s", why.c_str());
    const std::vector<uint32_t> f = target_shape(c).flatten();
    if (count) *count = f.size();
    if (out) {
        if (cap < f.size()) return fail(err, -5, "shape buffer too small: %zu words needed", f.size());
        std::memcpy(out, f.data(), f.size() * 4);
    }
    return 0;
}

int qpgpu_proof_shape_of_bytes(const uint64_t *inner_pack, size_t n_words, const uint8_t *proof, size_t len, uint32_t *out, size_t cap, size_t *count, char *err) {
    CircuitPack c;
    const std::string why = pack_of(inner_pack, n_words, c);
    if (!why.empty()) return fail(err, -1, "circuit pack: %s", why.c_str());
    if (!proof) return fail(err, -1, "null proof");
    Parsed p;
    const std::string bad = parse_proof(c, proof, len, p);
    if (!bad.empty()) return fail(err, -1, "%s", bad.c_str());
    const std::vector<uint32_t> f = p.shape.flatten();
    if (count) *count = f.size();
    if (out) {
        if (cap < f.size()) return fail(err, -5, "shape buffer too small: %zu words needed", f.size());
        std::memcpy(out, f.data(), f.size() * 4);
    }
    return 0;
}

int qpgpu_ensure_proof_shape_matches_targets(const uint32_t *target_shape_words, size_t n_target, const uint32_t *proof_shape_words, size_t n_proof,
                                             size_t slot, const char *label, char *err) {
    Shape t, p;
    if (!target_shape_words || !t.unflatten(target_shape_words, n_target)) return fail(err, -1, "target shape descriptor is not well formed");
    if (!proof_shape_words) return fail(err, -1, "proof shape descriptor is not well formed");
    // a proof's descriptor may have any vector counts: read it leniently, list by list, the way the reference's zips do
    // (the outer length is compared first; inner entries are only visited up to the target's count)
    {
        size_t q = 0;
        auto get = [&](uint32_t &v) { if (q >= n_proof) return false; v = proof_shape_words[q++]; return true; };
        bool ok = get(p.public_inputs) && get(p.wires_cap) && get(p.zs_pp_cap) && get(p.quotient_cap);
        for (int i = 0; ok && i < 9; i++) ok = get(p.openings[i]);
        uint32_t k = 0;
        ok = ok && get(k) && k <= 4096;
        if (ok) { p.commit_caps.resize(k); for (uint32_t &c : p.commit_caps) ok = ok && get(c); }
        ok = ok && get(k) && k <= 65536;
        if (ok) {
            p.rounds.assign(k, Shape::Round());
            for (Shape::Round &r : p.rounds) {
                uint32_t m = 0;
                ok = ok && get(m) && m <= 4096;
                if (!ok) break;
                r.evals.resize(m); r.siblings.resize(m);
                for (uint32_t j = 0; ok && j < m; j++) ok = get(r.evals[j]) && get(r.siblings[j]);
                ok = ok && get(m) && m <= 4096;
                if (!ok) break;
                r.step_evals.resize(m); r.step_siblings.resize(m);
                for (uint32_t j = 0; ok && j < m; j++) ok = get(r.step_evals[j]) && get(r.step_siblings[j]);
            }
        }
        ok = ok && get(p.final_poly) && q == n_proof;
        if (!ok) return fail(err, -1, "proof shape descriptor is not well formed");
    }
    // ensure_shape compares a list's length before it indexes into it, so any well-formed descriptor is safe to compare
    return ensure_shape(t, p, slot, label ? label : "proof", err);
}

size_t qpgpu_proof_target_count(const uint64_t *inner_pack, size_t n_words) {
    CircuitPack c;
    if (!pack_of(inner_pack, n_words, c).empty()) return 0;
    return count_targets(target_shape(c));
}

int qpgpu_proof_target_values(const uint64_t *inner_pack, size_t n_words, const uint8_t *proof, size_t len, size_t slot, const char *label,
                              uint64_t *values_out, size_t cap, size_t *count, char *err) {
    CircuitPack c;
    const std::string why = pack_of(inner_pack, n_words, c);
    if (!why.empty()) return fail(err, -1, "circuit pack: %s", why.c_str());
    if (!proof) return fail(err, -1, "null proof");
    Parsed p;
    const std::string bad = parse_proof(c, proof, len, p);
    if (!bad.empty()) return fail(err, -1, "%s at slot %zu is malformed: %s", label ? label : "proof", slot, bad.c_str());
    const int rc = ensure_shape(target_shape(c), p.shape, slot, label ? label : "proof", err);
    if (rc) return rc;
    std::vector<uint64_t> v;
    emit_values(p, v);
    if (count) *count = v.size();
    if (values_out) {
        if (cap < v.size()) return fail(err, -5, "value buffer too small: %zu words needed", v.size());
        std::memcpy(values_out, v.data(), v.size() * 8);
    }
    return 0;
}

int qpgpu_batch_fill_proof_targets(const uint64_t *inner_pack, size_t n_words, const uint8_t *const *proofs, const size_t *proof_lens, size_t num_proofs,
                                   size_t num_proof_targets, const uint64_t *dummy_nullifier_preimages, size_t num_preimages, size_t num_preimage_targets,
                                   const char *label, uint32_t *targets_out, uint64_t *values_out, size_t cap, size_t *count, char *err) {
    // the three count checks of fill_private_batch_witness, with its messages (witness.rs:23-45)
    if (num_proofs != num_proof_targets)
        return fail(err, -1, "proof count mismatch: got %zu, but circuit expects %zu leaf proofs", num_proofs, num_proof_targets);
    if (num_preimage_targets != num_proof_targets)
        return fail(err, -1, "target layout is inconsistent: dummy_nullifier_pre_image target count %zu != leaf proof target count %zu", num_preimage_targets, num_proof_targets);
    if (num_preimages != num_proof_targets)
        return fail(err, -1, "dummy nullifier preimage count mismatch: got %zu, but circuit expects %zu", num_preimages, num_proof_targets);
    if (num_proofs && (!proofs || !proof_lens || !dummy_nullifier_preimages)) return fail(err, -1, "null argument");
    CircuitPack c;
    const std::string why = pack_of(inner_pack, n_words, c);
    if (!why.empty()) return fail(err, -1, "circuit pack: %s", why.c_str());
    const Shape t = target_shape(c);
    const size_t T = count_targets(t), total = num_proofs * (T + 4);
    if (total > 0xFFFFFFFFull) return fail(err, -1, "too many targets for 32-bit logical ids");
    if (count) *count = total;
    const bool write = targets_out && values_out;
    if (write && cap < total) return fail(err, -5, "assignment buffers too small: %zu entries needed", total);
    size_t k = 0;
    std::vector<uint64_t> v;
    for (size_t i = 0; i < num_proofs; i++) {
        if (!proofs[i]) return fail(err, -1, "null proof at slot %zu", i);
        Parsed p;
        const std::string bad = parse_proof(c, proofs[i], proof_lens[i], p);
        if (!bad.empty()) return fail(err, -1, "%s at slot %zu is malformed: %s", label ? label : "leaf proof", i, bad.c_str());
        const int rc = ensure_shape(t, p.shape, i, label ? label : "leaf proof", err);
        if (rc) return rc;
        v.clear();
        emit_values(p, v);
        if (write) for (size_t j = 0; j < T; j++, k++) { targets_out[k] = (uint32_t)(i * T + j); values_out[k] = v[j]; }
    }
    for (size_t i = 0; i < num_proofs; i++)
        for (size_t limb = 0; limb < 4; limb++) {
            const uint64_t x = dummy_nullifier_preimages[4 * i + limb];
            if (x >= P) return fail(err, -1, "failed to set dummy nullifier preimage target at slot %zu, limb %zu: value is not a canonical field element", i, limb);
            if (write) { targets_out[k] = (uint32_t)(num_proofs * T + 4 * i + limb); values_out[k] = x; k++; }
        }
    return 0;
}

}  // extern "C"
