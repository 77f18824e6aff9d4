// wrapper_circuit.cpp — plonky2's recursive verifier and the reference's two batch circuits, restated on the native builder
// (builder.hpp): SURVEY.md §8 rows a3 / a4 / a6.
//
//   add_recursive_verifiers                      wormhole/aggregator/src/common/recursive.rs:74-102
//     builder.add_virtual_proof_with_pis(common)   one virtual target per field element of an inner proof (proof_targets.cpp's order)
//     builder.verify_proof::<C>(proof, vd, common) qp-plonky2 1.5.5, un-vendored; restated from upstream plonky2:
//       public_inputs_hash = hash_n_to_hash_no_pad(public_inputs)                               plonk/recursive_verifier.rs
//       get_challenges on a RecursiveChallenger, fri_verify_proof_of_work                       plonk/get_challenges.rs, iop/challenger.rs
//       eval_vanishing_poly_circuit at zeta == Z_H(zeta) * reduce(quotient chunks)             plonk/vanishing_poly.rs  (verify_math.hpp)
//       fri_verifier_query_round: x_index bits, cap_index = le_sum(high bits),                  fri/recursive_verifier.rs
//         fri_verify_initial_proof: verify_merkle_proof_to_cap_with_cap_index per oracle        hash/merkle_proofs.rs
//         fri_combine_initial (ReducingGate rows), per reduction step: random_access_extension of the previous evaluation,
//         compute_evaluation (CosetInterpolationGate), verify_merkle_proof_to_cap_with_cap_index(flatten(evals), ..), final polynomial
//   build_private_batch_constraints              wormhole/aggregator/src/private_batch/circuit/circuit_logic.rs:171-477
//   build_public_batch_constraints               wormhole/aggregator/src/public_batch/circuit/circuit_logic.rs:167-317
//   the gadgets both use (bytes_digest_eq, u32_lt, split_canonical_u32_halves, halves8_lt, sort_digests4)   common/src/gadgets.rs:144-334
//
// The flags of qpgpu_wrapper_circuit_build (include/qpgpu_batch.h) choose how much of this a circuit carries: none = the commitment
// half (Merkle paths of every opened row and FRI coset; query indices are inputs), TRANSCRIPT = the Fiat-Shamir replay (indices and
// proof of work in-circuit), VERIFY = the arithmetic half — with both, everything VerifierCircuitData::verify checks —,
// PRIVATE_BATCH / PUBLIC_BATCH = the layer's own constraints and public inputs, ZERO_KNOWLEDGE = CircuitBuilder::blind's rows.
// The inner circuit's verifier data (constants/sigmas cap, circuit digest) are CONSTANTS of the wrapper, which is what makes a proof
// of another circuit of the same shape unusable (private_batch_rejects_malicious_circuit_proofs). A byte flipped anywhere in an
// inner proof, or an inner proof made honestly from a trace that violates the inner circuit, leaves the wrapper without a witness
// ("set twice with different values"); tests/test_wrapper_circuit*.py, test_batch_circuits*.py, tools/fuzz_wrapper_tamper.py.
#include <array>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include "../../include/qpgpu.h"
#include "../../include/qpgpu_batch.h"
#include "builder.hpp"
#include "gadgets.hpp"
#include "poseidon.hpp"
#include "verify_math.hpp"

using cb::BoolTarget;
using cb::Builder;
using cb::HashOutTarget;
using cb::Target;
using gl::u64;

// ---- verify_math.hpp's element type over the circuit builder: an ExtensionTarget, every operation one ArithmeticExtensionGate
// slot (a b + c or a - b forms; constants are ConstantGate targets, so all operations of a circuit share two kinds of rows) ----
namespace cbx {
thread_local Builder *g_b = nullptr;
struct XT { cb::ExtTarget t; };
inline XT K(u64 c) { return {g_b->constant_ext(c)}; }
inline XT operator+(XT a, XT b) { return {g_b->arithmetic_ext(1, 1, g_b->one_ext(), a.t, b.t)}; }
inline XT operator-(XT a, XT b) { return {g_b->arithmetic_ext(1, gl::P - 1, g_b->one_ext(), a.t, b.t)}; }
inline XT operator*(XT a, XT b) { return {g_b->arithmetic_ext(1, 1, a.t, b.t, g_b->zero_ext())}; }
inline XT madd(XT a, XT b, XT c) { return {g_b->arithmetic_ext(1, 1, a.t, b.t, c.t)}; }
inline XT scale(XT x, u64 s) { return x * K(s); }
inline XT sadd(XT a, u64 s, XT c) { return madd(a, K(s), c); }
inline XT lift(Target t) { return {g_b->to_ext(t)}; }
}  // namespace cbx
namespace vmath { template <> inline cbx::XT konst<cbx::XT>(u64 c) { return cbx::K(c); } }
namespace cbx {
// the PoseidonGate's MDS layer at zeta: one PoseidonMdsGate row instead of 144 multiply-adds (what eval_unfiltered_circuit does
// upstream). Found by argument-dependent lookup from verify_math.hpp's poseidon_gate<XT> (a better match than the generic template).
inline void mds_ext(XT (&s)[12]) {
    std::array<cb::ExtTarget, 12> in;
    for (int i = 0; i < 12; i++) in[i] = s[i].t;
    const std::array<cb::ExtTarget, 12> out = g_b->poseidon_mds_ext(in);
    for (int i = 0; i < 12; i++) s[i].t = out[i];
}
}  // namespace cbx
using cbx::XT;

namespace {

struct QueryRoundTargets {
    std::vector<Target> evals[4]; std::vector<HashOutTarget> siblings[4];
    std::vector<std::vector<Target>> step_evals; std::vector<std::vector<HashOutTarget>> step_siblings;
};
struct ProofTargets {
    std::vector<Target> all;                         // every target in logical order (qpgpu_proof_target_count of them)
    std::vector<Target> public_inputs;
    std::vector<HashOutTarget> caps[3];              // wires, zs / partial products, quotient
    std::vector<Target> openings, final_poly;        // in transcript order, two targets per extension element
    Target pow_witness = cb::NO_TARGET;
    std::vector<std::vector<HashOutTarget>> commit_caps;
    std::vector<QueryRoundTargets> rounds;
};

// add_virtual_proof_with_pis, in the logical order of include/qpgpu_batch.h ("inner-proof targets")
ProofTargets add_virtual_proof(Builder &b, const CircuitPack &c) {
    ProofTargets t;
    auto one = [&]() { const Target x = b.add_virtual_target(); t.all.push_back(x); return x; };
    auto hash = [&]() { HashOutTarget h; for (auto &e : h.elements) e = one(); return h; };
    const uint32_t cap = 1u << c.cap_height, nch = (uint32_t)c.num_challenges, salt = c.zero_knowledge ? 4 : 0;
    const uint32_t L = (uint32_t)(c.degree_bits + c.rate_bits);
    for (uint64_t i = 0; i < c.num_public_inputs; i++) t.public_inputs.push_back(one());
    for (int k = 0; k < 3; k++) for (uint32_t i = 0; i < cap; i++) t.caps[k].push_back(hash());
    // openings at zeta (constants, plonk_sigmas, wires, plonk_zs, partial_products, quotient_polys), then plonk_zs_next: 2 per element
    const uint64_t openings = (c.num_selectors + c.num_constants) + c.num_routed_wires + c.num_wires + nch + nch * c.num_partial_products + c.num_quotient_cols() + nch;
    for (uint64_t i = 0; i < 2 * openings; i++) t.openings.push_back(one());
    t.pow_witness = one();
    uint32_t fin = (uint32_t)c.degree_bits;
    for (uint64_t ab : c.arity_bits) fin -= (uint32_t)ab;
    for (uint32_t i = 0; i < (2u << fin); i++) t.final_poly.push_back(one());
    for (size_t r = 0; r < c.arity_bits.size(); r++) { t.commit_caps.emplace_back(); for (uint32_t i = 0; i < cap; i++) t.commit_caps.back().push_back(hash()); }
    const uint32_t widths[4] = {(uint32_t)c.num_cs_cols(), (uint32_t)c.num_wires + salt, (uint32_t)c.num_zs_pp_cols() + salt, (uint32_t)c.num_quotient_cols() + salt};
    t.rounds.resize(c.num_query_rounds);
    for (QueryRoundTargets &q : t.rounds) {
        for (int o = 0; o < 4; o++) {
            for (uint32_t i = 0; i < widths[o]; i++) q.evals[o].push_back(one());
            for (uint32_t i = 0; i < L - (uint32_t)c.cap_height; i++) q.siblings[o].push_back(hash());
        }
        uint32_t lvl = L;
        for (uint64_t ab : c.arity_bits) {
            lvl -= (uint32_t)ab;
            q.step_evals.emplace_back(); q.step_siblings.emplace_back();
            for (uint32_t i = 0; i < (2u << ab); i++) q.step_evals.back().push_back(one());
            for (uint32_t i = 0; i < lvl - (uint32_t)c.cap_height; i++) q.step_siblings.back().push_back(hash());
        }
    }
    return t;
}

// RecursiveChallenger (iop/challenger.rs): the duplex sponge of the transcript, in-circuit. observe buffers; a challenge first
// absorbs the buffered inputs eight at a time (overwrite mode, one permutation per chunk), squeezes the rate part and pops from
// the END of the output buffer — the order the prover's and the host verifier's transcripts use (csrc/prover_host.hpp).
struct RecursiveChallenger {
    Builder &b;
    Builder::State state;
    std::vector<Target> in, out;
    explicit RecursiveChallenger(Builder &bb) : b(bb) { state.fill(b.zero()); }
    void observe(Target t) { out.clear(); in.push_back(t); }
    void observe(const std::vector<Target> &v) { for (Target t : v) observe(t); }
    void observe(const HashOutTarget &h) { for (Target t : h.elements) observe(t); }
    void observe_cap(const std::vector<HashOutTarget> &cap) { for (const HashOutTarget &h : cap) observe(h); }
    void absorb() {
        if (in.empty()) return;
        for (size_t i = 0; i < in.size(); i += 8) {
            const size_t len = std::min<size_t>(8, in.size() - i);
            for (size_t k = 0; k < len; k++) state[k] = in[i + k];
            state = b.permute_swapped(state, b._false());
        }
        in.clear();
        out.assign(state.begin(), state.begin() + 8);
    }
    cb::ExtTarget get_ext() { const Target a = get(), c = get(); return {{a, c}}; }
    Target get() {
        absorb();
        if (out.empty()) { state = b.permute_swapped(state, b._false()); out.assign(state.begin(), state.begin() + 8); }
        const Target t = out.back();
        out.pop_back();
        return t;
    }
};

// ---- the wrapper-specific logic of the two batch layers, over the inner proofs' public-input targets -------------------------
using gadgets::Digest;
using gadgets::digest_eq;
using gadgets::sort_digests4;
using gadgets::split_canonical_u32_halves;
using gadgets::u32_lt;
Digest digest_at(const std::vector<Target> &pis, size_t off) { return {pis[off], pis[off + 1], pis[off + 2], pis[off + 3]}; }

// leaf public inputs (wormhole/inputs: asset_id, output_amount_1, output_amount_2, volume_fee_bps, nullifier(4), exit_account_1(4),
// exit_account_2(4), block_hash(4), block_number)
enum : size_t { LEAF_ASSET = 0, LEAF_OUT_1 = 1, LEAF_OUT_2 = 2, LEAF_FEE = 3, LEAF_NULLIFIER = 4, LEAF_EXIT_1 = 8, LEAF_EXIT_2 = 12, LEAF_BLOCK_HASH = 16, LEAF_BLOCK_NUMBER = 20, LEAF_PI_LEN = 21 };

// build_private_batch_constraints (wormhole/aggregator/src/private_batch/circuit/circuit_logic.rs:171-477): the public inputs the
// private-batch circuit registers, [num_exit_slots, asset_id, volume_fee_bps, block_hash(4), block_number, (sum, exit(4)) x 2N,
// nullifier(4) x N sorted, zero padding to 21 N + 8]
std::vector<Target> private_batch_logic(Builder &b, const std::vector<std::vector<Target>> &pis, const std::vector<std::vector<Target>> &preimages) {
    const size_t n = pis.size();
    const Target one = b.one(), zero = b.zero();
    const Target num_exit_slots_t = b.constant(2 * n);
    const Target asset_ref = pis[0][LEAF_ASSET];
    const Digest sentinel = {zero, zero, zero, zero};
    std::vector<BoolTarget> is_dummy; std::vector<Digest> block_hashes;
    for (size_t i = 0; i < n; i++) { block_hashes.push_back(digest_at(pis[i], LEAF_BLOCK_HASH)); is_dummy.push_back(digest_eq(b, block_hashes[i], sentinel)); }
    // the references come from the first non-dummy slot (prefix scan)
    BoolTarget found_real = b._false();
    Digest block_ref = sentinel; Target block_number_ref = zero, fee_ref = zero;
    for (size_t i = 0; i < n; i++) {
        const BoolTarget is_real = b.not_(is_dummy[i]), not_found_yet = b.not_(found_real), take = b.and_(is_real, not_found_yet);
        for (int j = 0; j < 4; j++) block_ref[j] = b.select(take, block_hashes[i][j], block_ref[j]);
        block_number_ref = b.select(take, pis[i][LEAF_BLOCK_NUMBER], block_number_ref);
        fee_ref = b.select(take, pis[i][LEAF_FEE], fee_ref);
        found_real = b.or_(found_real, is_real);
    }
    std::vector<Target> out = {num_exit_slots_t, asset_ref, fee_ref};
    for (size_t i = 0; i < n; i++) {
        const BoolTarget matches_ref = digest_eq(b, block_hashes[i], block_ref);
        b.connect(b.or_(is_dummy[i], matches_ref).target, one);
        b.connect(pis[i][LEAF_ASSET], asset_ref);
        const BoolTarget fee_matches = b.is_equal(pis[i][LEAF_FEE], fee_ref);
        b.connect(b.or_(is_dummy[i], fee_matches).target, one);
    }
    out.insert(out.end(), block_ref.begin(), block_ref.end());
    out.push_back(block_number_ref);
    // exit-account grouping: dummy slots masked to (zero account, 0) at ingress, amounts summed per account, duplicates zeroed
    const size_t slots = 2 * n;
    std::vector<Digest> slot_exits(slots); std::vector<Target> slot_amounts(slots);
    for (size_t s = 0; s < slots; s++) {
        const std::vector<Target> &p = pis[s / 2];
        const Digest exit_raw = digest_at(p, s % 2 == 0 ? LEAF_EXIT_1 : LEAF_EXIT_2);
        const Target amount_raw = p[s % 2 == 0 ? LEAF_OUT_1 : LEAF_OUT_2];
        for (int j = 0; j < 4; j++) slot_exits[s][j] = b.select(is_dummy[s / 2], zero, exit_raw[j]);
        slot_amounts[s] = b.select(is_dummy[s / 2], zero, amount_raw);
    }
    for (size_t s = 0; s < slots; s++) {
        const Digest exit_slot = slot_exits[s];
        BoolTarget is_duplicate = b._false();
        for (size_t e = 0; e < s; e++) is_duplicate = b.or_(is_duplicate, digest_eq(b, slot_exits[e], exit_slot));
        Target acc = zero;
        for (size_t e = 0; e < slots; e++) {
            const BoolTarget m = digest_eq(b, slot_exits[e], exit_slot);
            acc = b.add(acc, b.select(m, slot_amounts[e], zero));
        }
        const Target final_sum = b.select(is_duplicate, zero, acc);
        Digest final_exit;
        for (int j = 0; j < 4; j++) final_exit[j] = b.select(is_duplicate, zero, exit_slot[j]);
        b.range_check(final_sum, 32);
        out.push_back(final_sum);
        out.insert(out.end(), final_exit.begin(), final_exit.end());
    }
    // pairwise distinct nullifiers among the real slots
    std::vector<Digest> real_nullifiers;
    for (size_t i = 0; i < n; i++) real_nullifiers.push_back(digest_at(pis[i], LEAF_NULLIFIER));
    for (size_t i = 0; i < n; i++) {
        const BoolTarget is_real_i = b.not_(is_dummy[i]);
        for (size_t j = i + 1; j < n; j++) {
            const BoolTarget is_real_j = b.not_(is_dummy[j]), both_real = b.and_(is_real_i, is_real_j);
            const BoolTarget equal = digest_eq(b, real_nullifiers[i], real_nullifiers[j]);
            b.connect(b.and_(both_real, equal).target, zero);
        }
    }
    // dummy slots' nullifiers replaced by H(H(preimage)), the selected nullifiers emitted in sorted order
    std::vector<Digest> selected(n);
    for (size_t i = 0; i < n; i++) {
        const HashOutTarget inner = b.hash_n_to_hash_no_pad_p2(preimages[i]);
        const HashOutTarget dummy_null = b.hash_n_to_hash_no_pad_p2(std::vector<Target>(inner.elements, inner.elements + 4));
        for (int j = 0; j < 4; j++) selected[i][j] = b.select(is_dummy[i], dummy_null.elements[j], real_nullifiers[i][j]);
    }
    for (const Digest &d : sort_digests4(b, selected)) out.insert(out.end(), d.begin(), d.end());
    out.resize(LEAF_PI_LEN * n + 8, zero);
    return out;
}

// build_public_batch_constraints (wormhole/aggregator/src/public_batch/circuit/circuit_logic.rs:167-317): [aggregator_address(4),
// asset_id, volume_fee_bps, block_hash(4), block_number, total_exit_slots, (sum, exit(4)) x m * 2N, nullifier(4) x m * N], the
// segments of dummy inner proofs zeroed; no shuffle, no grouping across inner proofs
std::vector<Target> public_batch_logic(Builder &b, const std::vector<std::vector<Target>> &pis, size_t n_leaf, const std::vector<Target> &aggregator_address) {
    enum : size_t { PB_ASSET = 1, PB_FEE = 2, PB_BLOCK_HASH = 3, PB_BLOCK_NUMBER = 7, PB_EXIT_SLOTS = 8 };
    const size_t m = pis.size();
    const Target one = b.one(), zero = b.zero();
    const Digest sentinel = {zero, zero, zero, zero};
    std::vector<BoolTarget> is_dummy; std::vector<Digest> block_hashes;
    for (size_t i = 0; i < m; i++) { block_hashes.push_back(digest_at(pis[i], PB_BLOCK_HASH)); is_dummy.push_back(digest_eq(b, block_hashes[i], sentinel)); }
    BoolTarget found_real = b._false();
    Digest block_ref = sentinel; Target block_number_ref = zero, asset_ref = zero, fee_ref = zero;
    for (size_t i = 0; i < m; i++) {
        const BoolTarget is_real = b.not_(is_dummy[i]), not_found_yet = b.not_(found_real), take = b.and_(is_real, not_found_yet);
        for (int j = 0; j < 4; j++) block_ref[j] = b.select(take, block_hashes[i][j], block_ref[j]);
        block_number_ref = b.select(take, pis[i][PB_BLOCK_NUMBER], block_number_ref);
        asset_ref = b.select(take, pis[i][PB_ASSET], asset_ref);
        fee_ref = b.select(take, pis[i][PB_FEE], fee_ref);
        found_real = b.or_(found_real, is_real);
    }
    std::vector<Target> out(aggregator_address);
    out.push_back(asset_ref); out.push_back(fee_ref);
    for (size_t i = 0; i < m; i++) {
        b.connect(b.or_(is_dummy[i], b.is_equal(pis[i][PB_ASSET], asset_ref)).target, one);
        b.connect(b.or_(is_dummy[i], b.is_equal(pis[i][PB_FEE], fee_ref)).target, one);
        b.connect(b.or_(is_dummy[i], digest_eq(b, block_hashes[i], block_ref)).target, one);
    }
    out.insert(out.end(), block_ref.begin(), block_ref.end());
    out.push_back(block_number_ref);
    out.push_back(b.constant(m * 2 * n_leaf));
    const size_t nullifiers_start = PB_EXIT_SLOTS + 2 * n_leaf * 5;
    for (size_t i = 0; i < m; i++) for (size_t k = PB_EXIT_SLOTS; k < nullifiers_start; k++) out.push_back(b.select(is_dummy[i], zero, pis[i][k]));
    for (size_t i = 0; i < m; i++) for (size_t k = nullifiers_start; k < nullifiers_start + 4 * n_leaf; k++) out.push_back(b.select(is_dummy[i], zero, pis[i][k]));
    return out;
}

// ---- the arithmetic half of verify_proof: the openings against the vanishing polynomial at zeta, and the FRI consistency checks ----
struct Challenges { std::vector<Target> betas, gammas, alphas; cb::ExtTarget zeta, fri_alpha; std::vector<cb::ExtTarget> fri_betas; };
struct FriPre { XT red0, red1, zeta, g_zeta, alpha, alpha_nch; };
struct FriQuery { XT old_eval; Target subgroup_x; };

// verify_proof_with_challenges_circuit up to the FRI call (plonk/recursive_verifier.rs): eval_vanishing_poly_circuit at zeta — every
// gate of the INNER circuit through verify_math.hpp's generic constraints — equals Z_H(zeta) * reduce_with_powers(quotient chunks,
// zeta^n) for every challenge; then precompute_reduced_evals (fri/recursive_verifier.rs). p.openings is in transcript order:
// constants/sigmas, wires, Zs, partial products, quotient chunks, Zs at g zeta.
std::string verify_openings(Builder &b, const CircuitPack &c, const ProofTargets &p, const HashOutTarget &pih, const Challenges &ch, FriPre &fri) {
    const size_t ncs = c.num_cs_cols(), NW = c.num_wires, nch = c.num_challenges, npp = c.num_partial_products, qdf = c.quotient_degree_factor, nq = nch * qdf;
    if (p.openings.size() != 2 * (ncs + NW + nch + nch * npp + nq + nch)) return "opening count disagrees with the inner circuit";
    std::vector<XT> o(p.openings.size() / 2);
    for (size_t i = 0; i < o.size(); i++) o[i] = XT{{{p.openings[2 * i], p.openings[2 * i + 1]}}};
    const XT *o_cs = o.data(), *o_w = o_cs + ncs, *o_zs = o_w + NW, *o_pp = o_zs + nch, *o_q = o_pp + nch * npp, *o_zn = o_q + nq;
    const XT zeta{ch.zeta}, one = cbx::K(1);
    XT zeta_n = zeta;
    for (uint64_t i = 0; i < c.degree_bits; i++) zeta_n = zeta_n * zeta_n;
    const XT zh = zeta_n - one;
    // eval_l_0_circuit: (zeta^n - 1) / (n (zeta - 1))
    const XT l0{b.div_ext(zh.t, scale(zeta - one, 1ull << c.degree_bits).t)};
    std::vector<XT> betas, gammas, alphas, van;
    for (size_t k = 0; k < nch; k++) { betas.push_back(cbx::lift(ch.betas[k])); gammas.push_back(cbx::lift(ch.gammas[k])); alphas.push_back(cbx::lift(ch.alphas[k])); }
    XT xpih[4];
    for (int i = 0; i < 4; i++) xpih[i] = cbx::lift(pih.elements[i]);
    const std::string why = vmath::vanishing_at_zeta<XT>(c, zeta, l0, o_cs, o_w, o_zs, o_zn, o_pp, betas.data(), gammas.data(), alphas.data(), xpih, van);
    if (!why.empty()) return why;
    for (size_t k = 0; k < nch; k++) {
        XT qv = cbx::K(0);
        for (size_t j = qdf; j-- > 0;) qv = madd(qv, zeta_n, o_q[k * qdf + j]);
        b.connect_ext(van[k].t, (zh * qv).t);
    }
    // precompute_reduced_evals: batch 0 = everything opened at zeta in oracle order, batch 1 = Zs at g zeta
    fri.alpha = XT{ch.fri_alpha};
    {
        std::vector<cb::ExtTarget> at_zeta, at_g_zeta;
        for (size_t j = 0; j < ncs + NW + nch + nch * npp + nq; j++) at_zeta.push_back(o[j].t);
        for (size_t j = 0; j < nch; j++) at_g_zeta.push_back(o_zn[j].t);
        fri.red0 = XT{b.reduce_ext(ch.fri_alpha, at_zeta)};
        fri.red1 = XT{b.reduce_ext(ch.fri_alpha, at_g_zeta)};
    }
    fri.zeta = zeta;
    fri.g_zeta = scale(zeta, gl::canon(gl::root_of_unity((unsigned)c.degree_bits)));
    fri.alpha_nch = cbx::K(1);
    for (size_t k = 0; k < nch; k++) fri.alpha_nch = fri.alpha_nch * fri.alpha;
    return "";
}

// fri_verifier_query_round, first part: subgroup_x = g * w^rev(x_index) (and its inverse, for the coset interpolation) from the
// index bits; fri_combine_initial: the opened rows reduced with alpha (salts are not opened), minus the reduced openings, over
// (x - zeta) and (x - g zeta)
FriQuery fri_query_begin(Builder &b, const CircuitPack &c, const QueryRoundTargets &r, const std::vector<BoolTarget> &bits, const Challenges &, const FriPre &fri) {
    const unsigned L = (unsigned)bits.size();
    const size_t nch = c.num_challenges, polys[4] = {(size_t)c.num_cs_cols(), (size_t)c.num_wires, (size_t)c.num_zs_pp_cols(), (size_t)c.num_quotient_cols()};
    const std::vector<BoolTarget> rev(bits.rbegin(), bits.rend());               // rev(x_index), little endian
    const u64 w = gl::canon(gl::root_of_unity(L));
    FriQuery q;
    q.subgroup_x = b.mul_const(gl::MULT_GEN, b.exp_from_bits_const_base(w, rev));
    std::vector<Target> at_zeta, at_g_zeta;
    for (int o = 0; o < 4; o++) at_zeta.insert(at_zeta.end(), r.evals[o].begin(), r.evals[o].begin() + (long)polys[o]);
    at_g_zeta.assign(r.evals[2].begin(), r.evals[2].begin() + (long)nch);
    const XT e0{b.reduce_base(fri.alpha.t, at_zeta)}, e1{b.reduce_base(fri.alpha.t, at_g_zeta)};
    const XT sx = cbx::lift(q.subgroup_x);
    XT sum{b.div_ext((e0 - fri.red0).t, (sx - fri.zeta).t)};
    const XT second{b.div_ext((e1 - fri.red1).t, (sx - fri.g_zeta).t)};
    q.old_eval = madd(sum, fri.alpha_nch, second);
    return q;
}

// one reduction step: the step's coset of evaluations holds the running evaluation at position x_index mod arity
// (random_access_extension); compute_evaluation: the coset's interpolant at beta, one CosetInterpolationGate row
// (builder.interpolate_coset: the points are coset_start g^i, the values evals[rev(i)] — the coset is stored bit-reversed).
void fri_query_fold(Builder &b, const CircuitPack &c, const QueryRoundTargets &r, size_t step, const std::vector<BoolTarget> &bits, const Challenges &ch, FriQuery &q) {
    const unsigned ab = (unsigned)c.arity_bits[step], arity = 1u << ab;
    const std::vector<BoolTarget> within(bits.begin(), bits.begin() + ab);
    const std::vector<Target> &ev = r.step_evals[step];
    std::vector<Target> c0(arity), c1(arity);
    for (unsigned i = 0; i < arity; i++) { c0[i] = ev[2 * i]; c1[i] = ev[2 * i + 1]; }
    const Target within_t = b.le_sum(within);
    b.connect(b.random_access(within_t, c0), q.old_eval.t.t[0]);
    b.connect(b.random_access(within_t, c1), q.old_eval.t.t[1]);
    // coset_start = subgroup_x * g^(-rev(within)); the coset's values in natural order of its points are evals[rev(i)]
    const u64 g_inv = gl::canon(gl::inv(gl::root_of_unity(ab)));
    const std::vector<BoolTarget> rev_within(within.rbegin(), within.rend());
    const Target coset_start = b.mul(q.subgroup_x, b.exp_from_bits_const_base(g_inv, rev_within));
    std::vector<cb::ExtTarget> ys(arity);
    for (unsigned i = 0; i < arity; i++) {
        unsigned src = 0;
        for (unsigned k = 0; k < ab; k++) src |= ((i >> k) & 1u) << (ab - 1 - k);
        ys[i] = {{ev[2 * src], ev[2 * src + 1]}};
    }
    q.old_eval = XT{b.interpolate_coset(ab, coset_start, ys, ch.fri_betas[step])};
    for (unsigned k = 0; k < ab; k++) q.subgroup_x = b.mul(q.subgroup_x, q.subgroup_x);
}

// the final polynomial at the last subgroup point equals the last folded evaluation
void fri_query_end(Builder &b, const ProofTargets &p, const FriQuery &q) {
    const XT sx = cbx::lift(q.subgroup_x);
    std::vector<cb::ExtTarget> coeffs;
    for (size_t i = 0; i < p.final_poly.size() / 2; i++) coeffs.push_back({{p.final_poly[2 * i], p.final_poly[2 * i + 1]}});
    b.connect_ext(b.reduce_ext(sx.t, coeffs), q.old_eval.t);
}

}  // namespace

extern "C" {

int qpgpu_wrapper_circuit_build(const uint64_t *inner_pack, size_t inner_words, const uint64_t *inner_cs_cap, size_t cap_words, unsigned num_proofs,
                                unsigned num_routed_wires, unsigned min_degree_bits, int inner_hasher, unsigned flags, uint64_t *pack_out, size_t pack_cap_words,
                                size_t *pack_words, uint64_t *target_map_out, size_t map_cap, size_t *map_count, uint64_t *info_out, char *err) {
    auto fail = [&](int code, const std::string &m) { if (err) std::snprintf(err, QPGPU_BATCH_ERR_CAP, "%s", m.c_str()); return code; };
    if (err) err[0] = 0;
    if (!inner_pack || !inner_cs_cap || !pack_words || num_proofs == 0 || num_proofs > 64) return fail(QPGPU_EINVAL, "wrapper_circuit_build: null argument or proof count outside 1..64");
    if (flags & ~(QPGPU_WRAPPER_TRANSCRIPT | QPGPU_WRAPPER_PRIVATE_BATCH | QPGPU_WRAPPER_PUBLIC_BATCH | QPGPU_WRAPPER_VERIFY | QPGPU_WRAPPER_ZERO_KNOWLEDGE)) return fail(QPGPU_EINVAL, "wrapper_circuit_build: unknown flag");
    const bool transcript = (flags & QPGPU_WRAPPER_TRANSCRIPT) != 0, private_batch = (flags & QPGPU_WRAPPER_PRIVATE_BATCH) != 0, public_batch = (flags & QPGPU_WRAPPER_PUBLIC_BATCH) != 0;
    const bool verify = (flags & QPGPU_WRAPPER_VERIFY) != 0;
    if (verify && !transcript) return fail(QPGPU_EINVAL, "wrapper_circuit_build: QPGPU_WRAPPER_VERIFY needs the in-circuit transcript (its challenges are what the arithmetic consumes)");
    if (private_batch && public_batch) return fail(QPGPU_EINVAL, "wrapper_circuit_build: a circuit is the private-batch or the public-batch layer, not both");
    CircuitPack inner;
    { const std::string why = inner.parse(inner_pack, inner_words); if (!why.empty()) return fail(QPGPU_EINVAL, "wrapper_circuit_build: inner pack: " + why); }
    if (cap_words != ((size_t)4 << inner.cap_height)) return fail(QPGPU_EINVAL, "wrapper_circuit_build: the inner constants/sigmas cap has the wrong size");
    // the shape checks of PrivateBatchCircuit::new / PublicBatchCircuit::new (circuit_logic.rs:94-104 / :74-87), with their messages
    size_t n_leaf_inner = 0;
    if (private_batch && inner.num_public_inputs != LEAF_PI_LEN)
        return fail(QPGPU_EINVAL, "leaf_common.num_public_inputs (" + std::to_string(inner.num_public_inputs) + ") != expected wormhole leaf PI len (21); refusing to build a private-batch circuit over a non-leaf-shaped inner circuit");
    if (public_batch) {
        if (inner.num_public_inputs < 8 + LEAF_PI_LEN || (inner.num_public_inputs - 8) % LEAF_PI_LEN || (inner.num_public_inputs - 8) / LEAF_PI_LEN > 64)
            return fail(QPGPU_EINVAL, "private_batch_common.num_public_inputs (" + std::to_string(inner.num_public_inputs) + ") is not a private-batch PI len 21 N + 8 for an N in 1..=64");
        n_leaf_inner = (size_t)(inner.num_public_inputs - 8) / LEAF_PI_LEN;
    }
    const size_t T = qpgpu_proof_target_count(inner_pack, inner_words), Q = inner.num_query_rounds;
    const size_t total = (size_t)num_proofs * (T + 4 + Q);
    try {
        cb::Config cfg;
        cfg.num_routed_wires = num_routed_wires ? num_routed_wires : 80;
        cfg.min_degree_bits = min_degree_bits;
        cfg.inner_hasher = inner_hasher;
        cfg.zero_knowledge = (flags & QPGPU_WRAPPER_ZERO_KNOWLEDGE) != 0;
        Builder b(cfg);
        struct Bind { explicit Bind(Builder *p) { cbx::g_b = p; } ~Bind() { cbx::g_b = nullptr; } } bind(&b);     // XT's operations build into `b` while it lives
        const unsigned L = (unsigned)(inner.degree_bits + inner.rate_bits), cap_h = (unsigned)inner.cap_height;
        // verifier data of the inner circuit: constants of this one (builder.constant_merkle_cap)
        std::vector<HashOutTarget> cs_cap((size_t)1 << cap_h);
        for (size_t i = 0; i < cs_cap.size(); i++) for (int k = 0; k < 4; k++) cs_cap[i].elements[k] = b.constant(inner_cs_cap[4 * i + k]);
        std::vector<ProofTargets> proofs;
        std::vector<std::vector<Target>> preimages, x_indices;
        for (unsigned i = 0; i < num_proofs; i++) {
            proofs.push_back(add_virtual_proof(b, inner));
            if (proofs.back().all.size() != T) return fail(QPGPU_EINVAL, "wrapper_circuit_build: proof target count disagrees with qpgpu_proof_target_count");
        }
        for (unsigned i = 0; i < num_proofs; i++) preimages.push_back(b.add_virtual_targets(4));      // the dummy-nullifier preimages: assigned by fill_private_batch_witness, used by the batch logic only
        for (unsigned i = 0; i < num_proofs; i++) x_indices.push_back(transcript ? std::vector<Target>() : b.add_virtual_targets(Q));
        size_t rows_hash = 0;
        for (unsigned i = 0; i < num_proofs; i++) {
            const ProofTargets &p = proofs[i];
            if (!private_batch && !public_batch) for (Target t : p.public_inputs) b.register_public_input(t);                // forwarded
            const HashOutTarget pih = b.hash_n_to_hash_no_pad(p.public_inputs);         // verify_proof's public_inputs_hash
            Challenges chal;
            if (transcript) {
                // get_challenges (plonk/get_challenges.rs) + fri_challenges: the prover's transcript replayed in-circuit, in the order
                // of csrc/verifier.cpp; the Plonk and FRI challenges themselves are not consumed yet (nothing in-circuit evaluates the
                // openings or folds), the proof-of-work response and the query indices are
                RecursiveChallenger ch(b);
                for (int k = 0; k < 4; k++) ch.observe(b.constant(inner.circuit_digest[k]));
                ch.observe(pih);
                ch.observe_cap(p.caps[0]);
                for (uint64_t k = 0; k < inner.num_challenges; k++) chal.betas.push_back(ch.get());
                for (uint64_t k = 0; k < inner.num_challenges; k++) chal.gammas.push_back(ch.get());
                ch.observe_cap(p.caps[1]);
                for (uint64_t k = 0; k < inner.num_challenges; k++) chal.alphas.push_back(ch.get());
                ch.observe_cap(p.caps[2]);
                chal.zeta = ch.get_ext();
                ch.observe(p.openings);
                chal.fri_alpha = ch.get_ext();
                for (size_t s = 0; s < inner.arity_bits.size(); s++) { ch.observe_cap(p.commit_caps[s]); chal.fri_betas.push_back(ch.get_ext()); }
                ch.observe(p.final_poly);
                ch.observe(p.pow_witness);
                const Target pow_response = ch.get();
                // fri_verify_proof_of_work: the response has proof_of_work_bits leading zeros as a 64-bit integer
                if (inner.proof_of_work_bits) b.range_check(pow_response, 64 - (unsigned)inner.proof_of_work_bits);
                x_indices[i].resize(Q);
                for (size_t q = 0; q < Q; q++) x_indices[i][q] = ch.get();
            }
            FriPre fri;
            if (verify) {
                const std::string why = verify_openings(b, inner, p, pih, chal, fri);
                if (!why.empty()) return fail(QPGPU_EINVAL, "wrapper_circuit_build: " + why);
            }
            for (size_t q = 0; q < Q; q++) {
                const QueryRoundTargets &r = p.rounds[q];
                // low_bits(x_index, n_log, F::BITS): the low L bits of a full 64-bit split of the challenge (or, without the transcript,
                // of an index handed in as it is)
                std::vector<BoolTarget> bits = b.split_le(x_indices[i][q], transcript ? 64 : L);
                bits.resize(L);
                const Target cap_index = b.le_sum(std::vector<BoolTarget>(bits.end() - cap_h, bits.end()));
                const std::vector<HashOutTarget> *caps[4] = {&cs_cap, &p.caps[0], &p.caps[1], &p.caps[2]};
                for (int o = 0; o < 4; o++) b.verify_merkle_proof_to_cap_with_cap_index(r.evals[o], bits, cap_index, *caps[o], r.siblings[o]);
                FriQuery fq;
                if (verify) fq = fri_query_begin(b, inner, r, bits, chal, fri);
                for (size_t s = 0; s < inner.arity_bits.size(); s++) {
                    if (verify) fri_query_fold(b, inner, r, s, bits, chal, fq);                    // (before the step's index bits are dropped)
                    bits.erase(bits.begin(), bits.begin() + (long)inner.arity_bits[s]);        // coset_index_bits
                    b.verify_merkle_proof_to_cap_with_cap_index(r.step_evals[s], bits, cap_index, p.commit_caps[s], r.step_siblings[s]);
                }
                if (verify) fri_query_end(b, p, fq);
            }
            rows_hash = b.num_gates();
        }
        if (private_batch || public_batch) {
            std::vector<std::vector<Target>> pis;
            for (const ProofTargets &p : proofs) pis.push_back(p.public_inputs);
            // public batch: the aggregator address takes the first four "preimage" targets
            const std::vector<Target> outs = private_batch ? private_batch_logic(b, pis, preimages) : public_batch_logic(b, pis, n_leaf_inner, preimages[0]);
            for (Target t : outs) b.register_public_input(t);
        }
        CircuitPack pack;
        const std::string why = b.build(pack);
        if (!why.empty()) return fail(QPGPU_EINVAL, "wrapper_circuit_build: " + why);
        const std::vector<uint64_t> words = pack.serialize();
        *pack_words = words.size();
        if (pack_out) {
            if (pack_cap_words < words.size()) return fail(QPGPU_EBUFSIZE, "wrapper_circuit_build: pack buffer too small");
            std::memcpy(pack_out, words.data(), words.size() * 8);
        }
        const std::vector<u64> &blind = b.blinding_cells();
        if (map_count) *map_count = total + blind.size();
        if (target_map_out) {
            if (map_cap < total + blind.size()) return fail(QPGPU_EBUFSIZE, "wrapper_circuit_build: target map buffer too small");
            if (!blind.empty()) std::memcpy(target_map_out + total, blind.data(), blind.size() * 8);             // the blinding rows' random wires: cells as they are
            auto cell = [&](Target t) { const u64 c = b.cell_of(t); return c == cb::NO_CELL ? UINT64_MAX : c; };
            size_t k = 0;
            for (unsigned i = 0; i < num_proofs; i++) for (Target t : proofs[i].all) target_map_out[k++] = cell(t);
            for (unsigned i = 0; i < num_proofs; i++) for (Target t : preimages[i]) target_map_out[k++] = cell(t);
            for (unsigned i = 0; i < num_proofs; i++) for (size_t q = 0; q < Q; q++) target_map_out[k++] = transcript ? UINT64_MAX : cell(x_indices[i][q]);   // (derived in-circuit: nothing to assign)
        }
        if (info_out) {
            const std::map<uint64_t, size_t> gc = b.gate_counts();
            auto cnt = [&](uint64_t t) { auto it = gc.find(t); return it == gc.end() ? (uint64_t)0 : (uint64_t)it->second; };
            const uint64_t info[QPGPU_WRAPPER_CIRCUIT_INFO_WORDS] = {pack.degree_bits, b.rows_before_padding(), (uint64_t)T, (uint64_t)Q, cnt(GATE_POSEIDON), cnt(GATE_RANDOM_ACCESS),
                                                                    cnt(GATE_BASE_SUM), cnt(GATE_ARITHMETIC), cnt(GATE_CONSTANT), pack.num_public_inputs, (uint64_t)rows_hash, (uint64_t)b.blinding_rows()};
            std::memcpy(info_out, info, sizeof info);
        }
    } catch (const std::exception &e) {
        return fail(QPGPU_EINVAL, std::string("wrapper_circuit_build: ") + e.what());
    }
    return QPGPU_OK;
}
}  // extern "C"
