// wrapper_circuit.cpp — a recursive wrapper circuit that really checks part of its inner proofs:
// the Merkle half of plonky2's in-circuit verifier, restated on the native builder (builder.hpp).
//
//   add_recursive_verifiers                      wormhole/aggregator/src/common/recursive.rs:74-102
//     builder.add_virtual_proof_with_pis(common)   one virtual target per field element of an inner proof (proof_targets.cpp's order)
//     builder.verify_proof::<C>(proof, vd, common) qp-plonky2 1.5.5, un-vendored; restated from upstream plonky2:
//       public_inputs_hash = hash_n_to_hash_no_pad(public_inputs)                               plonk/recursive_verifier.rs
//       fri_verifier_query_round: x_index bits, cap_index = le_sum(high bits),                  fri/recursive_verifier.rs
//         fri_verify_initial_proof: verify_merkle_proof_to_cap_with_cap_index per oracle        hash/merkle_proofs.rs
//         per reduction step: verify_merkle_proof_to_cap_with_cap_index(flatten(evals), coset_index_bits, ..)
//
// WHAT IS VERIFIED: for every inner proof and every one of its query rounds, the four opened rows (constants/sigmas, wires,
// Z / partial products, quotient) and every FRI step's coset of evaluations are hashed in-circuit (PoseidonGate rows) and
// walked up their Merkle paths (one PoseidonGate row per level, swap = the index bit) to the cap entry the index selects
// (RandomAccessGate rows): against the inner CIRCUIT's constants/sigmas cap (verifier data, constants of this circuit) and
// against the caps the proof itself carries. With QPGPU_WRAPPER_TRANSCRIPT the Fiat-Shamir transcript is replayed in-circuit
// too (RecursiveChallenger: circuit digest, public-input hash, caps, all openings, FRI caps, final polynomial, proof-of-work
// witness), the proof-of-work response is range-checked and the 28 query indices are the low bits of the transcript's
// challenges — so the indices cannot be chosen, and everything the transcript absorbs is bound to the rows that are opened.
// The inner public inputs are forwarded as this circuit's public inputs. A byte flipped anywhere in an inner proof except in
// places only the missing arithmetic looks at makes the witness unsatisfiable ("set twice with different values").
// WHAT IS NOT: the openings against the vanishing polynomial at zeta (gate constraints, permutation argument) and the
// reduced-opening / folding arithmetic that ties the opened rows to the FRI evaluations and to the final polynomial — the
// Plonk and FRI challenges are derived in-circuit but not consumed yet. A wrapper proof therefore attests "a proof-shaped
// object with a valid proof of work whose transcript-chosen rows and cosets are committed under its caps", not yet "the inner
// proofs verify". The wrapper-specific logic of the private / public batch (circuit_logic.rs) is not part of it either.
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include "../../include/qpgpu.h"
#include "../../include/qpgpu_batch.h"
#include "builder.hpp"
#include "poseidon.hpp"

using cb::BoolTarget;
using cb::Builder;
using cb::HashOutTarget;
using cb::Target;
using gl::u64;

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
    Target get() {
        absorb();
        if (out.empty()) { state = b.permute_swapped(state, b._false()); out.assign(state.begin(), state.begin() + 8); }
        const Target t = out.back();
        out.pop_back();
        return t;
    }
};

}  // namespace

extern "C" {

int qpgpu_wrapper_circuit_build(const uint64_t *inner_pack, size_t inner_words, const uint64_t *inner_cs_cap, size_t cap_words, unsigned num_proofs,
                                unsigned num_routed_wires, unsigned min_degree_bits, int inner_hasher, unsigned flags, uint64_t *pack_out, size_t pack_cap_words,
                                size_t *pack_words, uint64_t *target_map_out, size_t map_cap, size_t *map_count, uint64_t *info_out, char *err) {
    auto fail = [&](int code, const std::string &m) { if (err) std::snprintf(err, QPGPU_BATCH_ERR_CAP, "%s", m.c_str()); return code; };
    if (err) err[0] = 0;
    if (!inner_pack || !inner_cs_cap || !pack_words || num_proofs == 0 || num_proofs > 64) return fail(QPGPU_EINVAL, "wrapper_circuit_build: null argument or proof count outside 1..64");
    if (flags & ~QPGPU_WRAPPER_TRANSCRIPT) return fail(QPGPU_EINVAL, "wrapper_circuit_build: unknown flag");
    const bool transcript = (flags & QPGPU_WRAPPER_TRANSCRIPT) != 0;
    CircuitPack inner;
    { const std::string why = inner.parse(inner_pack, inner_words); if (!why.empty()) return fail(QPGPU_EINVAL, "wrapper_circuit_build: inner pack: " + why); }
    if (cap_words != ((size_t)4 << inner.cap_height)) return fail(QPGPU_EINVAL, "wrapper_circuit_build: the inner constants/sigmas cap has the wrong size");
    const size_t T = qpgpu_proof_target_count(inner_pack, inner_words), Q = inner.num_query_rounds;
    const size_t total = (size_t)num_proofs * (T + 4 + Q);
    if (map_count) *map_count = total;
    try {
        cb::Config cfg;
        cfg.num_routed_wires = num_routed_wires ? num_routed_wires : 80;
        cfg.min_degree_bits = min_degree_bits;
        cfg.inner_hasher = inner_hasher;
        Builder b(cfg);
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
            for (Target t : p.public_inputs) b.register_public_input(t);                // forwarded
            const HashOutTarget pih = b.hash_n_to_hash_no_pad(p.public_inputs);         // verify_proof's public_inputs_hash
            if (transcript) {
                // get_challenges (plonk/get_challenges.rs) + fri_challenges: the prover's transcript replayed in-circuit, in the order
                // of csrc/verifier.cpp; the Plonk and FRI challenges themselves are not consumed yet (nothing in-circuit evaluates the
                // openings or folds), the proof-of-work response and the query indices are
                RecursiveChallenger ch(b);
                for (int k = 0; k < 4; k++) ch.observe(b.constant(inner.circuit_digest[k]));
                ch.observe(pih);
                ch.observe_cap(p.caps[0]);
                for (uint64_t k = 0; k < 2 * inner.num_challenges; k++) (void)ch.get();       // betas, gammas
                ch.observe_cap(p.caps[1]);
                for (uint64_t k = 0; k < inner.num_challenges; k++) (void)ch.get();           // alphas
                ch.observe_cap(p.caps[2]);
                (void)ch.get(); (void)ch.get();                                               // zeta
                ch.observe(p.openings);
                (void)ch.get(); (void)ch.get();                                               // FRI alpha
                for (size_t s = 0; s < inner.arity_bits.size(); s++) { ch.observe_cap(p.commit_caps[s]); (void)ch.get(); (void)ch.get(); }   // FRI betas
                ch.observe(p.final_poly);
                ch.observe(p.pow_witness);
                const Target pow_response = ch.get();
                // fri_verify_proof_of_work: the response has proof_of_work_bits leading zeros as a 64-bit integer
                if (inner.proof_of_work_bits) b.range_check(pow_response, 64 - (unsigned)inner.proof_of_work_bits);
                x_indices[i].resize(Q);
                for (size_t q = 0; q < Q; q++) x_indices[i][q] = ch.get();
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
                for (size_t s = 0; s < inner.arity_bits.size(); s++) {
                    bits.erase(bits.begin(), bits.begin() + (long)inner.arity_bits[s]);        // coset_index_bits
                    b.verify_merkle_proof_to_cap_with_cap_index(r.step_evals[s], bits, cap_index, p.commit_caps[s], r.step_siblings[s]);
                }
            }
            rows_hash = b.num_gates();
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
        if (target_map_out) {
            if (map_cap < total) return fail(QPGPU_EBUFSIZE, "wrapper_circuit_build: target map buffer too small");
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
                                                                    cnt(GATE_BASE_SUM), cnt(GATE_ARITHMETIC), cnt(GATE_CONSTANT), pack.num_public_inputs, (uint64_t)rows_hash, 0};
            std::memcpy(info_out, info, sizeof info);
        }
    } catch (const std::exception &e) {
        return fail(QPGPU_EINVAL, std::string("wrapper_circuit_build: ") + e.what());
    }
    return QPGPU_OK;
}

}  // extern "C"
