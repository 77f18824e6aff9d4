// synth_api.cpp — C ABI of the synthetic circuit generator (host only; no GPU needed).
#include <cstring>
#include "circuit.hpp"
#include "ctx.hpp"
#include "poseidon.hpp"

std::string synth_build(unsigned degree_bits, unsigned num_wires, unsigned num_routed, unsigned num_public_inputs,
                        uint64_t seed, unsigned flags, CircuitPack &pack, std::vector<uint64_t> &wires, std::vector<uint64_t> &pis);

std::string synth_gate_layout(unsigned num_wires, unsigned num_routed, unsigned flags, std::vector<GateInfo> &gates, uint64_t &num_selectors);
struct P2Site { unsigned len, blocks, slot; };
std::vector<P2Site> synth_p2_sites(unsigned degree_bits, unsigned num_public_inputs, unsigned flags);

extern "C" {

// the hash constants the library derives at start-up (host only): 360 round constants, then the FAST_PARTIAL tables
size_t qpgpu_poseidon_constants(uint64_t *round_constants_360, uint64_t *fast_partial, size_t fast_partial_cap) {
    if (round_constants_360) std::memcpy(round_constants_360, poseidon::host_round_constants(), 360 * 8);
    if (fast_partial && fast_partial_cap >= (size_t)poseidon::FP_WORDS) std::memcpy(fast_partial, poseidon::host_fast_partial(), poseidon::FP_WORDS * 8);
    return poseidon::FP_WORDS;
}

size_t qpgpu_synth_pack_words_ex(unsigned degree_bits, unsigned num_wires, unsigned num_routed, unsigned flags) {
    std::vector<GateInfo> gates;
    uint64_t sels = 0;
    if (num_routed < 8 || !synth_gate_layout(num_wires, num_routed, flags, gates, sels).empty()) return 0;
    CircuitPack p;
    p.degree_bits = degree_bits; p.num_routed_wires = num_routed; p.num_selectors = sels; p.num_constants = 2;
    size_t arity = fri_reduction_arity_bits(degree_bits, 3, 4, 4, 5).size();
    // with hints (flag bit 4) the trailer's length depends on the seed: an upper bound is returned and
    // qpgpu_synth_circuit_ex reports the exact count
    const size_t hint_cap = (flags & 16) ? 2 + 8 * (((size_t)num_routed / 2 + 2) << degree_bits) : 0;
    // with Poseidon rows (flag bit 0) the pack carries the public-input cell trailer: room for 4094 public inputs here; for
    // more, qpgpu_synth_circuit_ex with pack_out = NULL reports the exact size
    const size_t pubi_cap = (flags & 1) ? 4096 : 0;
    const size_t p2_cap = (flags & 64) ? 2 + P2GateLayout::WORDS : 0;
    return 18 + arity + gates.size() * 8 + num_routed + 4 + ((size_t)p.num_cs_cols() << degree_bits) + hint_cap + pubi_cap + p2_cap;
}
size_t qpgpu_synth_pack_words(unsigned degree_bits, unsigned num_wires, unsigned num_routed) {
    return qpgpu_synth_pack_words_ex(degree_bits, num_wires, num_routed, 0);
}

// the Poseidon2 hash sites of a synthetic circuit built with flag bit 6: 3 words each (preimage length, gate rows, first slot);
// returns how many there are (writes at most cap / 3 of them)
size_t qpgpu_synth_p2_sites(unsigned degree_bits, unsigned num_public_inputs, unsigned flags, uint64_t *out, size_t cap_words) {
    const std::vector<P2Site> sites = synth_p2_sites(degree_bits, num_public_inputs, flags);
    for (size_t i = 0; i < sites.size() && 3 * i + 3 <= cap_words && out; i++) { out[3 * i] = sites[i].len; out[3 * i + 1] = sites[i].blocks; out[3 * i + 2] = sites[i].slot; }
    return sites.size();
}

int qpgpu_synth_circuit_ex(unsigned degree_bits, unsigned num_wires, unsigned num_routed, unsigned num_public_inputs,
                           uint64_t seed, unsigned flags, uint64_t *pack_out, size_t pack_cap_words, size_t *pack_words,
                           uint64_t *wires_out, uint64_t *pis_out);
int qpgpu_synth_circuit(unsigned degree_bits, unsigned num_wires, unsigned num_routed, unsigned num_public_inputs,
                        uint64_t seed, uint64_t *pack_out, size_t pack_cap_words, size_t *pack_words,
                        uint64_t *wires_out, uint64_t *pis_out) {
    return qpgpu_synth_circuit_ex(degree_bits, num_wires, num_routed, num_public_inputs, seed, 0, pack_out, pack_cap_words,
                                  pack_words, wires_out, pis_out);
}
int qpgpu_synth_circuit_ex(unsigned degree_bits, unsigned num_wires, unsigned num_routed, unsigned num_public_inputs,
                           uint64_t seed, unsigned flags, uint64_t *pack_out, size_t pack_cap_words, size_t *pack_words,
                           uint64_t *wires_out, uint64_t *pis_out) {
    CircuitPack pack;
    std::vector<uint64_t> wires, pis;
    std::string err = synth_build(degree_bits, num_wires, num_routed, num_public_inputs, seed, flags, pack, wires, pis);
    if (!err.empty()) return QPGPU_EINVAL;
    std::vector<uint64_t> words = pack.serialize();
    if (pack_words) *pack_words = words.size();
    if (pack_out) {
        if (pack_cap_words < words.size()) return QPGPU_EBUFSIZE;
        std::memcpy(pack_out, words.data(), words.size() * 8);
    }
    if (wires_out) std::memcpy(wires_out, wires.data(), wires.size() * 8);
    if (pis_out && !pis.empty()) std::memcpy(pis_out, pis.data(), pis.size() * 8);
    return QPGPU_OK;
}

}  // extern "C"
