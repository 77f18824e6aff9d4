// circuit.hpp — the "circuit pack": everything the prover needs that the Rust CircuitBuilder produces
// (SURVEY.md §8f-1). It is the serialised form of plonky2's CommonCircuitData + ProverOnlyCircuitData
// restricted to what `prove` reads after witness generation: sizes and FRI parameters, the gate list with
// selector groups, k_is, circuit_digest, and the constants/sigmas polynomials in value form.
//
// Format "QPCP1": little-endian u64 words.
//   [0]  magic 0x0000003150435051 ("QPCP1")
//   [1..17] degree_bits, num_wires, num_routed_wires, num_constants, num_selectors, num_challenges,
//           quotient_degree_factor, num_partial_products, num_public_inputs, rate_bits, cap_height,
//           proof_of_work_bits, num_query_rounds, zero_knowledge, num_gate_constraints, num_gates,
//           num_arity_rounds
//   arity_bits[num_arity_rounds]
//   gates[num_gates] x 8 words: type, param0, param1, selector_index, group_start, group_end,
//           num_constraints, param2
//           (type, param0, param1, param2): 0 Noop; 1 Constant(num_consts); 2 PublicInput; 3 Arithmetic(num_ops); 4 Poseidon;
//           5 BaseSum(num_limbs, base 2); 6 ArithmeticExtension(num_ops); 7 MulExtension(num_ops); 8 Reducing(num_coeffs);
//           9 ReducingExtension(num_coeffs); 10 RandomAccess(bits, num_copies, num_extra_constants);
//           11 Exponentiation(num_power_bits); 12 PoseidonMds; 13 CosetInterpolation(subgroup_bits, degree);
//           14 Poseidon2 (the qp fork's gate behind `hash_n_to_hash_no_pad_p2`; wire layout in the trailer "P2GL1" below)
//   k_is[num_routed_wires]; circuit_digest[4]
//   constants_sigmas values: (num_selectors + num_constants + num_routed_wires) columns x 2^degree_bits,
//           column-major, natural subgroup order (column order = plonky2's constants_sigmas oracle:
//           selectors, constants, sigmas)
//   optional trailer: witness hints = the generators that are not attached to a gate (stage s1 only; the prover
//           ignores them): magic 0x00000031544E4948 ("HINT1"), count, count x 8 words {opcode, a, b, c, d, e, f, 0}.
//           A cell is row * num_wires + column and must be a routed wire. Opcodes (plonky2 generator, arguments):
//           1 CopyGenerator (dst a <- src b); 2 EqualityGenerator (x a, y b -> equal c, inv d);
//           3 WireSplitGenerator, one BaseSum gate (integer a -> sum wire b = (a >> c) & (2^d - 1));
//           4 QuotientGeneratorExtension (numerator a,b / denominator c,d -> quotient e,f);
//           5 ConstantGenerator (a <- value b); 6 NonzeroTestGenerator (x a -> b = x == 0 ? 1 : 1/x);
//           7 LowHighGenerator (integer a -> low b, high c, split at bit d)
//   optional trailer after it: public-input cells = where `prover_only.public_inputs` (the registered targets) sit in
//           the trace: magic 0x0000003149425550 ("PUBI1"), count (= num_public_inputs), count cells. Stage s1 writes the
//           caller's public inputs there (PartialWitness::set_target for each public-input target); in plonky2 those
//           cells feed the PoseidonGate rows whose output is copy-connected to the PublicInputGate's wires.
//   optional trailer, anywhere among the others: the wire layout of the Poseidon2 gate (type 14): magic 0x000000314C473250
//           ("P2GL1"), count (= 10), then: first input wire, first output wire, swap wire (0xFFFFFFFF: the gate has no swap /
//           delta wires and no constraints for them), first delta wire, first S-box-input wire of the first half's full rounds,
//           first S-box-input wire of the 22 partial rounds, first S-box-input wire of the second half's four full rounds,
//           whether round 0 of the first half carries S-box-input wires too (0: its S-box inputs are linear in the gate's
//           inputs and are not recorded, as in upstream's PoseidonGate; 1: all four rounds are recorded), constraint order (0:
//           swap boolean, deltas, first-half rounds, partial rounds, second-half rounds, outputs), and one past the last wire
//           the gate uses. The fork's gate lives in un-vendored qp-plonky2 1.5.5 and its layout cannot be read offline: a pack
//           that selects gate type 14 WITHOUT the trailer is REFUSED (validate()): no layout is assumed. The synthetic generator
//           and the native builder write the trailer explicitly with the stand-in layout = upstream PoseidonGate's carried over
//           to Poseidon2 (135 wires, 123 constraints of degree 7, consistent with reference common/src/circuit.rs:428-431,447-449);
//           the Rust exporter (integration/qpgpu_backend.rs) fills it from the fork's gate. LAYOUT UNPINNED until then.
#pragma once
#include <stdint.h>
#include <cstring>
#include <string>
#include <vector>

enum GateType : uint64_t { GATE_NOOP = 0, GATE_CONSTANT = 1, GATE_PUBLIC_INPUT = 2, GATE_ARITHMETIC = 3, GATE_POSEIDON = 4, GATE_BASE_SUM = 5, GATE_ARITHMETIC_EXT = 6, GATE_MUL_EXT = 7,
                           GATE_REDUCING = 8, GATE_REDUCING_EXT = 9, GATE_RANDOM_ACCESS = 10, GATE_EXPONENTIATION = 11, GATE_POSEIDON_MDS = 12, GATE_COSET_INTERPOLATION = 13, GATE_POSEIDON2 = 14 };

struct GateInfo {
    uint64_t type, param0, param1, selector_index, group_start, group_end, num_constraints, param2;
};

enum HintOpcode : uint64_t { HINT_COPY = 1, HINT_EQUALITY = 2, HINT_WIRE_SPLIT = 3, HINT_QUOTIENT_EXT = 4, HINT_CONSTANT = 5,
                             HINT_NONZERO_TEST = 6, HINT_LOW_HIGH = 7 };
struct HintOp { uint64_t w[8]; };
constexpr uint64_t QPCP_HINT_MAGIC = 0x00000031544E4948ull;
constexpr uint64_t QPCP_PUBI_MAGIC = 0x0000003149425550ull;   // "PUBI1"
constexpr uint64_t QPCP_P2GL_MAGIC = 0x000000314C473250ull;   // "P2GL1"

// Wire layout of the Poseidon2 gate (see the format notes above). Plain 32-bit fields: handed to kernels by value.
struct P2GateLayout {
    uint32_t w_input = 0, w_output = 12, w_swap = 24, w_delta = 25, w_full0 = 29, w_partial = 65, w_full1 = 87;
    uint32_t first_round_wires = 0, constraint_order = 0, end_wire = 135;
    static constexpr uint32_t NO_SWAP = 0xFFFFFFFFu;
    static constexpr int WORDS = 10;
#if defined(__HIPCC__)
    __host__ __device__
#endif
    bool has_swap() const { return w_swap != NO_SWAP; }
#if defined(__HIPCC__)
    __host__ __device__
#endif
    uint32_t full0_rounds() const { return first_round_wires ? 4u : 3u; }
    uint32_t num_constraints() const { return (has_swap() ? 5u : 0u) + 12u * full0_rounds() + 22u + 48u + 12u; }
    bool is_default() const { P2GateLayout d; return std::memcmp(this, &d, sizeof d) == 0; }
    std::string validate(uint64_t num_wires, uint64_t num_routed) const;
};

struct CircuitPack {
    uint64_t degree_bits = 0, num_wires = 0, num_routed_wires = 0, num_constants = 0, num_selectors = 0,
             num_challenges = 0, quotient_degree_factor = 0, num_partial_products = 0, num_public_inputs = 0,
             rate_bits = 0, cap_height = 0, proof_of_work_bits = 0, num_query_rounds = 0, zero_knowledge = 0,
             num_gate_constraints = 0;
    std::vector<uint64_t> arity_bits;
    std::vector<GateInfo> gates;
    std::vector<uint64_t> k_is;
    uint64_t circuit_digest[4] = {0, 0, 0, 0};
    std::vector<uint64_t> constants_sigmas;  // column-major values
    std::vector<HintOp> hints;               // optional: free-standing witness generators (stage s1)
    std::vector<uint64_t> pi_cells;          // optional: the wire cell (row * num_wires + column) of every public input
    P2GateLayout p2_layout;                  // wire layout of the Poseidon2 gate (default unless the pack carries "P2GL1")
    bool has_p2_layout = false;              // the trailer was present (serialize() writes it back)

    uint64_t n() const { return 1ull << degree_bits; }
    uint64_t num_cs_cols() const { return num_selectors + num_constants + num_routed_wires; }
    uint64_t num_chunks() const { return num_partial_products + 1; }
    uint64_t num_zs_pp_cols() const { return num_challenges * (1 + num_partial_products); }
    uint64_t num_quotient_cols() const { return num_challenges * quotient_degree_factor; }

    std::vector<uint64_t> serialize() const;
    // returns empty string on success, otherwise the reason
    std::string parse(const uint64_t *words, size_t n_words);
    std::string validate() const;
};

constexpr uint64_t QPCP_MAGIC = 0x0000003150435051ull;

// FriParams::reduction_arity_bits for ConstantArityBits(arity_bits, final_poly_bits)
std::vector<uint64_t> fri_reduction_arity_bits(uint64_t degree_bits, uint64_t rate_bits, uint64_t cap_height,
                                               uint64_t arity_bits, uint64_t final_poly_bits);
