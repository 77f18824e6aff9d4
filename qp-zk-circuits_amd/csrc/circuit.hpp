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
//           11 Exponentiation(num_power_bits); 12 PoseidonMds; 13 CosetInterpolation(subgroup_bits, degree)
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
#pragma once
#include <stdint.h>
#include <string>
#include <vector>

enum GateType : uint64_t { GATE_NOOP = 0, GATE_CONSTANT = 1, GATE_PUBLIC_INPUT = 2, GATE_ARITHMETIC = 3, GATE_POSEIDON = 4, GATE_BASE_SUM = 5, GATE_ARITHMETIC_EXT = 6, GATE_MUL_EXT = 7,
                           GATE_REDUCING = 8, GATE_REDUCING_EXT = 9, GATE_RANDOM_ACCESS = 10, GATE_EXPONENTIATION = 11, GATE_POSEIDON_MDS = 12, GATE_COSET_INTERPOLATION = 13 };

struct GateInfo {
    uint64_t type, param0, param1, selector_index, group_start, group_end, num_constraints, param2;
};

enum HintOpcode : uint64_t { HINT_COPY = 1, HINT_EQUALITY = 2, HINT_WIRE_SPLIT = 3, HINT_QUOTIENT_EXT = 4, HINT_CONSTANT = 5,
                             HINT_NONZERO_TEST = 6, HINT_LOW_HIGH = 7 };
struct HintOp { uint64_t w[8]; };
constexpr uint64_t QPCP_HINT_MAGIC = 0x00000031544E4948ull;
constexpr uint64_t QPCP_PUBI_MAGIC = 0x0000003149425550ull;   // "PUBI1"

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
