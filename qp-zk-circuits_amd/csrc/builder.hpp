// builder.hpp — a circuit builder in the image of plonky2's `CircuitBuilder<F, D>`, for restating the reference's circuit
// definitions natively: the reference's circuits are Rust programs over that builder
// (wormhole/circuit/src/circuit.rs:115-152 and the fragments it wires), the builder itself lives in un-vendored qp-plonky2,
// and this image has no Rust toolchain to run either. What is restated here is the builder's *observable* behaviour as
// upstream plonky2 has it — virtual targets, copy constraints as a union-find over targets, `arithmetic` with its special
// cases and memoised operations, operations packed into ArithmeticGate rows per constant pair, `split_le` on BaseSumGate<2>
// rows, `is_equal` with its EqualityGenerator, random access, sponges and Merkle verification, extension-field arithmetic on
// ArithmeticExtensionGate rows with its QuotientGeneratorExtension, Reducing / ReducingExtension / CosetInterpolation / PoseidonMds
// gate gadgets (what the recursive verifier consists of), constants gathered into ConstantGate rows at build time, the public-input
// hash wired into a PublicInputGate, `blind()` for zero-knowledge circuits, Noop padding to a power of two, gates sorted by
// (degree, id) into selector groups —
// and its output is a circuit pack (circuit.hpp) plus the wire cell every target ended up in.
//
// Not byte-reproducible against the fork (row order of gates, the fork's Poseidon2 sponge wiring and its `circuit_digest`
// formula cannot be read offline): a circuit built here proves the same STATEMENT as the reference's, over the same gate
// set, but its verifier data differs from the reference's. DESIGN.md §5 says so wherever a number rests on it.
#pragma once
#include <array>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>
#include "circuit.hpp"
#include "gl64.hpp"

namespace cb {
using gl::u64;

using Target = uint32_t;
constexpr Target NO_TARGET = 0xFFFFFFFFu;
constexpr u64 NO_CELL = ~0ull;
struct BoolTarget { Target target; };
struct HashOutTarget { Target elements[4]; };
struct ExtTarget { Target t[2]; };                  // ExtensionTarget<2>: an element of F[x] / (x^2 - 7)

// CircuitConfig (plonk/circuit_data.rs) + FriConfig: the values of standard_recursion_config unless changed
struct Config {
    unsigned num_wires = 135, num_routed_wires = 80, num_constants = 2, num_challenges = 2, max_quotient_degree_factor = 8;
    unsigned rate_bits = 3, cap_height = 4, proof_of_work_bits = 16, num_query_rounds = 28, arity_bits = 4, final_poly_bits = 5;
    bool zero_knowledge = false;
    int inner_hasher = 0;            // hasher::POSEIDON / hasher::POSEIDON2: the permutation gate the public-input hash is built from
    P2GateLayout p2_layout;          // wire layout of the fork's Poseidon2 gate (LAYOUT UNPINNED, circuit.hpp)
    unsigned min_degree_bits = 0;    // pad with NoopGate rows to at least 2^min_degree_bits rows
    std::string validate() const;
};

class Builder {
public:
    explicit Builder(const Config &cfg);
    const Config &config() const { return cfg_; }

    // ---- targets ----
    Target add_virtual_target();
    std::vector<Target> add_virtual_targets(size_t n);
    HashOutTarget add_virtual_hash();
    BoolTarget add_virtual_bool_target_safe();          // + assert_bool
    BoolTarget add_virtual_bool_target_unsafe() { return {add_virtual_target()}; }
    void register_public_input(Target t) { public_inputs_.push_back(t); }
    Target add_virtual_public_input() { Target t = add_virtual_target(); register_public_input(t); return t; }
    HashOutTarget add_virtual_hash_public_input();
    size_t num_public_inputs() const { return public_inputs_.size(); }

    // ---- constants ----
    Target constant(u64 c);
    Target zero() { return constant(0); }
    Target one() { return constant(1); }
    Target two() { return constant(2); }
    Target neg_one() { return constant(gl::P - 1); }
    BoolTarget constant_bool(bool b) { return {constant(b ? 1 : 0)}; }
    BoolTarget _true() { return constant_bool(true); }
    BoolTarget _false() { return constant_bool(false); }
    bool target_as_constant(Target t, u64 &out) const;

    // ---- copy constraints ----
    void connect(Target a, Target b);
    void connect_hashes(const HashOutTarget &a, const HashOutTarget &b) { for (int i = 0; i < 4; i++) connect(a.elements[i], b.elements[i]); }
    void assert_zero(Target t) { connect(t, zero()); }
    void assert_one(Target t) { connect(t, one()); }
    void assert_bool(BoolTarget b);

    // ---- base-field arithmetic (gadgets/arithmetic.rs) ----
    Target arithmetic(u64 const_0, u64 const_1, Target multiplicand_0, Target multiplicand_1, Target addend);
    Target mul_add(Target x, Target y, Target z) { return arithmetic(1, 1, x, y, z); }
    Target mul_sub(Target x, Target y, Target z) { return arithmetic(1, gl::P - 1, x, y, z); }
    Target add(Target x, Target y) { return arithmetic(1, 1, x, one(), y); }
    Target sub(Target x, Target y) { return arithmetic(1, gl::P - 1, x, one(), y); }
    Target mul(Target x, Target y) { return arithmetic(1, 0, x, y, zero()); }
    Target add_const(Target x, u64 c) { return add(x, constant(c)); }
    Target mul_const(u64 c, Target x) { return mul(constant(c), x); }
    Target mul_const_add(u64 c, Target x, Target y) { return mul_add(constant(c), x, y); }
    BoolTarget not_(BoolTarget b) { return {sub(one(), b.target)}; }
    BoolTarget and_(BoolTarget a, BoolTarget b) { return {mul(a.target, b.target)}; }
    BoolTarget or_(BoolTarget a, BoolTarget b) { const Target t = mul_sub(a.target, b.target, a.target); return {sub(b.target, t)}; }
    Target select(BoolTarget b, Target x, Target y) { const Target tmp = mul_sub(b.target, y, y); return mul_sub(b.target, x, tmp); }
    BoolTarget is_equal(Target x, Target y);

    // ---- range checks / bit decomposition (gadgets/split_base.rs, range_check.rs) ----
    std::vector<BoolTarget> split_le(Target integer, unsigned num_bits);
    void range_check(Target x, unsigned n_log) { (void)split_le(x, n_log); }
    // (low, high) with low < 2^n_log and high < 2^(num_bits - n_log): gadgets/range_check.rs split_low_high
    void split_low_high(Target x, unsigned n_log, unsigned num_bits, Target &low, Target &high);

    // ---- hashing ----
    using State = std::array<Target, 12>;
    State permute_poseidon(const State &in, BoolTarget swap);     // one PoseidonGate row (upstream Poseidon, permute_swapped)
    // every Poseidon2 gate row records the tag current when it was made (a circuit names its hash call sites with it) and its row:
    // the circuit can then tell a witness front-end which cells hold each site's sponge states (see qpgpu_leaf_circuit_hash_hint_cells)
    void set_hash_tag(int tag) { hash_tag_ = tag; }
    const std::vector<std::pair<int, uint32_t>> &poseidon2_rows() const { return p2_rows_; }
    u64 poseidon2_output_cell(uint32_t row, uint32_t i) { return cell_of(wire(row, cfg_.p2_layout.w_output + i)); }
    // further targets a circuit offers a front-end for hinting (values on a serial path that the front-end knows), in the order added
    void add_hint_target(Target t) { hint_targets_.push_back(t); }
    const std::vector<Target> &hint_targets() const { return hint_targets_; }
    State permute_poseidon2(const State &in);                     // one row of the fork's Poseidon2 gate, swap wire (if any) = 0
    // hash_n_to_hash_no_pad::<PoseidonHash>: overwrite-mode sponge, no padding (hashing.rs)
    HashOutTarget hash_n_to_hash_no_pad(const std::vector<Target> &inputs);
    State permute_swapped(const State &in, BoolTarget swap);      // H::permute_swapped for the configured inner hasher
    // hash_or_noop: at most 4 elements are the digest themselves (zero padded), more are hashed (hashing.rs)
    HashOutTarget hash_or_noop(const std::vector<Target> &inputs);
    // gadgets/random_access.rs: v[access_index] on a RandomAccessGate slot (|v| a power of two, at most 2^6)
    Target random_access(Target access_index, const std::vector<Target> &v);
    // gadgets/split_base.rs le_sum: sum of bits[i] * 2^i as a chain of mul_add operations (what upstream picks for <= num_ops bits)
    Target le_sum(const std::vector<BoolTarget> &bits);
    // hash/merkle_proofs.rs verify_merkle_proof_to_cap_with_cap_index: the leaf's digest walks up `siblings` (one permute_swapped
    // per level, swap = the index bit) and must equal cap[cap_index] (4 random accesses into the cap)
    void verify_merkle_proof_to_cap_with_cap_index(const std::vector<Target> &leaf_data, const std::vector<BoolTarget> &leaf_index_bits, Target cap_index,
                                                   const std::vector<HashOutTarget> &cap, const std::vector<HashOutTarget> &siblings);
    // hash_n_to_hash_no_pad_p2::<Poseidon2Hash> of the fork: `input || 1 || 0*` to a multiple of the rate 8, every block ADDED
    // into the rate part (wormhole/circuit/tests/heap_zeroization.rs:133-160 pins the padding; the block-header vectors the
    // additive absorption, tests/test_leaf_witness.py)
    HashOutTarget hash_n_to_hash_no_pad_p2(const std::vector<Target> &inputs);

    // ---- extension-field arithmetic (gadgets/arithmetic_extension.rs) on ArithmeticExtensionGate rows ----
    ExtTarget constant_ext(u64 a, u64 b = 0) { return {{constant(a), constant(b)}}; }
    ExtTarget zero_ext() { return constant_ext(0); }
    ExtTarget one_ext() { return constant_ext(1); }
    ExtTarget to_ext(Target x) { return {{x, zero()}}; }                       // convert_to_ext
    // const_0 * multiplicand_0 * multiplicand_1 + const_1 * addend, with the special cases and the per-(const_0, const_1) slot packing
    ExtTarget arithmetic_ext(u64 const_0, u64 const_1, ExtTarget multiplicand_0, ExtTarget multiplicand_1, ExtTarget addend);
    ExtTarget mul_ext(ExtTarget a, ExtTarget b) { return arithmetic_ext(1, 0, a, b, zero_ext()); }
    ExtTarget mul_add_ext(ExtTarget a, ExtTarget b, ExtTarget c) { return arithmetic_ext(1, 1, a, b, c); }
    ExtTarget mul_sub_ext(ExtTarget a, ExtTarget b, ExtTarget c) { return arithmetic_ext(1, gl::P - 1, a, b, c); }
    ExtTarget add_ext(ExtTarget a, ExtTarget b) { return arithmetic_ext(1, 1, one_ext(), a, b); }
    ExtTarget sub_ext(ExtTarget a, ExtTarget b) { return arithmetic_ext(1, gl::P - 1, one_ext(), a, b); }
    // x / y with its QuotientGeneratorExtension: q is a hint output, q * y is connected to x (div_extension)
    ExtTarget div_ext(ExtTarget x, ExtTarget y);
    void connect_ext(ExtTarget a, ExtTarget b) { connect(a.t[0], b.t[0]); connect(a.t[1], b.t[1]); }
    bool ext_as_constant(ExtTarget x, u64 &a, u64 &b) const { return target_as_constant(x.t[0], a) && target_as_constant(x.t[1], b); }
    // ReducingFactorTarget::reduce_base / reduce (util/reducing.rs): sum_i terms[i] alpha^i on ReducingGate / ReducingExtensionGate
    // rows of max_coeffs_len coefficients each (highest power first, zero-padded at the front, rows chained through old_acc)
    ExtTarget reduce_base(ExtTarget alpha, const std::vector<Target> &terms);
    ExtTarget reduce_ext(ExtTarget alpha, const std::vector<ExtTarget> &terms);
    // interpolate_coset (gadgets/interpolation.rs): the interpolant through (shift * w^i, values[i]), w the generator of the subgroup of
    // order 2^subgroup_bits, evaluated at `point`: one CosetInterpolationGate row (with_max_degree(subgroup_bits, max_quotient_degree_factor))
    ExtTarget interpolate_coset(unsigned subgroup_bits, Target shift, const std::vector<ExtTarget> &values, ExtTarget point);
    // PoseidonMdsGate (gates/poseidon_mds.rs): the Poseidon MDS layer on twelve extension elements, one row
    std::array<ExtTarget, 12> poseidon_mds_ext(const std::array<ExtTarget, 12> &in);
    // base^(sum bits[i] 2^i) for a CONSTANT base, as a product of select(bit, base^(2^i), 1) factors (exp_from_bits_const_base)
    Target exp_from_bits_const_base(u64 base, const std::vector<BoolTarget> &exponent_bits);

    size_t num_gates() const { return rows_.size(); }

    // ---- build ----
    // Consumes the builder's state: hashes the public inputs into a PublicInputGate, places the constants, pads, orders the
    // gates into selector groups, turns the copy classes into sigma, emits the hint trailer (EqualityGenerator,
    // LowHighGenerator instances) and the public-input cells. Returns "" or the reason.
    std::string build(CircuitPack &pack);
    // after build(): the wire cell (row * num_wires + column) a target is copy-connected to, NO_CELL for a target that
    // touches no gate (plonky2 keeps such targets in the partition witness only; they constrain nothing)
    u64 cell_of(Target t);
    // gate rows by gate type after build() (GateType -> count), for profiles like the reference's print_gate_counts
    std::map<uint64_t, size_t> gate_counts() const;
    size_t rows_before_padding() const { return rows_before_padding_; }
    // after build() of a zero-knowledge circuit: the wire cells of the blinding rows that take a fresh random value per proof
    // (CircuitBuilder::blind's RandomValueGenerator targets); the prover's caller assigns them with the PartialWitness
    const std::vector<u64> &blinding_cells() const { return blinding_cells_; }
    size_t blinding_rows() const { return blinding_rows_; }

private:
    struct GateSpec { uint64_t type, p0, p1, p2, degree, ncons; std::string id; };
    struct Row { uint32_t spec; u64 consts[2]; };
    struct ArithKey {
        u64 c0, c1; Target m0, m1, ad;
        bool operator<(const ArithKey &o) const {
            if (c0 != o.c0) return c0 < o.c0;
            if (c1 != o.c1) return c1 < o.c1;
            if (m0 != o.m0) return m0 < o.m0;
            if (m1 != o.m1) return m1 < o.m1;
            return ad < o.ad;
        }
    };
    struct EqHint { Target x, y, equal, inv; };
    struct LowHighHint { Target x, low, high; unsigned n_log; };
    struct SplitHint { Target integer, sum; unsigned shift, bits; };
    struct QuotHint { ExtTarget num, den, quot; };
    struct ExtKey {
        u64 c0, c1; Target v[6];
        bool operator<(const ExtKey &o) const {
            if (c0 != o.c0) return c0 < o.c0;
            if (c1 != o.c1) return c1 < o.c1;
            for (int i = 0; i < 6; i++) if (v[i] != o.v[i]) return v[i] < o.v[i];
            return false;
        }
    };

    Target new_node(u64 cell);
    Target wire(uint32_t row, uint32_t col);
    uint32_t find(uint32_t x);
    uint32_t spec_index(uint64_t type, uint64_t p0, uint64_t p1, uint64_t p2);
    uint32_t add_gate(uint32_t spec, u64 c0 = 0, u64 c1 = 0);
    bool arithmetic_special_cases(u64 c0, u64 c1, Target m0, Target m1, Target ad, Target &out);

    Config cfg_;
    std::vector<uint32_t> parent_;
    std::vector<u64> cell_;                                  // per target: its own wire cell or NO_CELL (virtual)
    std::unordered_map<u64, Target> wire_targets_;           // cell -> target
    std::vector<GateSpec> specs_;
    std::vector<Row> rows_;
    std::map<u64, Target> constants_to_targets_;
    std::unordered_map<Target, u64> targets_to_constants_;
    std::map<ArithKey, Target> arith_results_;
    std::map<std::pair<u64, u64>, std::pair<uint32_t, uint32_t>> arith_slots_;   // (c0, c1) -> (row, next free operation)
    std::map<uint32_t, std::pair<uint32_t, uint32_t>> ra_slots_;                  // bits -> (row, next free copy)
    std::vector<Target> public_inputs_;
    std::vector<EqHint> eq_hints_;
    std::vector<LowHighHint> lh_hints_;
    std::vector<SplitHint> split_hints_;
    std::vector<QuotHint> quot_hints_;
    std::map<ExtKey, ExtTarget> ext_results_;
    std::map<std::pair<u64, u64>, std::pair<uint32_t, uint32_t>> ext_slots_;     // (c0, c1) -> (row, next free operation)
    bool built_ = false;
    size_t rows_before_padding_ = 0, blinding_rows_ = 0;
    int hash_tag_ = -1;
    std::vector<std::pair<int, uint32_t>> p2_rows_;
    std::vector<Target> hint_targets_;
    std::vector<u64> blinding_cells_;
    void blind();
    std::vector<u64> class_cell_;                            // after build(): representative wire cell per class root
};

}  // namespace cb
