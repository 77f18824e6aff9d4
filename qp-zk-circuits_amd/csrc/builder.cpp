// builder.cpp — see builder.hpp. Host only.
#include "builder.hpp"
#include <algorithm>
#include <numeric>
#include <stdexcept>
#include "poseidon.hpp"

namespace cb {

std::string Config::validate() const {
    // the structural policy of the reference's validate_circuit_config (common/src/circuit.rs:426-570), as far as this
    // builder depends on it
    if (num_wires < 135) return "num_wires below the Poseidon gate floor (135)";
    if (num_routed_wires < 37 || num_routed_wires > num_wires) return "num_routed_wires outside 37..num_wires";
    if (num_constants != 2) return "num_constants must be 2";
    if (max_quotient_degree_factor != 8) return "max_quotient_degree_factor must be 8";
    if (rate_bits == 0 || rate_bits > 8 || cap_height > 8) return "FRI parameters out of range";
    if (num_challenges == 0 || num_challenges > 4) return "num_challenges out of range";
    if (min_degree_bits > 20) return "min_degree_bits out of range";
    return p2_layout.validate(num_wires, num_routed_wires);
}

Builder::Builder(const Config &cfg) : cfg_(cfg) {
    const std::string why = cfg_.validate();
    if (!why.empty()) throw std::invalid_argument("circuit config: " + why);
}

Target Builder::new_node(u64 cell) {
    const Target t = (Target)parent_.size();
    parent_.push_back(t);
    cell_.push_back(cell);
    return t;
}
Target Builder::add_virtual_target() { return new_node(NO_CELL); }
std::vector<Target> Builder::add_virtual_targets(size_t n) { std::vector<Target> v(n); for (auto &t : v) t = add_virtual_target(); return v; }
HashOutTarget Builder::add_virtual_hash() { HashOutTarget h; for (auto &e : h.elements) e = add_virtual_target(); return h; }
HashOutTarget Builder::add_virtual_hash_public_input() { HashOutTarget h; for (auto &e : h.elements) e = add_virtual_public_input(); return h; }
BoolTarget Builder::add_virtual_bool_target_safe() { BoolTarget b{add_virtual_target()}; assert_bool(b); return b; }

Target Builder::wire(uint32_t row, uint32_t col) {
    const u64 cell = (u64)row * cfg_.num_wires + col;
    auto it = wire_targets_.find(cell);
    if (it != wire_targets_.end()) return it->second;
    const Target t = new_node(cell);
    wire_targets_.emplace(cell, t);
    return t;
}

uint32_t Builder::find(uint32_t x) {
    while (parent_[x] != x) { parent_[x] = parent_[parent_[x]]; x = parent_[x]; }
    return x;
}

void Builder::connect(Target a, Target b) {
    // Target::is_routable: a wire beyond the routed prefix cannot take part in the permutation argument
    for (Target t : {a, b})
        if (cell_[t] != NO_CELL && cell_[t] % cfg_.num_wires >= cfg_.num_routed_wires) throw std::logic_error("connect: wire is not routable");
    const uint32_t ra = find(a), rb = find(b);
    if (ra != rb) parent_[std::max(ra, rb)] = std::min(ra, rb);
}

void Builder::assert_bool(BoolTarget b) {
    const Target z = mul_sub(b.target, b.target, b.target);
    connect(z, zero());
}

Target Builder::constant(u64 c) {
    c = gl::canon(c);
    auto it = constants_to_targets_.find(c);
    if (it != constants_to_targets_.end()) return it->second;
    const Target t = add_virtual_target();
    constants_to_targets_.emplace(c, t);
    targets_to_constants_.emplace(t, c);
    return t;
}
bool Builder::target_as_constant(Target t, u64 &out) const {
    auto it = targets_to_constants_.find(t);
    if (it == targets_to_constants_.end()) return false;
    out = it->second;
    return true;
}

uint32_t Builder::spec_index(uint64_t type, uint64_t p0, uint64_t p1, uint64_t p2) {
    for (size_t i = 0; i < specs_.size(); i++)
        if (specs_[i].type == type && specs_[i].p0 == p0 && specs_[i].p1 == p1 && specs_[i].p2 == p2) return (uint32_t)i;
    GateSpec s{type, p0, p1, p2, 0, 0, ""};
    switch (type) {
    case GATE_NOOP: s.degree = 0; s.ncons = 0; s.id = "NoopGate"; break;
    case GATE_CONSTANT: s.degree = 1; s.ncons = p0; s.id = "ConstantGate { num_consts: " + std::to_string(p0) + " }"; break;
    case GATE_PUBLIC_INPUT: s.degree = 1; s.ncons = 4; s.id = "PublicInputGate"; break;
    case GATE_ARITHMETIC: s.degree = 3; s.ncons = p0; s.id = "ArithmeticGate { num_ops: " + std::to_string(p0) + " }"; break;
    case GATE_ARITHMETIC_EXT: s.degree = 3; s.ncons = 2 * p0; s.id = "ArithmeticExtensionGate { num_ops: " + std::to_string(p0) + " }"; break;
    case GATE_REDUCING: s.degree = 2; s.ncons = 2 * p0; s.id = "ReducingGate { num_coeffs: " + std::to_string(p0) + " }"; break;
    case GATE_REDUCING_EXT: s.degree = 2; s.ncons = 2 * p0; s.id = "ReducingExtensionGate { num_coeffs: " + std::to_string(p0) + " }"; break;
    case GATE_COSET_INTERPOLATION: { const uint64_t np = 1ull << p0; s.degree = p1; s.ncons = 2 * (2 + 2 * ((np - 2) / (p1 - 1)));
                                     s.id = "CosetInterpolationGate { subgroup_bits: " + std::to_string(p0) + ", degree: " + std::to_string(p1) + " }"; break; }
    case GATE_POSEIDON_MDS: s.degree = 1; s.ncons = 24; s.id = "PoseidonMdsGate(PhantomData<plonky2_field::goldilocks_field::GoldilocksField>)<WIDTH=12>"; break;
    case GATE_BASE_SUM: s.degree = 2; s.ncons = p0 + 1; s.id = "BaseSumGate { num_limbs: " + std::to_string(p0) + " } + Base: 2"; break;
    case GATE_POSEIDON: s.degree = 7; s.ncons = 123; s.id = "PoseidonGate(PhantomData<plonky2_field::goldilocks_field::GoldilocksField>)<WIDTH=12>"; break;
    case GATE_RANDOM_ACCESS: s.degree = p0 + 1; s.ncons = p1 * (p0 + 2) + p2; s.id = "RandomAccessGate { bits: " + std::to_string(p0) + ", num_copies: " + std::to_string(p1) + " }"; break;
    case GATE_POSEIDON2: s.degree = 7; s.ncons = cfg_.p2_layout.num_constraints(); s.id = "Poseidon2Gate(PhantomData<plonky2_field::goldilocks_field::GoldilocksField>)<WIDTH=12>"; break;
    default: throw std::logic_error("builder: gate type not supported");
    }
    specs_.push_back(s);
    return (uint32_t)specs_.size() - 1;
}

uint32_t Builder::add_gate(uint32_t spec, u64 c0, u64 c1) {
    if (rows_.size() >= (1u << 22)) throw std::length_error("builder: too many rows");
    rows_.push_back({spec, {gl::canon(c0), gl::canon(c1)}});
    return (uint32_t)rows_.size() - 1;
}

// gadgets/arithmetic.rs: arithmetic_special_cases
bool Builder::arithmetic_special_cases(u64 c0, u64 c1, Target m0, Target m1, Target ad, Target &out) {
    const Target z = zero();
    u64 m0c = 0, m1c = 0, adc = 0;
    const bool m0_const = target_as_constant(m0, m0c), m1_const = target_as_constant(m1, m1c), ad_const = target_as_constant(ad, adc);
    const bool first_term_zero = c0 == 0 || m0 == z || m1 == z;
    const bool second_term_zero = c1 == 0 || ad == z;
    bool first_known = false, second_known = false;
    u64 first = 0, second = 0;
    if (first_term_zero) first_known = true;
    else if (m0_const && m1_const) { first_known = true; first = gl::mul(gl::mul(m0c, m1c), c0); }
    if (second_term_zero) second_known = true;
    else if (ad_const) { second_known = true; second = gl::mul(adc, c1); }
    if (first_known && second_known) { out = constant(gl::add(first, second)); return true; }
    if (first_term_zero && c1 == 1) { out = ad; return true; }
    if (second_term_zero) {
        if (m0_const && gl::canon(gl::mul(m0c, c0)) == 1) { out = m1; return true; }
        if (m1_const && gl::canon(gl::mul(m1c, c0)) == 1) { out = m0; return true; }
    }
    return false;
}

Target Builder::arithmetic(u64 c0, u64 c1, Target m0, Target m1, Target ad) {
    c0 = gl::canon(c0); c1 = gl::canon(c1);
    Target out;
    if (arithmetic_special_cases(c0, c1, m0, m1, ad, out)) return out;
    const ArithKey key{c0, c1, m0, m1, ad};
    auto it = arith_results_.find(key);
    if (it != arith_results_.end()) return it->second;
    // find_slot: operations with the same constants share ArithmeticGate rows
    const uint32_t num_ops = cfg_.num_routed_wires / 4;
    auto slot = arith_slots_.find({c0, c1});
    uint32_t row, op;
    if (slot == arith_slots_.end()) { row = add_gate(spec_index(GATE_ARITHMETIC, num_ops, 0, 0), c0, c1); op = 0; }
    else { row = slot->second.first; op = slot->second.second; }
    if (op + 1 < num_ops) arith_slots_[{c0, c1}] = {row, op + 1};
    else arith_slots_.erase({c0, c1});
    connect(m0, wire(row, 4 * op));
    connect(m1, wire(row, 4 * op + 1));
    connect(ad, wire(row, 4 * op + 2));
    out = wire(row, 4 * op + 3);
    arith_results_.emplace(key, out);
    return out;
}

// gadgets/arithmetic_extension.rs: arithmetic_extension. Operands that are all constants fold; a vanishing product term with
// const_1 = 1 is the addend; a vanishing addend term with const_0 * (a constant multiplicand) = 1 is the other multiplicand.
ExtTarget Builder::arithmetic_ext(u64 c0, u64 c1, ExtTarget m0, ExtTarget m1, ExtTarget ad) {
    c0 = gl::canon(c0); c1 = gl::canon(c1);
    u64 a0, a1, b0, b1, d0, d1;
    const bool m0c = ext_as_constant(m0, a0, a1), m1c = ext_as_constant(m1, b0, b1), adc = ext_as_constant(ad, d0, d1);
    const bool first_zero = c0 == 0 || (m0c && a0 == 0 && a1 == 0) || (m1c && b0 == 0 && b1 == 0);
    const bool second_zero = c1 == 0 || (adc && d0 == 0 && d1 == 0);
    if ((first_zero || (m0c && m1c)) && (second_zero || adc)) {
        gl::e2 r = gl::e2_make(0, 0);
        if (!first_zero) r = gl::e2_scale(gl::e2_mul(gl::e2_make(a0, a1), gl::e2_make(b0, b1)), c0);
        if (!second_zero) r = gl::e2_add(r, gl::e2_scale(gl::e2_make(d0, d1), c1));
        r = gl::e2_canon(r);
        return constant_ext(r.a, r.b);
    }
    if (first_zero && c1 == 1) return ad;
    if (second_zero) {
        if (m0c && a1 == 0 && gl::canon(gl::mul(a0, c0)) == 1) return m1;
        if (m1c && b1 == 0 && gl::canon(gl::mul(b0, c0)) == 1) return m0;
    }
    const ExtKey key{c0, c1, {m0.t[0], m0.t[1], m1.t[0], m1.t[1], ad.t[0], ad.t[1]}};
    auto it = ext_results_.find(key);
    if (it != ext_results_.end()) return it->second;
    const uint32_t num_ops = cfg_.num_routed_wires / 8;
    auto slot = ext_slots_.find({c0, c1});
    uint32_t row, op;
    if (slot == ext_slots_.end()) { row = add_gate(spec_index(GATE_ARITHMETIC_EXT, num_ops, 0, 0), c0, c1); op = 0; }
    else { row = slot->second.first; op = slot->second.second; }
    if (op + 1 < num_ops) ext_slots_[{c0, c1}] = {row, op + 1};
    else ext_slots_.erase({c0, c1});
    const ExtTarget in[3] = {m0, m1, ad};
    for (uint32_t k = 0; k < 3; k++) for (uint32_t e = 0; e < 2; e++) connect(in[k].t[e], wire(row, 8 * op + 2 * k + e));
    const ExtTarget out = {{wire(row, 8 * op + 6), wire(row, 8 * op + 7)}};
    ext_results_.emplace(key, out);
    return out;
}

ExtTarget Builder::reduce_base(ExtTarget alpha, const std::vector<Target> &terms) {
    const uint32_t n = std::min<uint32_t>((cfg_.num_wires - 6) / 3, cfg_.num_routed_wires - 6);      // ReducingGate::max_coeffs_len
    std::vector<Target> rev(terms);
    while (rev.size() % n) rev.push_back(zero());
    std::reverse(rev.begin(), rev.end());
    ExtTarget acc = zero_ext();
    for (size_t at = 0; at < rev.size(); at += n) {
        const uint32_t row = add_gate(spec_index(GATE_REDUCING, n, 0, 0));
        for (uint32_t e = 0; e < 2; e++) { connect(alpha.t[e], wire(row, 2 + e)); connect(acc.t[e], wire(row, 4 + e)); }
        for (uint32_t i = 0; i < n; i++) connect(rev[at + i], wire(row, 6 + i));
        acc = {{wire(row, 0), wire(row, 1)}};
    }
    return acc;
}
ExtTarget Builder::reduce_ext(ExtTarget alpha, const std::vector<ExtTarget> &terms) {
    const uint32_t n = std::min<uint32_t>((cfg_.num_wires - 6) / 4, (cfg_.num_routed_wires - 6) / 2);  // ReducingExtensionGate::max_coeffs_len
    std::vector<ExtTarget> rev(terms);
    while (rev.size() % n) rev.push_back(zero_ext());
    std::reverse(rev.begin(), rev.end());
    ExtTarget acc = zero_ext();
    for (size_t at = 0; at < rev.size(); at += n) {
        const uint32_t row = add_gate(spec_index(GATE_REDUCING_EXT, n, 0, 0));
        for (uint32_t e = 0; e < 2; e++) { connect(alpha.t[e], wire(row, 2 + e)); connect(acc.t[e], wire(row, 4 + e)); }
        for (uint32_t i = 0; i < n; i++) for (uint32_t e = 0; e < 2; e++) connect(rev[at + i].t[e], wire(row, 6 + 2 * i + e));
        acc = {{wire(row, 0), wire(row, 1)}};
    }
    return acc;
}

ExtTarget Builder::interpolate_coset(unsigned bits, Target shift, const std::vector<ExtTarget> &values, ExtTarget point) {
    const uint64_t np = 1ull << bits, max_degree = cfg_.max_quotient_degree_factor;
    if (values.size() != np || bits < 2 || bits > 5) throw std::logic_error("interpolate_coset: 2^subgroup_bits values, 2..5 bits");
    // CosetInterpolationGate::with_max_degree: as few intermediates as the degree bound allows, then the smallest degree with that many
    const uint64_t n_int = (np - 2) / (max_degree - 1), degree = (np - 2) / (n_int + 1) + 2;
    const uint32_t s_ep = 1 + 2 * (uint32_t)np, row = add_gate(spec_index(GATE_COSET_INTERPOLATION, bits, degree, 0));
    if (s_ep + 4 > cfg_.num_routed_wires) throw std::logic_error("interpolate_coset: too few routed wires");
    connect(shift, wire(row, 0));
    for (uint32_t i = 0; i < np; i++) for (uint32_t e = 0; e < 2; e++) connect(values[i].t[e], wire(row, 1 + 2 * i + e));
    for (uint32_t e = 0; e < 2; e++) connect(point.t[e], wire(row, s_ep + e));
    return {{wire(row, s_ep + 2), wire(row, s_ep + 3)}};
}
std::array<ExtTarget, 12> Builder::poseidon_mds_ext(const std::array<ExtTarget, 12> &in) {
    if (cfg_.num_routed_wires < 48) throw std::logic_error("poseidon_mds_ext: too few routed wires");
    const uint32_t row = add_gate(spec_index(GATE_POSEIDON_MDS, 0, 0, 0));
    std::array<ExtTarget, 12> out;
    for (uint32_t i = 0; i < 12; i++) for (uint32_t e = 0; e < 2; e++) { connect(in[i].t[e], wire(row, 2 * i + e)); out[i].t[e] = wire(row, 24 + 2 * i + e); }
    return out;
}

ExtTarget Builder::div_ext(ExtTarget x, ExtTarget y) {
    const ExtTarget q = {{add_virtual_target(), add_virtual_target()}};
    quot_hints_.push_back({x, y, q});
    connect_ext(mul_ext(q, y), x);
    return q;
}

// gadgets/arithmetic.rs exp_from_bits_const_base: prod_i (bit_i ? base^(2^i) : 1), two factors folded per operation where
// upstream does (mul_many over select-by-constant terms; the product is the same field element)
Target Builder::exp_from_bits_const_base(u64 base, const std::vector<BoolTarget> &bits) {
    Target acc = one();
    u64 pw = gl::canon(base);
    for (const BoolTarget &b : bits) {
        // factor = 1 + bit * (pw - 1); acc' = acc * factor = (pw - 1) * acc * bit + acc
        acc = arithmetic(gl::sub(pw, 1), 1, acc, b.target, acc);
        pw = gl::canon(gl::mul(pw, pw));
    }
    return acc;
}

// gadgets/arithmetic.rs: is_equal, with its EqualityGenerator (equal = [x == y], inv = 1 / (x - y) or 0)
BoolTarget Builder::is_equal(Target x, Target y) {
    const Target z = zero();
    const BoolTarget equal = add_virtual_bool_target_unsafe();
    const BoolTarget not_equal = not_(equal);
    const Target inv = add_virtual_target();
    eq_hints_.push_back({x, y, equal.target, inv});
    const Target diff = sub(x, y);
    const Target not_equal_check = mul(diff, equal.target);
    const Target diff_normalized = mul(diff, inv);
    connect(diff_normalized, not_equal.target);
    connect(not_equal_check, z);
    return equal;
}

// gadgets/split_join.rs: split_le on BaseSumGate<2> rows of min(63, num_routed_wires - 1) limbs
std::vector<BoolTarget> Builder::split_le(Target integer, unsigned num_bits) {
    std::vector<BoolTarget> bits;
    if (num_bits == 0) return bits;
    const unsigned limbs = std::min<unsigned>(63, cfg_.num_routed_wires - 1);
    const unsigned k = (num_bits + limbs - 1) / limbs;
    if (num_bits > 64) throw std::logic_error("split_le: more than 64 bits");
    std::vector<uint32_t> gates(k);
    for (unsigned g = 0; g < k; g++) gates[g] = add_gate(spec_index(GATE_BASE_SUM, limbs, 2, 0));
    for (unsigned g = 0; g < k; g++)
        for (unsigned i = 0; i < limbs; i++) {
            const Target b = wire(gates[g], 1 + i);
            if (g * limbs + i < num_bits) bits.push_back({b});
            else assert_zero(b);
        }
    // acc = sum_{k-1} * 2^(limbs (k-1)) + .. + sum_0, connected to the integer. One gate: the sum wire is a copy of the integer (the
    // WireSplitGenerator's write and the copy constraint agree) and the gate's own generator splits it into limbs; several gates
    // (a 64-bit split: upstream does not exclude the non-canonical decomposition there, and neither does this): every gate's sum
    // wire is filled by its WireSplitGenerator (hint trailer)
    if (k == 1) { connect(wire(gates[0], 0), integer); return bits; }
    Target acc = zero();
    for (unsigned g = k; g-- > 0;) acc = mul_const_add(1ull << limbs, acc, wire(gates[g], 0));
    connect(acc, integer);
    for (unsigned g = 0; g < k; g++) split_hints_.push_back({integer, wire(gates[g], 0), g * limbs, limbs});
    return bits;
}

void Builder::split_low_high(Target x, unsigned n_log, unsigned num_bits, Target &low, Target &high) {
    low = add_virtual_target();
    high = add_virtual_target();
    lh_hints_.push_back({x, low, high, n_log});
    range_check(low, n_log);
    range_check(high, num_bits - n_log);
    const Target pow2 = constant(1ull << n_log);
    const Target comp_x = mul_add(high, pow2, low);
    connect(x, comp_x);
}

Builder::State Builder::permute_poseidon(const State &in, BoolTarget swap) {
    const uint32_t row = add_gate(spec_index(GATE_POSEIDON, 0, 0, 0));
    connect(swap.target, wire(row, 24));
    for (uint32_t i = 0; i < 12; i++) connect(in[i], wire(row, i));
    State out;
    for (uint32_t i = 0; i < 12; i++) out[i] = wire(row, 12 + i);
    return out;
}

Builder::State Builder::permute_poseidon2(const State &in) {
    const P2GateLayout &l = cfg_.p2_layout;
    const uint32_t row = add_gate(spec_index(GATE_POSEIDON2, 0, 0, 0));
    p2_rows_.push_back({hash_tag_, row});
    if (l.has_swap()) connect(zero(), wire(row, l.w_swap));
    for (uint32_t i = 0; i < 12; i++) connect(in[i], wire(row, l.w_input + i));
    State out;
    for (uint32_t i = 0; i < 12; i++) out[i] = wire(row, l.w_output + i);
    return out;
}

Builder::State Builder::permute_swapped(const State &in, BoolTarget swap) {
    if (cfg_.inner_hasher != hasher::POSEIDON2) return permute_poseidon(in, swap);
    const P2GateLayout &l = cfg_.p2_layout;
    if (!l.has_swap()) {
        u64 c = 1;
        if (target_as_constant(swap.target, c) && c == 0) return permute_poseidon2(in);
        throw std::logic_error("permute_swapped: the Poseidon2 gate layout has no swap wire");
    }
    const uint32_t row = add_gate(spec_index(GATE_POSEIDON2, 0, 0, 0));
    connect(swap.target, wire(row, l.w_swap));
    for (uint32_t i = 0; i < 12; i++) connect(in[i], wire(row, l.w_input + i));
    State out;
    for (uint32_t i = 0; i < 12; i++) out[i] = wire(row, l.w_output + i);
    return out;
}

HashOutTarget Builder::hash_or_noop(const std::vector<Target> &inputs) {
    if (inputs.size() > 4) return hash_n_to_hash_no_pad(inputs);
    HashOutTarget h;
    for (size_t i = 0; i < 4; i++) h.elements[i] = i < inputs.size() ? inputs[i] : zero();
    return h;
}

Target Builder::random_access(Target access_index, const std::vector<Target> &v) {
    if (v.size() == 1) return v[0];
    uint32_t bits = 0;
    while ((1ull << bits) < v.size()) bits++;
    if ((1ull << bits) != v.size() || bits > 6) throw std::logic_error("random_access: the vector length must be a power of two up to 64");
    // RandomAccessGate::new_from_config: as many copies as the routed wires (2 + 2^bits each) and the wires (+ bits each) hold
    const uint32_t vec = 1u << bits;
    const uint32_t copies = std::min<uint32_t>(cfg_.num_routed_wires / (2 + vec), cfg_.num_wires / (2 + vec + bits));
    if (copies == 0) throw std::logic_error("random_access: the gate does not fit the routed wires");
    const uint32_t extra = std::min<uint32_t>(cfg_.num_routed_wires - (2 + vec) * copies, cfg_.num_constants);
    auto slot = ra_slots_.find(bits);
    uint32_t row, copy;
    if (slot == ra_slots_.end()) { row = add_gate(spec_index(GATE_RANDOM_ACCESS, bits, copies, extra)); copy = 0; }
    else { row = slot->second.first; copy = slot->second.second; }
    if (copy + 1 < copies) ra_slots_[bits] = {row, copy + 1};
    else ra_slots_.erase(bits);
    const uint32_t b0 = (2 + vec) * copy;
    const Target claimed = add_virtual_target();
    for (uint32_t i = 0; i < vec; i++) connect(v[i], wire(row, b0 + 2 + i));
    connect(access_index, wire(row, b0));
    connect(claimed, wire(row, b0 + 1));
    return claimed;
}

Target Builder::le_sum(const std::vector<BoolTarget> &bits) {
    if (bits.empty()) return zero();
    if (bits.size() - 1 > cfg_.num_routed_wires / 4) throw std::logic_error("le_sum: more bits than one ArithmeticGate row of operations (the BaseSumGate form is not restated)");
    const Target two_ = two();
    Target sum = bits.back().target;
    for (size_t i = bits.size() - 1; i-- > 0;) sum = mul_add(two_, sum, bits[i].target);
    return sum;
}

void Builder::verify_merkle_proof_to_cap_with_cap_index(const std::vector<Target> &leaf_data, const std::vector<BoolTarget> &leaf_index_bits, Target cap_index,
                                                        const std::vector<HashOutTarget> &cap, const std::vector<HashOutTarget> &siblings) {
    if (leaf_index_bits.size() < siblings.size()) throw std::logic_error("verify_merkle_proof: fewer index bits than siblings");
    const Target z = zero();
    HashOutTarget state = hash_or_noop(leaf_data);
    for (size_t l = 0; l < siblings.size(); l++) {
        State in;
        for (int i = 0; i < 4; i++) { in[i] = state.elements[i]; in[4 + i] = siblings[l].elements[i]; in[8 + i] = z; }
        const State out = permute_swapped(in, leaf_index_bits[l]);
        for (int i = 0; i < 4; i++) state.elements[i] = out[i];
    }
    for (int i = 0; i < 4; i++) {
        std::vector<Target> column(cap.size());
        for (size_t k = 0; k < cap.size(); k++) column[k] = cap[k].elements[i];
        connect(random_access(cap_index, column), state.elements[i]);
    }
}

HashOutTarget Builder::hash_n_to_hash_no_pad(const std::vector<Target> &inputs) {
    State state;
    state.fill(zero());
    for (size_t i = 0; i < inputs.size(); i += 8) {
        const size_t len = std::min<size_t>(8, inputs.size() - i);
        for (size_t k = 0; k < len; k++) state[k] = inputs[i + k];          // overwrite mode
        state = permute_swapped(state, _false());
    }
    HashOutTarget h;
    for (int i = 0; i < 4; i++) h.elements[i] = state[i];
    return h;
}

HashOutTarget Builder::hash_n_to_hash_no_pad_p2(const std::vector<Target> &inputs) {
    State state;
    state.fill(zero());
    const size_t padded = (inputs.size() + 1 + 7) / 8 * 8;
    for (size_t i = 0; i < padded; i += 8) {
        for (size_t j = 0; j < 8; j++) {
            const size_t k = i + j;
            const Target m = k < inputs.size() ? inputs[k] : (k == inputs.size() ? one() : zero());
            state[j] = add(state[j], m);        // (the first block's additions onto zero fold away: arithmetic's special cases)
        }
        state = permute_poseidon2(state);
    }
    HashOutTarget h;
    for (int i = 0; i < 4; i++) h.elements[i] = state[i];
    return h;
}

std::map<uint64_t, size_t> Builder::gate_counts() const {
    std::map<uint64_t, size_t> m;
    for (const Row &r : rows_) m[specs_[r.spec].type]++;
    return m;
}

// CircuitBuilder::blind (plonk/circuit_builder.rs, upstream plonky2): blinding_counts looks for the smallest degree estimate
// 2^k >= the gate count whose own blinding rows still fit under it; one NoopGate row of random wires per opening of a regular
// polynomial (D at zeta + what the FRI queries reveal), and per opening of Z (2 D: zeta and g zeta) a PAIR of NoopGate rows whose
// routed wires hold the same random values under a copy constraint (random factors of the permutation product that cancel).
// Restated from upstream: the fork's "row blinding" mode may count differently (un-vendored).
void Builder::blind() {
    const size_t num_gates = rows_.size(), D = 2;
    unsigned k = 0;
    while ((1ull << k) < num_gates) k++;
    size_t regular = 0, zs = 0;
    for (;; k++) {
        if (k > 22) throw std::length_error("blind: circuit too large");
        const std::vector<uint64_t> arity_bits = fri_reduction_arity_bits(k, cfg_.rate_bits, cfg_.cap_height, cfg_.arity_bits, cfg_.final_poly_bits);
        size_t folding_points = 0, reduced_bits = 0;
        for (uint64_t ab : arity_bits) { folding_points += ((size_t)1 << ab) - 1; reduced_bits += ab; }
        const size_t final_poly_coeffs = (size_t)1 << (k - reduced_bits);
        const size_t fri_openings = cfg_.num_query_rounds * (1 + D * folding_points + D * final_poly_coeffs);
        regular = D + fri_openings; zs = 2 * D + fri_openings;
        if (num_gates + regular + 2 * zs <= (1ull << k)) break;
    }
    const uint32_t noop = spec_index(GATE_NOOP, 0, 0, 0);
    for (size_t i = 0; i < regular; i++) {
        const uint32_t row = add_gate(noop);
        for (uint32_t w = 0; w < cfg_.num_wires; w++) blinding_cells_.push_back((u64)row * cfg_.num_wires + w);
    }
    for (size_t i = 0; i < zs; i++) {
        const uint32_t g1 = add_gate(noop), g2 = add_gate(noop);
        for (uint32_t w = 0; w < cfg_.num_routed_wires; w++) {
            connect(wire(g1, w), wire(g2, w));
            blinding_cells_.push_back((u64)g1 * cfg_.num_wires + w);
        }
    }
    blinding_rows_ = regular + 2 * zs;
}

u64 Builder::cell_of(Target t) {
    if (!built_ || t >= parent_.size()) return NO_CELL;
    return class_cell_[find(t)];
}

std::string Builder::build(CircuitPack &pack) {
    if (built_) return "build: already built";
    const uint32_t NW = cfg_.num_wires, R = cfg_.num_routed_wires;
    // CircuitBuilder::build: hash the public inputs and route the hash into a PublicInputGate
    const HashOutTarget pih = hash_n_to_hash_no_pad(public_inputs_);
    const uint32_t pi_row = add_gate(spec_index(GATE_PUBLIC_INPUT, 0, 0, 0));
    for (uint32_t i = 0; i < 4; i++) connect(pih.elements[i], wire(pi_row, i));
    // the operation slots left over in the last ArithmeticGate row of every constant pair are wired to zero, so that every
    // generator of the row has its inputs (plonky2's fill_batched_gates; without it `generate_partial_witness` ends with
    // generators that never ran)
    {
        const uint32_t num_ops = R / 4;
        const Target z = zero();
        for (const auto &slot : arith_slots_)
            for (uint32_t op = slot.second.second; op < num_ops; op++)
                for (uint32_t k = 0; k < 3; k++) connect(z, wire(slot.second.first, 4 * op + k));
        arith_slots_.clear();
        for (const auto &slot : ext_slots_)
            for (uint32_t op = slot.second.second; op < R / 8; op++)
                for (uint32_t k = 0; k < 6; k++) connect(z, wire(slot.second.first, 8 * op + k));
        ext_slots_.clear();
        // the copies left over in the last RandomAccessGate row of every width: index 0 into a list of zeros
        for (const auto &slot : ra_slots_) {
            const uint32_t bits = slot.first, vec = 1u << bits;
            const uint32_t copies = std::min<uint32_t>(R / (2 + vec), NW / (2 + vec + bits));
            for (uint32_t copy = slot.second.second; copy < copies; copy++) {
                const uint32_t b0 = (2 + vec) * copy;
                connect(z, wire(slot.second.first, b0));
                for (uint32_t i = 0; i < vec; i++) connect(z, wire(slot.second.first, b0 + 2 + i));
            }
        }
        ra_slots_.clear();
    }
    // every constant used gets a ConstantGate slot, in the order of the constants' values
    {
        const uint32_t per_row = cfg_.num_constants;
        uint32_t row = 0, used = per_row;
        for (const auto &ct : constants_to_targets_) {      // std::map: sorted by value
            if (used == per_row) { row = add_gate(spec_index(GATE_CONSTANT, per_row, 0, 0)); used = 0; }
            rows_[row].consts[used] = ct.first;
            connect(wire(row, used), ct.second);
            used++;
        }
    }
    rows_before_padding_ = rows_.size();
    if (cfg_.zero_knowledge) blind();
    // blind_and_pad: NoopGate rows up to a power of two
    unsigned degree_bits = std::max<unsigned>(cfg_.min_degree_bits, 5);
    while ((1ull << degree_bits) < rows_.size()) degree_bits++;
    if (degree_bits > 20) return "build: circuit too large";
    const u64 n = 1ull << degree_bits;
    if (rows_.size() < n) { const uint32_t noop = spec_index(GATE_NOOP, 0, 0, 0); while (rows_.size() < n) add_gate(noop); }

    // gates sorted by (degree, id), grouped into selector polynomials (plonk/circuit_builder.rs: selector_polynomials)
    std::vector<uint32_t> order(specs_.size());
    std::iota(order.begin(), order.end(), 0u);
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return specs_[a].degree != specs_[b].degree ? specs_[a].degree < specs_[b].degree : specs_[a].id < specs_[b].id; });
    const u64 max_degree = cfg_.max_quotient_degree_factor + 1;
    std::vector<std::pair<size_t, size_t>> groups;
    if (specs_[order.back()].degree + order.size() - 1 <= max_degree) groups.push_back({0, order.size()});
    else {
        for (size_t start = 0; start < order.size();) {
            size_t size = 0;
            while (start + size < order.size() && size + specs_[order[start + size]].degree < max_degree) size++;
            if (size == 0) return "build: gate degree too high for the quotient degree";
            groups.push_back({start, start + size});
            start += size;
        }
    }
    std::vector<uint32_t> index_of_spec(specs_.size()), group_of_spec(specs_.size());
    pack = CircuitPack();
    pack.gates.resize(order.size());
    for (size_t i = 0; i < order.size(); i++) {
        size_t grp = 0;
        while (!(groups[grp].first <= i && i < groups[grp].second)) grp++;
        const GateSpec &s = specs_[order[i]];
        pack.gates[i] = {s.type, s.p0, s.p1, grp, groups[grp].first, groups[grp].second, s.ncons, s.p2};
        index_of_spec[order[i]] = (uint32_t)i; group_of_spec[order[i]] = (uint32_t)grp;
        pack.num_gate_constraints = std::max<uint64_t>(pack.num_gate_constraints, s.ncons);
    }
    pack.degree_bits = degree_bits; pack.num_wires = NW; pack.num_routed_wires = R;
    pack.num_constants = cfg_.num_constants; pack.num_selectors = groups.size(); pack.num_challenges = cfg_.num_challenges;
    pack.quotient_degree_factor = cfg_.max_quotient_degree_factor;
    pack.num_partial_products = (R + pack.quotient_degree_factor - 1) / pack.quotient_degree_factor - 1;
    pack.num_public_inputs = public_inputs_.size();
    pack.rate_bits = cfg_.rate_bits; pack.cap_height = cfg_.cap_height; pack.proof_of_work_bits = cfg_.proof_of_work_bits;
    pack.num_query_rounds = cfg_.num_query_rounds; pack.zero_knowledge = cfg_.zero_knowledge ? 1 : 0;
    pack.arity_bits = fri_reduction_arity_bits(degree_bits, cfg_.rate_bits, cfg_.cap_height, cfg_.arity_bits, cfg_.final_poly_bits);
    for (const GateSpec &s : specs_) if (s.type == GATE_POSEIDON2) { pack.p2_layout = cfg_.p2_layout; pack.has_p2_layout = true; }
    pack.k_is.resize(R);
    { u64 k = 1; for (uint32_t j = 0; j < R; j++) { pack.k_is[j] = gl::canon(k); k = gl::mul(k, gl::MULT_GEN); } }

    const u64 ncs = pack.num_cs_cols(), sel_cols = pack.num_selectors, UNUSED = 0xFFFFFFFFull;
    pack.constants_sigmas.assign(ncs * n, 0);
    auto CS = [&](u64 row, u64 col) -> u64 & { return pack.constants_sigmas[col * n + row]; };
    for (u64 r = 0; r < n; r++) {
        const Row &row = rows_[r];
        for (u64 s = 0; s < sel_cols; s++) CS(r, s) = group_of_spec[row.spec] == s ? index_of_spec[row.spec] : UNUSED;
        for (u64 i = 0; i < cfg_.num_constants; i++) CS(r, sel_cols + i) = row.consts[i];
    }

    // sigma: one cycle through the routed cells of every copy class; a class's representative cell = its first wire
    std::vector<uint32_t> cells_by_class;                   // routed wire cells as (row * R + col), sorted by class root
    class_cell_.assign(parent_.size(), NO_CELL);
    {
        std::vector<std::pair<uint32_t, uint32_t>> rc;      // (class root, row * R + col)
        for (Target t = 0; t < parent_.size(); t++) {
            if (cell_[t] == NO_CELL) continue;
            const u64 row = cell_[t] / NW, col = cell_[t] % NW;
            const uint32_t root = find(t);
            if (class_cell_[root] == NO_CELL || cell_[t] < class_cell_[root]) class_cell_[root] = cell_[t];
            if (col < R) rc.push_back({root, (uint32_t)(row * R + col)});
        }
        std::sort(rc.begin(), rc.end());
        std::vector<uint32_t> next((size_t)n * R);
        std::iota(next.begin(), next.end(), 0u);
        for (size_t s = 0; s < rc.size();) {
            size_t e = s;
            while (e < rc.size() && rc[e].first == rc[s].first) e++;
            for (size_t k = s; k < e; k++) next[rc[k].second] = rc[k + 1 < e ? k + 1 : s].second;
            s = e;
        }
        std::vector<u64> omega_pow(n);
        { const u64 w = gl::root_of_unity(degree_bits); u64 a = 1; for (u64 i = 0; i < n; i++) { omega_pow[i] = gl::canon(a); a = gl::mul(a, w); } }
        const u64 sig0 = sel_cols + cfg_.num_constants;
        for (u64 r = 0; r < n; r++)
            for (u64 c = 0; c < R; c++) {
                const uint32_t t = next[r * R + c];
                CS(r, sig0 + c) = gl::canon(gl::mul(pack.k_is[t % R], omega_pow[t / R]));
            }
    }
    built_ = true;

    // generators that are not attached to a gate
    auto need_cell = [&](Target t, const char *what) -> u64 {
        const u64 c = cell_of(t);
        if (c == NO_CELL) throw std::logic_error(std::string("build: a generator's ") + what + " touches no gate");
        return c;
    };
    for (const EqHint &h : eq_hints_)
        pack.hints.push_back({{HINT_EQUALITY, need_cell(h.x, "input"), need_cell(h.y, "input"), need_cell(h.equal, "output"), need_cell(h.inv, "output"), 0, 0, 0}});
    for (const SplitHint &h : split_hints_)
        pack.hints.push_back({{HINT_WIRE_SPLIT, need_cell(h.integer, "input"), need_cell(h.sum, "output"), h.shift, h.bits, 0, 0, 0}});
    for (const LowHighHint &h : lh_hints_)
        pack.hints.push_back({{HINT_LOW_HIGH, need_cell(h.x, "input"), need_cell(h.low, "output"), need_cell(h.high, "output"), h.n_log, 0, 0, 0}});
    for (const QuotHint &h : quot_hints_)
        pack.hints.push_back({{HINT_QUOTIENT_EXT, need_cell(h.num.t[0], "input"), need_cell(h.num.t[1], "input"), need_cell(h.den.t[0], "input"), need_cell(h.den.t[1], "input"),
                               need_cell(h.quot.t[0], "output"), need_cell(h.quot.t[1], "output"), 0}});
    pack.pi_cells.resize(public_inputs_.size());
    for (size_t i = 0; i < public_inputs_.size(); i++) pack.pi_cells[i] = need_cell(public_inputs_[i], "public input");

    // circuit_digest: upstream hashes the constants/sigmas cap, the domain separator and the degree; the cap needs the
    // commitment, which is computed at load time on the device. Here: the proof-system hash of the shape words and of a running
    // hash over the constants/sigmas VALUES, which binds the transcript to this circuit. NOT the fork's formula.
    {
        auto hash_no_pad = [](const u64 *in, size_t cnt, u64 out[4]) {
            u64 st[12] = {0};
            for (size_t i = 0; i < cnt; i += 8) {
                const size_t len = std::min<size_t>(8, cnt - i);
                for (size_t k = 0; k < len; k++) st[k] = gl::canon(in[i + k]);
                hasher::host_permute(st);
            }
            for (int i = 0; i < 4; i++) out[i] = st[i];
        };
        u64 vh[4];
        hash_no_pad(pack.constants_sigmas.data(), pack.constants_sigmas.size(), vh);
        const u64 shape[12] = {degree_bits, NW, R, pack.num_public_inputs, pack.num_selectors, (u64)pack.gates.size(), pack.zero_knowledge,
                               0x51504342ull /* "BCPQ": built by cb::Builder */, vh[0], vh[1], vh[2], vh[3]};
        hash_no_pad(shape, 12, pack.circuit_digest);
    }
    return pack.validate();
}

}  // namespace cb
