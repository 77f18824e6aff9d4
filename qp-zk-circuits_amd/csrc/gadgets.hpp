// gadgets.hpp — the reference's shared circuit gadgets (common/src/gadgets.rs) on the native builder: constant / variable comparisons,
// canonical 32-bit half-limb splits, digest equality and the sorting network the private batch layer orders its nullifiers with.
// Gate for gate what the Rust builds: the leaf circuit (csrc/leaf_circuit.cpp), the two batch circuits (csrc/wrapper_circuit.cpp) and the
// gadget test circuits (qpgpu_builder_gadget_circuit) share these definitions.
#pragma once
#include <array>
#include <stdexcept>
#include <vector>
#include "builder.hpp"

namespace gadgets {

using cb::BoolTarget;
using cb::Builder;
using cb::Target;
using Digest = std::array<Target, 4>;

// bytes_digest_eq (common/src/gadgets.rs:144-157): limb-wise is_equal, and-ed pairwise
inline BoolTarget digest_eq(Builder &b, const Digest &a, const Digest &c) {
    const BoolTarget e0 = b.is_equal(a[0], c[0]), e1 = b.is_equal(a[1], c[1]), e2 = b.is_equal(a[2], c[2]), e3 = b.is_equal(a[3], c[3]);
    const BoolTarget e01 = b.and_(e0, e1), e23 = b.and_(e2, e3);
    return b.and_(e01, e23);
}

// u32_lt (gadgets.rs:187-199): x < y for range-checked 32-bit values; bit 32 of x + 2^32 - y is x >= y
inline BoolTarget u32_lt(Builder &b, Target x, Target y) {
    const Target t = b.sub(b.add(x, b.constant(1ull << 32)), y);
    Target low, ge;
    b.split_low_high(t, 32, 33, low, ge);
    return b.not_({ge});
}
// split_canonical_u32_halves (gadgets.rs:211-226): (lo, hi) of the CANONICAL representative — the region hi == 2^32 - 1 && lo >= 1
// (the integers >= p) is excluded
inline void split_canonical_u32_halves(Builder &b, Target x, Target &lo, Target &hi) {
    b.split_low_high(x, 32, 64, lo, hi);
    const BoolTarget hi_is_max = b.is_equal(hi, b.constant((1ull << 32) - 1));
    const Target zero = b.zero();
    const BoolTarget lo_is_zero = b.is_equal(lo, zero);
    const BoolTarget in_wraparound = b.and_(hi_is_max, b.not_(lo_is_zero));
    b.connect(in_wraparound.target, zero);
}
// halves8_lt (gadgets.rs:239-254): lexicographic lhs < rhs over 8 half-limbs, most significant first
inline BoolTarget halves8_lt(Builder &b, const std::array<Target, 8> &lhs, const std::array<Target, 8> &rhs) {
    BoolTarget lt = b._false();
    for (int i = 7; i >= 0; i--) {
        const BoolTarget lt_i = u32_lt(b, lhs[i], rhs[i]);
        const BoolTarget eq_i = b.is_equal(lhs[i], rhs[i]);
        const BoolTarget carry = b.and_(eq_i, lt);
        lt = b.or_(lt_i, carry);
    }
    return lt;
}
// sort_digests4 (gadgets.rs:285-334): odd-even transposition network over digests split once into canonical 32-bit halves
inline std::vector<Digest> sort_digests4(Builder &b, const std::vector<Digest> &values) {
    const size_t n = values.size();
    if (n <= 1) return values;
    std::vector<std::array<Target, 8>> v(n);
    for (size_t i = 0; i < n; i++)
        for (int j = 0; j < 4; j++) { Target lo, hi; split_canonical_u32_halves(b, values[i][j], lo, hi); v[i][2 * j] = hi; v[i][2 * j + 1] = lo; }
    for (size_t round = 0; round < n; round++)
        for (size_t i = round % 2; i + 1 < n; i += 2) {
            const std::array<Target, 8> lhs = v[i], rhs = v[i + 1];
            const BoolTarget lhs_lt = halves8_lt(b, lhs, rhs);
            for (int j = 0; j < 8; j++) { v[i][j] = b.select(lhs_lt, lhs[j], rhs[j]); v[i + 1][j] = b.select(lhs_lt, rhs[j], lhs[j]); }
        }
    std::vector<Digest> out(n);
    for (size_t i = 0; i < n; i++) for (int j = 0; j < 4; j++) out[i][j] = b.mul_const_add(1ull << 32, v[i][2 * j], v[i][2 * j + 1]);
    return out;
}

// xor (gadgets.rs:116-134): a + b - 2ab
inline BoolTarget gadget_xor(Builder &b, BoolTarget x, BoolTarget y) {
    const Target ab = b.mul(x.target, y.target);
    const Target two_ab = b.mul_const(2, ab);
    const Target a_plus_b = b.add(x.target, y.target);
    return {b.sub(a_plus_b, two_ab)};
}
// assert_comparison_width (gadgets.rs): the width is 1..64 bits and holds the constant
inline void assert_comparison_width(gl::u64 left, unsigned n_log) {
    if (n_log == 0) throw std::logic_error("comparison width must be greater than zero");
    if (n_log > 64) throw std::logic_error("comparison width " + std::to_string(n_log) + " exceeds 64 bits");
    if (n_log < 64 && (left >> n_log) != 0) throw std::logic_error("constant does not fit the comparison width");
}
// is_const_less_than (gadgets.rs:40-78): left < right for a constant left; range-constrains right to n_log bits. Widths up to 63 compare
// the bits of split_le (unique: 2^n_log < p); width 64 goes through the canonical half split, so that the alias x + p of a small x cannot
// be witnessed (gadgets.rs:80-97)
inline BoolTarget is_const_less_than(Builder &b, gl::u64 left, Target right, unsigned n_log) {
    assert_comparison_width(left, n_log);
    if (n_log == 64) {
        Target right_lo, right_hi;
        split_canonical_u32_halves(b, right, right_lo, right_hi);
        const Target left_lo = b.constant(left & 0xFFFFFFFFull), left_hi = b.constant(left >> 32);
        const BoolTarget hi_lt = u32_lt(b, left_hi, right_hi);
        const BoolTarget lo_lt = u32_lt(b, left_lo, right_lo);
        const BoolTarget hi_eq = b.is_equal(left_hi, right_hi);
        const BoolTarget lo_lt_and_hi_eq = b.and_(hi_eq, lo_lt);
        return b.or_(hi_lt, lo_lt_and_hi_eq);
    }
    const std::vector<BoolTarget> right_bits = b.split_le(right, n_log);
    BoolTarget lt = b._false(), eq = b._true();
    for (int i = (int)n_log - 1; i >= 0; i--) {
        const BoolTarget a = b.constant_bool((left >> i) & 1);
        const BoolTarget bb = right_bits[i];
        const BoolTarget not_a = b.not_(a);
        const BoolTarget not_a_and_b = b.and_(not_a, bb);
        const BoolTarget this_lt = b.and_(not_a_and_b, eq);
        lt = b.or_(lt, this_lt);
        const BoolTarget a_xor_b = gadget_xor(b, a, bb);
        const BoolTarget not_xor = b.not_(a_xor_b);
        eq = b.and_(eq, not_xor);
    }
    return lt;
}
// enforce_target_less_than_const (gadgets.rs:99-114)
inline void enforce_target_less_than_const(Builder &b, Target target, gl::u64 upper_bound_exclusive, unsigned n_log) {
    if (upper_bound_exclusive == 0) throw std::logic_error("exclusive upper bound must be greater than zero");
    const BoolTarget overflow = is_const_less_than(b, upper_bound_exclusive - 1, target, n_log);
    b.connect(overflow.target, b.zero());
}

}  // namespace gadgets
