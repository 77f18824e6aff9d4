// gadget_circuits.cpp — small circuits over the native builder's gadgets, one family at a time, plus random programs over all of
// them: TEST HOOKS of the library (tests/test_builder_gadgets*.py, tests/soak/fuzz_gadget_programs.py, tools/dead_cell_lint.py), kept
// apart from the circuits the provers use (leaf_circuit.cpp, wrapper_circuit.cpp). Declared in include/qpgpu_batch.h.
#include <array>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include "../../include/qpgpu.h"
#include "../../include/qpgpu_batch.h"
#include "builder.hpp"
#include "gadgets.hpp"

using cb::BoolTarget;
using cb::Builder;
using cb::HashOutTarget;
using cb::Target;
using gadgets::Digest;
using gadgets::sort_digests4;
using gadgets::digest_eq;
using gl::u64;

extern "C" {

int qpgpu_builder_sort_gate_cost(unsigned n, unsigned num_routed_wires, size_t *gates, char *err) {
    if (err) err[0] = 0;
    if (!gates || n > 4096) { if (err) std::snprintf(err, QPGPU_BATCH_ERR_CAP, "builder_sort_gate_cost: bad argument"); return QPGPU_EINVAL; }
    try {
        cb::Config cfg;
        cfg.num_routed_wires = num_routed_wires;
        Builder b(cfg);
        std::vector<Digest> values(n);
        for (Digest &d : values) for (Target &t : d) t = b.add_virtual_target();
        const size_t before = b.num_gates();
        (void)sort_digests4(b, values);
        *gates = b.num_gates() - before;
    } catch (const std::exception &e) {
        if (err) std::snprintf(err, QPGPU_BATCH_ERR_CAP, "builder_sort_gate_cost: %s", e.what());
        return QPGPU_EINVAL;
    }
    return QPGPU_OK;
}

int qpgpu_builder_gadget_circuit(unsigned kind, uint64_t *pack_out, size_t pack_cap_words, size_t *pack_words, uint64_t *cells_out, size_t cells_cap,
                                 size_t *n_inputs, size_t *n_outputs, char *err) {
    auto fail = [&](int code, const std::string &m) { if (err) std::snprintf(err, QPGPU_BATCH_ERR_CAP, "%s", m.c_str()); return code; };
    if (err) err[0] = 0;
    if (!pack_words || !n_inputs || !n_outputs) return fail(QPGPU_EINVAL, "builder_gadget_circuit: null argument");
    try {
        cb::Config cfg;
        Builder b(cfg);
        std::vector<Target> in, out;
        auto input = [&]() { const Target t = b.add_virtual_target(); in.push_back(t); return t; };
        auto input_ext = [&]() { cb::ExtTarget e; e.t[0] = input(); e.t[1] = input(); return e; };
        auto output_ext = [&](cb::ExtTarget e) { out.push_back(e.t[0]); out.push_back(e.t[1]); };
        if (kind == 0) {
            const cb::ExtTarget x = input_ext(), y = input_ext(), z = input_ext();
            output_ext(b.mul_ext(x, y)); output_ext(b.mul_add_ext(x, y, z)); output_ext(b.sub_ext(x, y)); output_ext(b.div_ext(x, y));
        } else if (kind == 1) {
            const cb::ExtTarget alpha = input_ext();
            std::vector<Target> base; std::vector<cb::ExtTarget> ext;
            for (int i = 0; i < 100; i++) base.push_back(input());
            for (int i = 0; i < 40; i++) ext.push_back(input_ext());
            output_ext(b.reduce_base(alpha, base)); output_ext(b.reduce_ext(alpha, ext));
        } else if (kind == 2) {
            const Target shift = input();
            std::vector<cb::ExtTarget> v;
            for (int i = 0; i < 16; i++) v.push_back(input_ext());
            const cb::ExtTarget pt = input_ext();
            output_ext(b.interpolate_coset(4, shift, v, pt));
        } else if (kind == 3) {
            const Target x = input(), y = input();
            const std::vector<BoolTarget> xb = b.split_le(x, 10);
            out.push_back(b.exp_from_bits_const_base(7, xb));
            out.push_back(b.le_sum(xb));
            for (const BoolTarget &bit : b.split_le(y, 64)) out.push_back(bit.target);
        } else if (kind == 4) {
            const Target idx = input();
            std::vector<Target> v;
            for (int i = 0; i < 16; i++) v.push_back(input());
            const Target sel = input(), u = input(), w = input();
            b.assert_bool({sel});
            out.push_back(b.random_access(idx, v));
            out.push_back(b.select({sel}, u, w));
            out.push_back(b.is_equal(u, w).target);
        } else if (kind == 5) {
            std::vector<Digest> ds(5);
            for (Digest &d : ds) for (Target &t : d) t = input();
            for (const Digest &d : sort_digests4(b, ds)) for (Target t : d) out.push_back(t);
            out.push_back(digest_eq(b, ds[0], ds[1]).target);
        } else if (kind == 6) {
            // common/src/gadgets.rs:343-391: is_const_less_than at widths 8, 1 and 64 (the canonical-half path), as public booleans
            const Target r8 = input(), r1 = input(), x = input();
            out.push_back(gadgets::is_const_less_than(b, 3, r8, 8).target);
            out.push_back(gadgets::is_const_less_than(b, 0, r1, 1).target);
            for (u64 left : {(u64)0, (u64)1, (u64)(gl::P - 2), (u64)(gl::P - 1)}) out.push_back(gadgets::is_const_less_than(b, left, x, 64).target);
        } else if (kind == 7) {
            // gadgets.rs:393-412: 0 < right FORCED true at width 64 — unsatisfiable for right = 0 (no 64-bit alias of zero is admitted)
            const Target right = input();
            const BoolTarget lt = gadgets::is_const_less_than(b, 0, right, 64);
            b.connect(lt.target, b._true().target);
            out.push_back(b.add_const(right, 0));
        } else if (kind == 8) {
            // gadgets.rs:414-421: a comparison width above 64 is refused when the circuit is built
            const Target right = input();
            out.push_back(gadgets::is_const_less_than(b, 0, right, 65).target);
        } else if (kind >= 3000 && kind < 3000 + 4096) {
            // a stand-in inner circuit of kind - 3000 unconstrained public inputs (the role test-helpers' build_fake_leaf_circuit plays for
            // the leaf, wormhole/tests/test-helpers/src/lib.rs:281-340, at any public-input length): proofs of it carry whatever a test puts in
            for (unsigned i = 0; i < kind - 3000; i++) out.push_back(input());
        } else if (kind >= 1000) {
            // a random program over the builder's gadgets (seed = kind - 1000): 6 inputs, ~60 operations drawn from the base and
            // extension arithmetic, bits, selection, hashing and the recursion gadgets, each consuming earlier values; every sixth value
            // an output. For differential tests of builder + stage s1 + prover on circuits nobody wrote by hand.
            uint64_t st = 0x9E3779B97F4A7C15ull * (kind - 999);
            auto next = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
            std::vector<Target> vals;
            for (int i = 0; i < 6; i++) vals.push_back(input());
            for (int i = 0; i < 6; i++) vals.push_back(b.mul_add(vals[i], vals[(i + 1) % 6], vals[(i + 2) % 6]));      // every input sits in a gate
            auto pick = [&]() { return vals[next() % vals.size()]; };
            auto pick_ext = [&]() { cb::ExtTarget e; e.t[0] = pick(); e.t[1] = pick(); return e; };
            const int ops = 40 + (int)(next() % 40);
            for (int i = 0; i < ops; i++) {
                const unsigned op = (unsigned)(next() % 14);
                switch (op) {
                case 0: vals.push_back(b.mul(pick(), pick())); break;
                case 1: vals.push_back(b.add(pick(), pick())); break;
                case 2: vals.push_back(b.sub(pick(), pick())); break;
                case 3: vals.push_back(b.mul_const_add(next() % gl::P, pick(), pick())); break;
                case 4: { const BoolTarget e = b.is_equal(pick(), pick()); vals.push_back(b.select(e, pick(), pick())); vals.push_back(e.target); break; }
                case 5: { const cb::ExtTarget r = b.mul_add_ext(pick_ext(), pick_ext(), pick_ext()); vals.push_back(r.t[0]); vals.push_back(r.t[1]); break; }
                case 6: { cb::ExtTarget d = pick_ext(); d.t[0] = b.add_const(b.mul(d.t[0], d.t[0]), 1);          // a denominator that is not zero: (x^2 + 1, y) has norm x^4 + 2 x^2 + 1 - 7 y^2
                          const cb::ExtTarget r = b.div_ext(pick_ext(), d); vals.push_back(r.t[0]); vals.push_back(r.t[1]); break; }
                case 7: { Target lo, hi; b.split_low_high(pick(), 32, 64, lo, hi); vals.push_back(lo); vals.push_back(hi); break; }
                case 8: { const std::vector<BoolTarget> bits = b.split_le(pick(), 64); vals.push_back(b.le_sum(std::vector<BoolTarget>(bits.begin(), bits.begin() + 5)));
                          vals.push_back(b.exp_from_bits_const_base(3 + next() % 100, std::vector<BoolTarget>(bits.begin() + 5, bits.begin() + 12))); break; }
                case 9: { std::vector<Target> h; const int cnt = 1 + (int)(next() % 11); for (int k = 0; k < cnt; k++) h.push_back(pick());
                          const HashOutTarget d = (next() & 1) ? b.hash_n_to_hash_no_pad(h) : b.hash_n_to_hash_no_pad_p2(h); for (Target t : d.elements) vals.push_back(t); break; }
                case 10: { const std::vector<BoolTarget> bits = b.split_le(pick(), 64); std::vector<Target> list; for (int k = 0; k < 8; k++) list.push_back(pick());
                           vals.push_back(b.random_access(b.le_sum(std::vector<BoolTarget>(bits.begin(), bits.begin() + 3)), list)); break; }
                case 11: { std::vector<Target> t; const int cnt = 1 + (int)(next() % 60); for (int k = 0; k < cnt; k++) t.push_back(pick());
                           const cb::ExtTarget r = b.reduce_base(pick_ext(), t); vals.push_back(r.t[0]); vals.push_back(r.t[1]); break; }
                case 12: { std::vector<cb::ExtTarget> v; for (int k = 0; k < 16; k++) v.push_back(pick_ext());
                           const Target sq = pick(); const Target shift = b.add_const(b.mul(sq, sq), 1);     // never zero for the tests' inputs: the pool holds zeros (x - x, a false
                                                                                                                    // is_equal), and a zero shift has no interpolant (upstream's generator panics on it)
                           const cb::ExtTarget r = b.interpolate_coset(4, shift, v, pick_ext()); vals.push_back(r.t[0]); vals.push_back(r.t[1]); break; }
                default: { std::vector<Digest> ds(3); for (Digest &d : ds) for (Target &t : d) t = pick();
                           for (const Digest &d : sort_digests4(b, ds)) vals.push_back(d[0]); break; }
                }
            }
            for (size_t i = 6; i < vals.size(); i += 6) out.push_back(vals[i]);
            out.push_back(vals.back());
        } else return fail(QPGPU_EINVAL, "builder_gadget_circuit: unknown kind");
        // the outputs are public inputs (so that each sits in a gate and has a cell); every input is consumed by its gadget
        for (Target t : out) b.register_public_input(t);
        CircuitPack pack;
        const std::string why = b.build(pack);
        if (!why.empty()) return fail(QPGPU_EINVAL, "builder_gadget_circuit: " + why);
        const std::vector<uint64_t> words = pack.serialize();
        *pack_words = words.size(); *n_inputs = in.size(); *n_outputs = out.size();
        if (pack_out) {
            if (pack_cap_words < words.size()) return fail(QPGPU_EBUFSIZE, "builder_gadget_circuit: pack buffer too small");
            std::memcpy(pack_out, words.data(), words.size() * 8);
        }
        if (cells_out) {
            if (cells_cap < in.size() + out.size()) return fail(QPGPU_EBUFSIZE, "builder_gadget_circuit: cell buffer too small");
            size_t k = 0;
            for (Target t : in) { const u64 c = b.cell_of(t); if (c == cb::NO_CELL) return fail(QPGPU_EINVAL, "builder_gadget_circuit: an input touches no gate"); cells_out[k++] = c; }
            for (Target t : out) { const u64 c = b.cell_of(t); if (c == cb::NO_CELL) return fail(QPGPU_EINVAL, "builder_gadget_circuit: an output touches no gate"); cells_out[k++] = c; }
        }
    } catch (const std::exception &e) {
        return fail(QPGPU_EINVAL, std::string("builder_gadget_circuit: ") + e.what());
    }
    return QPGPU_OK;
}


}  // extern "C"
