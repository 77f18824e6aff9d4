// synth.cpp — synthetic, satisfied plonky2-shaped circuits for tests and bench ("shape-equivalent
// synthetic"): the Rust CircuitBuilder that would export the real Wormhole circuit pack
// cannot run in this image, so this generator produces a circuit with the same shape parameters
// (135 wires, 80 routed, 2 constants, standard_recursion_config FRI) and a witness that satisfies it:
// a PublicInputGate row, ConstantGate rows, ArithmeticGate rows wired together by copy constraints, optional
// BaseSumGate<2> / PoseidonGate / ArithmeticExtensionGate / MulExtensionGate rows, NoopGate padding. Gates are sorted
// and grouped into selector polynomials with the builder's own rule. It plays the role of reference rows a1/a6
// (witness + circuit shape providers).
#include <algorithm>
#include <map>
#include <numeric>
#include <string>
#include "circuit.hpp"
#include "ctx.hpp"
#include "gl64.hpp"
#include "poseidon.hpp"

using gl::e2;
using gl::u64;

namespace {
struct SplitMix {
    u64 s;
    u64 next() { u64 z = (s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
    u64 felt() { for (;;) { u64 v = next(); if (v < gl::P) return v; } }
    u64 below(u64 n) { return next() % n; }
};
void host_hash_no_pad(const u64 *in, size_t n, u64 out[4]) {
    u64 st[12] = {0};
    for (size_t i = 0; i < n; i += 8) {
        size_t len = std::min<size_t>(8, n - i);
        for (size_t k = 0; k < len; k++) st[k] = gl::canon(in[i + k]);
        hasher::host_permute(st);
    }
    for (int i = 0; i < 4; i++) out[i] = st[i];
}
struct GateSpec { uint64_t type, p0, p1, degree, ncons; std::string id; uint64_t p2 = 0; };
}  // namespace

// Wire layout of the synthetic circuits' Poseidon2 gate: the default (upstream PoseidonGate's layout carried over), or with
// flag bit 7 a deliberately different one — outputs first, no swap / delta wires, other block order, 118 constraints — which
// exists to prove that prover, generator and verifiers really read the layout table instead of assuming the default.
P2GateLayout synth_p2_layout(unsigned flags) {
    P2GateLayout l;
    if (flags & 128) {
        l.w_output = 0; l.w_input = 12; l.w_swap = P2GateLayout::NO_SWAP; l.w_delta = 0;
        l.w_partial = 24; l.w_full1 = 46; l.w_full0 = 94; l.first_round_wires = 0; l.end_wire = 130;
    }
    return l;
}

// The application hashes of the Wormhole leaf circuit, as preimage lengths: unspendable account
// H(H(7)) (wormhole/circuit/src/unspendable_account.rs:215-237), nullifier H(H(9)) (nullifier.rs:285-325), the ZK-tree leaf
// H(8) (zk_merkle_proof.rs:486-504), the block header H(45) (block_header/mod.rs:61-108) and one H(16) per level of the 4-ary
// Merkle walk, MAX_DEPTH = 16 of them (zk_merkle_proof.rs:515-618): 61 permutations with the pad-10 sponge of rate 8. A
// synthetic circuit takes as many of them, in this order, as its rows have room for.
static const unsigned P2_SITE_LENS[] = {7, 4, 9, 4, 8, 45, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16};
// Hash site h of a synthetic circuit: `blocks` Poseidon2-gate rows at rows 8 * (slot + k) + 3, k < blocks; between two of them
// the ArithmeticGate row 8 * (slot + k) + 4 whose operations j < 8 add block k + 1 into the rate part (prev_out[j] * 1 + m).
// Message element i sits in input wire i of the first gate row (i < 8) or in the addend wire (4 j + 2, j = i % 8) of the add
// row in front of block i / 8; the digest is the first four output wires of the last gate row.
struct P2Site { unsigned len, blocks, slot; };
std::vector<P2Site> synth_p2_sites(unsigned degree_bits, unsigned num_public_inputs, unsigned flags) {
    std::vector<P2Site> out;
    if (!(flags & 64)) return out;
    const u64 n = 1ull << degree_bits, n_noop = std::max<u64>(1, n / 16);
    const u64 slots = (n - n_noop) / 8 > 1 ? (n - n_noop - 4) / 8 : 0;     // slot s uses rows 8 s + 3 and 8 s + 4
    // rows 3 .. 3 + ceil(num_public_inputs / 8) belong to the public-input hash: the first slot lies behind them
    unsigned next = std::max<unsigned>(1, ((num_public_inputs + 7) / 8 + 7) / 8);
    for (unsigned len : P2_SITE_LENS) {
        const unsigned blocks = (len + 1 + 7) / 8;
        if (next + blocks > slots) break;
        out.push_back({len, blocks, next});
        next += blocks;
    }
    return out;
}

// Gate list of a synthetic circuit: sorted by (degree, id) and grouped into selector polynomials the way
// CircuitBuilder::build does (a group holds gates while size + degree < max_degree; one group if everything fits).
std::string synth_gate_layout(unsigned num_wires, unsigned num_routed, unsigned flags, std::vector<GateInfo> &gates, u64 &num_selectors) {
    const bool with_poseidon = (flags & 1) != 0, with_base_sum = (flags & 2) != 0, with_ext = (flags & 4) != 0, with_rec = (flags & 8) != 0;
    const bool with_p2 = (flags & 64) != 0;
    if (with_rec && (num_routed < 48 || num_wires < 64)) return "recursion gates need at least 48 routed wires and 64 wires";
    const u64 num_limbs = std::min<u64>(63, num_routed - 1), num_ops = num_routed / 4;
    const u64 ext_ops = num_routed / 8, mul_ops = num_routed / 6;
    std::vector<GateSpec> gs = {
        {GATE_NOOP, 0, 0, 0, 0, "NoopGate"},
        {GATE_CONSTANT, 2, 0, 1, 2, "ConstantGate { num_consts: 2 }"},
        {GATE_PUBLIC_INPUT, 0, 0, 1, 4, "PublicInputGate"},
        {GATE_ARITHMETIC, num_ops, 0, 3, num_ops, "ArithmeticGate { num_ops: " + std::to_string(num_ops) + " }"},
    };
    if (with_base_sum) gs.push_back({GATE_BASE_SUM, num_limbs, 2, 2, num_limbs + 1, "BaseSumGate { num_limbs: " + std::to_string(num_limbs) + " } + Base: 2"});
    if (with_poseidon) gs.push_back({GATE_POSEIDON, 0, 0, 7, 123, "PoseidonGate(PhantomData<plonky2_field::goldilocks_field::GoldilocksField>)<WIDTH=12>"});
    if (with_p2) {
        const P2GateLayout lay = synth_p2_layout(flags);
        gs.push_back({GATE_POSEIDON2, 0, 0, 7, lay.num_constraints(), "Poseidon2Gate(PhantomData<plonky2_field::goldilocks_field::GoldilocksField>)<WIDTH=12>"});
    }
    if (with_ext) {
        gs.push_back({GATE_ARITHMETIC_EXT, ext_ops, 0, 3, 2 * ext_ops, "ArithmeticExtensionGate { num_ops: " + std::to_string(ext_ops) + " }"});
        gs.push_back({GATE_MUL_EXT, mul_ops, 0, 3, 2 * mul_ops, "MulExtensionGate { num_ops: " + std::to_string(mul_ops) + " }"});
    }
    if (with_rec) {
        // parameters as the gates' new_from_config / max_coeffs_len pick them for this (num_wires, num_routed)
        const u64 red = std::min<u64>((num_wires - 6) / 3, num_routed - 6), redx = std::min<u64>((num_wires - 6) / 4, (num_routed - 6) / 2);
        const u64 ra_bits = 4, ra_copies = std::min<u64>(num_routed / 18, num_wires / 22), ra_extra = std::min<u64>(num_routed - 18 * ra_copies, 2);
        const u64 exp_bits = std::min<u64>(num_routed - 2, (num_wires - 2) / 2);
        gs.push_back({GATE_REDUCING, red, 0, 2, 2 * red, "ReducingGate { num_coeffs: " + std::to_string(red) + " }", 0});
        gs.push_back({GATE_REDUCING_EXT, redx, 0, 2, 2 * redx, "ReducingExtensionGate { num_coeffs: " + std::to_string(redx) + " }", 0});
        gs.push_back({GATE_RANDOM_ACCESS, ra_bits, ra_copies, ra_bits + 1, ra_copies * (ra_bits + 2) + ra_extra, "RandomAccessGate { bits: 4, num_copies: " + std::to_string(ra_copies) + " }", ra_extra});
        gs.push_back({GATE_EXPONENTIATION, exp_bits, 0, 4, exp_bits + 1, "ExponentiationGate { num_power_bits: " + std::to_string(exp_bits) + " }", 0});
        // CosetInterpolationGate::with_max_degree(4, max_quotient_degree_factor = 8): 2 intermediates, degree 6
        { const u64 np = 16, max_degree = 8, ni = (np - 2) / (max_degree - 1), deg = (np - 2) / (ni + 1) + 2;
          gs.push_back({GATE_COSET_INTERPOLATION, 4, deg, deg, 2 * (2 + 2 * ((np - 2) / (deg - 1))), "CosetInterpolationGate { subgroup_bits: 4, degree: " + std::to_string(deg) + " }", 0}); }
        gs.push_back({GATE_POSEIDON_MDS, 0, 0, 1, 24, "PoseidonMdsGate(PhantomData<plonky2_field::goldilocks_field::GoldilocksField>)<WIDTH=12>", 0});
    }
    std::sort(gs.begin(), gs.end(), [](const GateSpec &a, const GateSpec &b) { return a.degree != b.degree ? a.degree < b.degree : a.id < b.id; });
    const u64 max_degree = 9;   // quotient_degree_factor + 1
    std::vector<std::pair<u64, u64>> groups;
    if (gs.back().degree + gs.size() - 1 <= max_degree) groups.push_back({0, gs.size()});
    else {
        for (size_t start = 0; start < gs.size();) {
            size_t size = 0;
            while (start + size < gs.size() && size + gs[start + size].degree < max_degree) size++;
            if (size == 0) return "gate degree too high for the quotient degree";
            groups.push_back({start, start + size});
            start += size;
        }
    }
    gates.clear();
    for (size_t i = 0; i < gs.size(); i++) {
        size_t grp = 0;
        while (!(groups[grp].first <= i && i < groups[grp].second)) grp++;
        gates.push_back({gs[i].type, gs[i].p0, gs[i].p1, grp, groups[grp].first, groups[grp].second, gs[i].ncons, gs[i].p2});
    }
    num_selectors = groups.size();
    return "";
}

void synth_public_inputs_hash(const u64 *pis, size_t n, u64 out[4]) { host_hash_no_pad(pis, n, out); }

// flags: bit 0 PoseidonGate rows, bit 1 BaseSumGate<2> rows, bit 2 ArithmeticExtension + MulExtension rows,
// bit 3 Reducing / ReducingExtension / RandomAccess / Exponentiation / PoseidonMds / CosetInterpolation rows (the
// recursive verifier's set), bit 4 free-standing witness hints (Equality / LowHigh / NonzeroTest / Constant / Copy outputs
// feeding arithmetic operations, WireSplit feeding BaseSum rows, extension quotients feeding extension multiplications):
// the pack then carries a hint trailer and those cells are produced by stage s1 instead of being caller inputs.
// Builds the pack and the witness. wires: num_wires x n column-major. Returns "" or an error.
std::string synth_build(unsigned degree_bits, unsigned num_wires, unsigned num_routed, unsigned num_public_inputs,
                        u64 seed, unsigned flags, CircuitPack &pack, std::vector<u64> &wires, std::vector<u64> &pis) {
    const bool with_poseidon = (flags & 1) != 0, with_base_sum = (flags & 2) != 0, with_ext = (flags & 4) != 0, with_rec = (flags & 8) != 0;
    const bool with_hints = (flags & 16) != 0, with_p2 = (flags & 64) != 0;
    if (degree_bits < 3 || degree_bits > 20) return "degree_bits out of range";
    if (with_p2 && (num_wires < 135 || num_routed < 40 || degree_bits < 5)) return "poseidon2 gates need 135 wires, 40 routed wires and 32 rows";
    if (num_routed < 8 || num_routed > num_wires || num_routed % 4) return "num_routed_wires must be a multiple of 4, >= 8";
    if (with_poseidon && (num_wires < 135 || num_routed < 28)) return "poseidon gates need 135 wires";
    SplitMix rng{seed ^ 0x5EED5EED5EEDull};
    const u64 n = 1ull << degree_bits;
    const u64 num_limbs = std::min<u64>(63, num_routed - 1), num_ops = num_routed / 4;
    const u64 ext_ops = num_routed / 8, mul_ops = num_routed / 6;

    std::vector<GateInfo> layout;
    u64 n_groups = 0;
    { std::string e = synth_gate_layout(num_wires, num_routed, flags, layout, n_groups); if (!e.empty()) return e; }
    pack = CircuitPack();
    pack.degree_bits = degree_bits; pack.num_wires = num_wires; pack.num_routed_wires = num_routed;
    pack.num_constants = 2; pack.num_selectors = n_groups; pack.num_challenges = 2; pack.quotient_degree_factor = 8;
    pack.num_partial_products = (num_routed + 7) / 8 - 1; pack.num_public_inputs = num_public_inputs;
    pack.rate_bits = 3; pack.cap_height = 4; pack.proof_of_work_bits = 16; pack.num_query_rounds = 28;
    pack.zero_knowledge = 0;
    pack.arity_bits = fri_reduction_arity_bits(degree_bits, 3, 4, 4, 5);
    pack.num_gate_constraints = 0;
    if (with_p2) { pack.p2_layout = synth_p2_layout(flags); pack.has_p2_layout = true; }
    const P2GateLayout &P2L = pack.p2_layout;
    uint64_t idx_of[16] = {0}, sel_of[16] = {0};
    const GateInfo *info_of[16] = {nullptr};
    pack.gates = layout;
    for (size_t i = 0; i < layout.size(); i++) {
        pack.num_gate_constraints = std::max(pack.num_gate_constraints, layout[i].num_constraints);
        idx_of[layout[i].type] = i; sel_of[layout[i].type] = layout[i].selector_index; info_of[layout[i].type] = &layout[i];
    }
    pack.k_is.resize(num_routed);
    { u64 k = 1; for (unsigned j = 0; j < num_routed; j++) { pack.k_is[j] = gl::canon(k); k = gl::mul(k, gl::MULT_GEN); } }

    pis.resize(num_public_inputs);
    for (auto &v : pis) v = rng.felt();
    u64 pih[4];
    host_hash_no_pad(pis.data(), pis.size(), pih);

    // ---- row kinds ----
    const u64 n_const_rows = 2, n_noop = std::max<u64>(1, n / 16);
    std::vector<uint8_t> row_gate(n, GATE_ARITHMETIC);
    row_gate[0] = GATE_PUBLIC_INPUT;
    for (u64 r = 1; r <= n_const_rows; r++) row_gate[r] = GATE_CONSTANT;
    for (u64 r = n - n_noop; r < n; r++) row_gate[r] = GATE_NOOP;
    if (with_poseidon) for (u64 r = 8; r + n_noop < n; r += 8) row_gate[r] = GATE_POSEIDON;      // every 8th row hashes
    if (with_base_sum) for (u64 r = 5; r + n_noop < n; r += 8) row_gate[r] = GATE_BASE_SUM;      // every 8th row range-checks
    if (with_ext) for (u64 r = 6; r + n_noop < n; r += 8) row_gate[r] = (r & 8) ? GATE_MUL_EXT : GATE_ARITHMETIC_EXT;
    if (with_rec) {
        const uint8_t cyc[6] = {GATE_REDUCING, GATE_REDUCING_EXT, GATE_RANDOM_ACCESS, GATE_EXPONENTIATION, GATE_POSEIDON_MDS, GATE_COSET_INTERPOLATION};
        for (u64 r = 7, k = 0; r + n_noop < n; r += 8, k++) row_gate[r] = cyc[k % 6];
    }

    // Poseidon2-gate rows: the leaf circuit's application hashes as sponge chains (synth_p2_sites), then a few free-standing
    // permutation rows (random swap bit where the layout has one)
    struct P2Role { int site, block; };
    std::map<u64, P2Role> p2_row, p2_add_row;      // gate row -> (site, block); add row -> (site, block it feeds)
    const std::vector<P2Site> p2_sites = synth_p2_sites(degree_bits, num_public_inputs, flags);
    if (with_p2) {
        unsigned next_slot = std::max<unsigned>(1, ((num_public_inputs + 7) / 8 + 7) / 8);
        for (size_t h = 0; h < p2_sites.size(); h++)
            for (unsigned k = 0; k < p2_sites[h].blocks; k++) {
                const u64 r = 8ull * (p2_sites[h].slot + k) + 3;
                row_gate[r] = GATE_POSEIDON2; p2_row[r] = {(int)h, (int)k};
                if (k + 1 < p2_sites[h].blocks) p2_add_row[r + 1] = {(int)h, (int)k + 1};
                next_slot = p2_sites[h].slot + k + 1;
            }
        for (unsigned extra = 0; extra < 4; extra++) {
            const u64 r = 8ull * (next_slot + extra) + 3;
            if (r + n_noop < n) { row_gate[r] = GATE_POSEIDON2; p2_row[r] = {-1, 0}; }
        }
    }

    // the public-input hash in-circuit (plonky2's CircuitBuilder::build): ceil(npis / 8) chained PoseidonGate rows right
    // after the constant rows; their last output is copy-connected to the PublicInputGate's wires. Without Poseidon rows
    // (or without room) the PublicInputGate row holds the hash as a plain source, as the first synthetic circuits did.
    std::map<u64, u64> pi_hash_row;      // row -> absorption index
    {
        const u64 chunks = (num_public_inputs + 7) / 8;
        // (the PoseidonGate rows compute plonky2's Poseidon: only when that is also the proof-system hasher does their
        // output equal the public-input hash the prover binds)
        if (with_poseidon && !(flags & 32) && hasher::kind() == hasher::POSEIDON && chunks > 0 && 1 + n_const_rows + chunks + n_noop <= n) {
            for (u64 k = 0; k < chunks; k++) { row_gate[1 + n_const_rows + k] = GATE_POSEIDON; pi_hash_row[1 + n_const_rows + k] = k; }
            pack.pi_cells.assign(num_public_inputs, 0);
        }
    }

    wires.assign((size_t)num_wires * n, 0);
    auto W = [&](u64 row, u64 col) -> u64 & { return wires[col * n + row]; };
    for (u64 c = 0; c < num_wires; c++) for (u64 r = 0; r < n; r++) W(r, c) = rng.felt();  // unconstrained cells

    const u64 ncs = pack.num_cs_cols(), sel_cols = pack.num_selectors, UNUSED = 0xFFFFFFFFull;
    pack.constants_sigmas.assign(ncs * n, 0);
    auto CS = [&](u64 row, u64 col) -> u64 & { return pack.constants_sigmas[col * n + row]; };

    // union-find over routed cells for copy constraints
    std::vector<uint32_t> parent((size_t)n * num_routed);
    std::iota(parent.begin(), parent.end(), 0u);
    auto find = [&](uint32_t x) { while (parent[x] != x) { parent[x] = parent[parent[x]]; x = parent[x]; } return x; };
    auto cell = [&](u64 row, u64 col) { return (uint32_t)(row * num_routed + col); };
    std::vector<uint32_t> pool;  // cells whose value may be copied
    // input cell (r, col): with probability 0.7 a copy of an earlier output (adds the copy constraint), else stays random
    auto input = [&](u64 r, u64 col) -> u64 {
        if (!pool.empty() && rng.below(10) < 7) {
            uint32_t src = pool[rng.below(pool.size())];
            W(r, col) = wires[(size_t)(src % num_routed) * n + src / num_routed];
            uint32_t a = find(src), b = find(cell(r, col));
            if (a != b) parent[b] = a;
        }
        return W(r, col);
    };
    auto output = [&](u64 r, u64 col, u64 v) {
        W(r, col) = gl::canon(v);
        pool.push_back(cell(r, col));
        if (pool.size() > 4096) pool.erase(pool.begin(), pool.begin() + 2048);
    };

    // hint plumbing: a pool entry as a hint cell (row * num_wires + column) and its value
    auto pool_pick = [&]() -> uint32_t { return pool[rng.below(pool.size())]; };
    auto hint_cell = [&](uint32_t pc) -> u64 { return (u64)(pc / num_routed) * num_wires + pc % num_routed; };
    auto pool_val = [&](uint32_t pc) -> u64 { return wires[(size_t)(pc % num_routed) * n + pc / num_routed]; };
    auto own_cell = [&](u64 r, u64 col) -> u64 { return r * num_wires + col; };
    auto add_hint = [&](u64 op, u64 a, u64 b, u64 c_, u64 d, u64 e, u64 f) { pack.hints.push_back({{op, a, b, c_, d, e, f, 0}}); };
    // feeds the two multiplicand cells (col, col + 1) of an arithmetic operation from a hint; returns false to use plain inputs
    auto hinted_pair = [&](u64 r, u64 col) -> bool {
        if (!with_hints || pool.size() < 8 || rng.below(8) != 0) return false;
        const uint32_t px = pool_pick();
        const u64 x = pool_val(px);
        switch (rng.below(5)) {
        case 0: {   // EqualityGenerator: equal, inv
            const uint32_t py = rng.below(4) == 0 ? px : pool_pick();
            const u64 y = pool_val(py);
            W(r, col) = x == y ? 1 : 0; W(r, col + 1) = x == y ? 0 : gl::inv(gl::sub(x, y));
            add_hint(HINT_EQUALITY, hint_cell(px), hint_cell(py), own_cell(r, col), own_cell(r, col + 1), 0, 0);
            break;
        }
        case 1: {   // LowHighGenerator
            const u64 bits = 1 + rng.below(40);
            W(r, col) = x & ((1ull << bits) - 1); W(r, col + 1) = x >> bits;
            add_hint(HINT_LOW_HIGH, hint_cell(px), own_cell(r, col), own_cell(r, col + 1), bits, 0, 0);
            break;
        }
        case 2: {   // NonzeroTestGenerator + ConstantGenerator
            const u64 cst = rng.felt();
            W(r, col) = x == 0 ? 1 : gl::inv(x); W(r, col + 1) = cst;
            add_hint(HINT_NONZERO_TEST, hint_cell(px), own_cell(r, col), 0, 0, 0, 0);
            add_hint(HINT_CONSTANT, own_cell(r, col + 1), cst, 0, 0, 0, 0);
            break;
        }
        case 3: {   // CopyGenerator twice (a copy made by a generator instead of a copy constraint)
            const uint32_t py = pool_pick();
            W(r, col) = x; W(r, col + 1) = pool_val(py);
            add_hint(HINT_COPY, own_cell(r, col), hint_cell(px), 0, 0, 0, 0);
            add_hint(HINT_COPY, own_cell(r, col + 1), hint_cell(py), 0, 0, 0, 0);
            break;
        }
        default: {  // equality of a value with itself: equal = 1, inv = 0
            W(r, col) = 1; W(r, col + 1) = 0;
            add_hint(HINT_EQUALITY, hint_cell(px), hint_cell(px), own_cell(r, col), own_cell(r, col + 1), 0, 0);
            break;
        }
        }
        for (u64 k = 0; k < 2; k++) W(r, col + k) = gl::canon(W(r, col + k));
        return true;
    };

    // fills the swap bit, deltas and recorded S-box inputs of a PoseidonGate row; st receives the permutation output
    auto poseidon_row = [&](u64 r, const u64 (&in)[12], u64 swap, u64 (&st)[12]) {
        const u64 *rcs = poseidon::host_round_constants();
        W(r, 24) = swap;
        for (int i = 0; i < 4; i++) {
            u64 delta = swap ? gl::canon(gl::sub(in[i + 4], in[i])) : 0;
            W(r, 25 + i) = delta;
            st[i] = gl::canon(gl::add(in[i], delta)); st[i + 4] = gl::canon(gl::sub(in[i + 4], delta));
        }
        for (int i = 8; i < 12; i++) st[i] = in[i];
        int rc = 0;
        for (int rr = 0; rr < 4; rr++, rc++) {
            for (int i = 0; i < 12; i++) st[i] = gl::canon(gl::add(st[i], rcs[rc * 12 + i]));
            if (rr) for (int i = 0; i < 12; i++) W(r, 29 + 12 * (rr - 1) + i) = st[i];
            for (int i = 0; i < 12; i++) st[i] = poseidon::sbox7(st[i]);
            poseidon::mds_layer(st);
            for (int i = 0; i < 12; i++) st[i] = gl::canon(st[i]);
        }
        // partial rounds in plonky2's fast basis: the S-box input wires are state[0] of that formulation
        const u64 *fpt = poseidon::host_fast_partial();
        poseidon::fast_partial_enter(st, fpt);
        for (int rr = 0; rr < 22; rr++) {
            st[0] = gl::canon(st[0]);
            W(r, 65 + rr) = st[0];
            st[0] = poseidon::sbox7(st[0]);
            poseidon::fast_partial_linear(st, fpt, rr);
        }
        for (int i = 0; i < 12; i++) st[i] = gl::canon(st[i]);
        rc += 22;
        for (int rr = 0; rr < 4; rr++, rc++) {
            for (int i = 0; i < 12; i++) st[i] = gl::canon(gl::add(st[i], rcs[rc * 12 + i]));
            for (int i = 0; i < 12; i++) W(r, 87 + 12 * rr + i) = st[i];
            for (int i = 0; i < 12; i++) st[i] = poseidon::sbox7(st[i]);
            poseidon::mds_layer(st);
            for (int i = 0; i < 12; i++) st[i] = gl::canon(st[i]);
        }
    };

    // fills the delta wires and the recorded S-box inputs of a Poseidon2-gate row (layout P2L); st receives the permutation output
    auto poseidon2_row = [&](u64 r, const u64 (&in)[12], u64 swap, u64 (&st)[12]) {
        const poseidon2::Params &P = poseidon2::qp_params();
        for (int i = 0; i < 12; i++) st[i] = in[i];
        if (P2L.has_swap()) {
            W(r, P2L.w_swap) = swap;
            for (int i = 0; i < 4; i++) {
                const u64 delta = swap ? gl::canon(gl::sub(in[i + 4], in[i])) : 0;
                W(r, P2L.w_delta + i) = delta;
                st[i] = gl::canon(gl::add(in[i], delta)); st[i + 4] = gl::canon(gl::sub(in[i + 4], delta));
            }
        }
        poseidon2::ext_layer(st, P);
        u64 rec = P2L.w_full0;
        for (int rr = 0; rr < 4; rr++) {
            for (int i = 0; i < 12; i++) st[i] = gl::canon(gl::add(st[i], P.rc_ext[rr * 12 + i]));
            if (rr || P2L.first_round_wires) { for (int i = 0; i < 12; i++) W(r, rec + i) = st[i]; rec += 12; }
            for (int i = 0; i < 12; i++) st[i] = poseidon::sbox7(st[i]);
            poseidon2::ext_layer(st, P);
        }
        for (int rr = 0; rr < 22; rr++) {
            st[0] = gl::canon(gl::add(st[0], P.rc_int[rr]));
            W(r, P2L.w_partial + rr) = st[0];
            st[0] = poseidon::sbox7(st[0]);
            poseidon2::int_layer(st, P);
        }
        for (int rr = 0; rr < 4; rr++) {
            for (int i = 0; i < 12; i++) st[i] = gl::canon(gl::add(st[i], P.rc_ext[(4 + rr) * 12 + i]));
            for (int i = 0; i < 12; i++) W(r, P2L.w_full1 + 12 * rr + i) = st[i];
            for (int i = 0; i < 12; i++) st[i] = poseidon::sbox7(st[i]);
            poseidon2::ext_layer(st, P);
        }
        for (int i = 0; i < 12; i++) st[i] = gl::canon(st[i]);
    };
    // cell (r, col) := copy of cell (sr, sc): value and copy constraint
    auto tie = [&](u64 r, u64 col, u64 sr, u64 sc) {
        W(r, col) = W(sr, sc);
        const uint32_t a = find(cell(sr, sc)), b = find(cell(r, col));
        if (a != b) parent[b] = a;
    };
    const u64 P2_ZERO_ROW = 2, P2_ZERO_COL = 0, P2_ONE_COL = 1;     // with Poseidon2 rows the second ConstantGate row holds 0 and 1

    for (u64 r = 0; r < n; r++) {
        const u64 kind = row_gate[r];
        for (u64 s = 0; s < sel_cols; s++) CS(r, s) = sel_of[kind] == s ? idx_of[kind] : UNUSED;   // selector polynomials
        if (kind == GATE_PUBLIC_INPUT) {
            for (int i = 0; i < 4; i++) W(r, i) = pih[i];
        } else if (kind == GATE_CONSTANT) {
            for (int i = 0; i < 2; i++) {
                u64 c = rng.felt();
                // builder.zero(): the constant the unused inputs of the public-input hash are wired to (kept out of the copy pool)
                if (r == 1 && i == 0 && !pi_hash_row.empty()) { CS(r, sel_cols) = 0; W(r, 0) = 0; continue; }
                // builder.zero() / builder.one() for the Poseidon2 sponges' padding, capacity and additions (kept out of the copy pool)
                if (with_p2 && r == P2_ZERO_ROW) { CS(r, sel_cols + i) = (u64)i; W(r, i) = (u64)i; continue; }
                CS(r, sel_cols + i) = c; output(r, i, c);
            }
        } else if (kind == GATE_ARITHMETIC) {
            const u64 c0 = (r & 1) ? rng.felt() : 1, c1 = (r & 2) ? rng.felt() : 1;
            CS(r, sel_cols) = c0; CS(r, sel_cols + 1) = c1;
            u64 op0 = 0;
            if (p2_add_row.count(r)) {
                // additive absorption of block k of a sponge: rate element j of the previous permutation's output plus message
                // element 8 k + j (builder.add = 1 * prev * one + 1 * m; this row's constants are 1, 1 since r = 4 mod 8)
                const P2Role role = p2_add_row[r];
                const P2Site &site = p2_sites[role.site];
                for (u64 j = 0; j < 8; j++) {
                    const u64 idx = 8ull * role.block + j;
                    tie(r, 4 * j, r - 1, P2L.w_output + j);
                    tie(r, 4 * j + 1, P2_ZERO_ROW, P2_ONE_COL);
                    if (idx < site.len) input(r, 4 * j + 2);
                    else tie(r, 4 * j + 2, P2_ZERO_ROW, idx == site.len ? P2_ONE_COL : P2_ZERO_COL);
                    W(r, 4 * j + 3) = gl::canon(gl::add(gl::mul(gl::mul(W(r, 4 * j), W(r, 4 * j + 1)), c0), gl::mul(W(r, 4 * j + 2), c1)));
                }
                op0 = 8;
            }
            for (u64 op = op0; op < num_ops; op++) {
                u64 m0, m1;
                if (hinted_pair(r, 4 * op)) { m0 = W(r, 4 * op); m1 = W(r, 4 * op + 1); }
                else { m0 = input(r, 4 * op); m1 = input(r, 4 * op + 1); }
                const u64 ad = input(r, 4 * op + 2);
                output(r, 4 * op + 3, gl::add(gl::mul(gl::mul(m0, m1), c0), gl::mul(ad, c1)));
            }
        } else if (kind == GATE_ARITHMETIC_EXT || kind == GATE_MUL_EXT) {
            // extension-field arithmetic over F[x]/(x^2-7): each operand occupies two consecutive wires
            const bool mul_only = kind == GATE_MUL_EXT;
            const u64 c0 = rng.felt(), c1 = rng.felt(), stride = mul_only ? 6 : 8, ops = mul_only ? mul_ops : ext_ops;
            CS(r, sel_cols) = c0; CS(r, sel_cols + 1) = mul_only ? 0 : c1;
            for (u64 op = 0; op < ops; op++) {
                const u64 b = stride * op;
                u64 in[6] = {0, 0, 0, 0, 0, 0};
                u64 k0 = 0;
                if (with_hints && pool.size() >= 8 && rng.below(4) == 0) {   // QuotientGeneratorExtension feeds the first multiplicand
                    const uint32_t pn0 = pool_pick(), pn1 = pool_pick(), pd0 = pool_pick(), pd1 = pool_pick();
                    e2 den = gl::e2_make(pool_val(pd0), pool_val(pd1));
                    if (gl::canon(den.a) != 0 || gl::canon(den.b) != 0) {
                        const e2 q = gl::e2_canon(gl::e2_mul(gl::e2_make(pool_val(pn0), pool_val(pn1)), gl::e2_inv(den)));
                        W(r, b) = q.a; W(r, b + 1) = q.b; in[0] = q.a; in[1] = q.b; k0 = 2;
                        add_hint(HINT_QUOTIENT_EXT, hint_cell(pn0), hint_cell(pn1), hint_cell(pd0), hint_cell(pd1), own_cell(r, b), own_cell(r, b + 1));
                    }
                }
                for (u64 k = k0; k < stride - 2; k++) in[k] = input(r, b + k);   // in wire order: the rng stream is part of the fixture
                e2 res = gl::e2_scale(gl::e2_mul(gl::e2_make(in[0], in[1]), gl::e2_make(in[2], in[3])), c0);
                if (!mul_only) res = gl::e2_add(res, gl::e2_scale(gl::e2_make(in[4], in[5]), c1));
                output(r, b + stride - 2, res.a); output(r, b + stride - 1, res.b);
            }
        } else if (kind == GATE_REDUCING || kind == GATE_REDUCING_EXT) {
            // acc_{i} = acc_{i-1} * alpha + coeff_i over the extension; the last accumulator is the output (wires 0..1)
            const bool ext = kind == GATE_REDUCING_EXT;
            const u64 nc = info_of[kind]->param0, start_accs = 6 + (ext ? 2 * nc : nc);
            u64 head[4];
            for (u64 k = 0; k < 4; k++) head[k] = input(r, 2 + k);
            const e2 alpha = gl::e2_make(head[0], head[1]);
            e2 acc = gl::e2_make(head[2], head[3]);
            for (u64 i = 0; i < nc; i++) {
                e2 cf = gl::e2_from(0);
                if (ext) { const u64 c0 = input(r, 6 + 2 * i), c1 = input(r, 7 + 2 * i); cf = gl::e2_make(c0, c1); }
                else cf = gl::e2_from(input(r, 6 + i));
                acc = gl::e2_canon(gl::e2_add(gl::e2_mul(acc, alpha), cf));
                if (i == nc - 1) { output(r, 0, acc.a); output(r, 1, acc.b); }
                else { W(r, start_accs + 2 * i) = acc.a; W(r, start_accs + 2 * i + 1) = acc.b; }
            }
        } else if (kind == GATE_RANDOM_ACCESS) {
            const GateInfo &g = *info_of[kind];
            const u64 bits = g.param0, copies = g.param1, extra = g.param2, vec = 1ull << bits, routed = (2 + vec) * copies + extra;
            for (u64 cp = 0; cp < copies; cp++) {
                const u64 b0 = (2 + vec) * cp, idx = rng.below(vec);
                W(r, b0) = idx;
                u64 chosen = 0;
                for (u64 i = 0; i < vec; i++) { const u64 v = input(r, b0 + 2 + i); if (i == idx) chosen = v; }
                output(r, b0 + 1, chosen);
                for (u64 i = 0; i < bits; i++) W(r, routed + cp * bits + i) = (idx >> i) & 1;
            }
            for (u64 i = 0; i < extra; i++) { const u64 c = rng.felt(); CS(r, sel_cols + i) = c; output(r, (2 + vec) * copies + i, c); }
        } else if (kind == GATE_EXPONENTIATION) {
            const u64 nb = info_of[kind]->param0, base = input(r, 0);
            u64 cur = 1;
            for (u64 i = 0; i < nb; i++) W(r, 1 + i) = rng.below(2);
            for (u64 i = 0; i < nb; i++) {
                const u64 prev = i == 0 ? 1 : gl::mul(cur, cur), bit = W(r, 1 + (nb - 1 - i));
                cur = gl::canon(bit ? gl::mul(prev, base) : prev);
                W(r, 2 + nb + i) = cur;
            }
            output(r, 1 + nb, cur);
        } else if (kind == GATE_POSEIDON_MDS) {
            u64 in[24];
            for (u64 k = 0; k < 24; k++) in[k] = input(r, k);
            for (int comp = 0; comp < 2; comp++) {
                u64 st[12];
                for (int i = 0; i < 12; i++) st[i] = in[2 * i + comp];
                poseidon::mds_layer(st);
                for (int i = 0; i < 12; i++) output(r, 24 + 2 * i + comp, st[i]);
            }
        } else if (kind == GATE_COSET_INTERPOLATION) {
            // interpolate 2^bits extension values given on the coset shift*H at an extension point, in chunks of `degree` points
            const GateInfo &g = *info_of[kind];
            const u64 bits = g.param0, deg = g.param1, np = 1ull << bits, ni = (np - 2) / (deg - 1);
            const u64 s_ep = 1 + 2 * np, s_ev = s_ep + 2, s_int = s_ev + 2;
            u64 shift = input(r, 0);
            if (gl::canon(shift) == 0) { shift = 1; W(r, 0) = 1; }   // (a copied zero cannot happen: pool values are outputs of random data)
            std::vector<u64> vals(2 * np);
            for (u64 k = 0; k < 2 * np; k++) vals[k] = input(r, 1 + k);
            const u64 e0 = input(r, s_ep), e1 = input(r, s_ep + 1), sinv = gl::inv(shift);
            const e2 sp = gl::e2_canon(gl::e2_scale(gl::e2_make(e0, e1), sinv));
            W(r, s_int + 4 * ni) = sp.a; W(r, s_int + 4 * ni + 1) = sp.b;
            const u64 omega = gl::root_of_unity((unsigned)bits), inv_n = gl::inv(np);
            e2 ev = gl::e2_from(0), pr = gl::e2_from(1);
            u64 x = 1, lo = 0, hi = deg;
            for (u64 c = 0; c <= ni; c++) {
                for (u64 q = lo; q < hi; q++) {
                    e2 term = sp; term.a = gl::sub(term.a, x);
                    const e2 t = gl::e2_scale(gl::e2_mul(gl::e2_make(vals[2 * q], vals[2 * q + 1]), pr), gl::mul(x, inv_n));
                    ev = gl::e2_add(gl::e2_mul(ev, term), t);
                    pr = gl::e2_mul(pr, term);
                    x = gl::mul(x, omega);
                }
                ev = gl::e2_canon(ev); pr = gl::e2_canon(pr);
                if (c == ni) break;
                W(r, s_int + 2 * c) = ev.a; W(r, s_int + 2 * c + 1) = ev.b;
                W(r, s_int + 2 * (ni + c)) = pr.a; W(r, s_int + 2 * (ni + c) + 1) = pr.b;
                lo = 1 + (deg - 1) * (c + 1); hi = std::min<u64>(lo + deg - 1, np);
            }
            output(r, s_ev, ev.a); output(r, s_ev + 1, ev.b);
        } else if (kind == GATE_BASE_SUM) {
            // BaseSumGate<2> row: a value below 2^num_limbs and its bits
            u64 v = rng.next() & ((1ull << num_limbs) - 1);
            if (with_hints && pool.size() >= 8 && rng.below(2) == 0) {   // WireSplitGenerator: this gate's chunk of an earlier value
                const uint32_t px = pool_pick();
                const u64 shift = rng.below(64 - num_limbs + 1);
                v = (pool_val(px) >> shift) & ((1ull << num_limbs) - 1);
                add_hint(HINT_WIRE_SPLIT, hint_cell(px), own_cell(r, 0), shift, num_limbs, 0, 0);
            }
            output(r, 0, v);
            for (u64 i = 0; i < num_limbs; i++) W(r, 1 + i) = (v >> i) & 1;
        } else if (kind == GATE_POSEIDON2) {
            u64 in[12], st[12];
            const P2Role role = p2_row[r];
            const u64 wi = P2L.w_input, wo = P2L.w_output;
            if (role.site < 0) {
                // free-standing permutation row: inputs partly copied, a random swap bit where the gate has one, outputs offered for copying
                for (int k = 0; k < 12; k++) in[k] = input(r, wi + k);
                poseidon2_row(r, in, P2L.has_swap() ? rng.below(2) : 0, st);
                for (int i = 0; i < 12; i++) output(r, wo + i, st[i]);
            } else {
                const P2Site &site = p2_sites[role.site];
                if (role.block == 0) {
                    // first block: the message elements are this row's own input cells, then the terminator 1 and zeros; capacity zero
                    for (u64 j = 0; j < 12; j++) {
                        if (j < 8 && j < site.len) input(r, wi + j);
                        else tie(r, wi + j, P2_ZERO_ROW, (j < 8 && j == site.len) ? P2_ONE_COL : P2_ZERO_COL);
                    }
                } else {
                    // later blocks: rate part = the add row's sums, capacity = the previous permutation's capacity
                    for (u64 j = 0; j < 8; j++) tie(r, wi + j, r - 7, 4 * j + 3);
                    for (u64 j = 8; j < 12; j++) tie(r, wi + j, r - 8, wo + j);
                }
                for (int k = 0; k < 12; k++) in[k] = W(r, wi + k);
                if (P2L.has_swap()) tie(r, P2L.w_swap, P2_ZERO_ROW, P2_ZERO_COL);
                poseidon2_row(r, in, 0, st);
                const bool last = role.block + 1 == (int)site.blocks;
                for (int i = 0; i < 12; i++) {
                    if (last && i < 4) output(r, wo + i, st[i]);      // the digest: available to the rest of the circuit
                    else W(r, wo + i) = st[i];                        // sponge state: not offered to the copy pool
                }
            }
        } else if (kind == GATE_POSEIDON) {
            u64 in[12], st[12];
            if (pi_hash_row.count(r)) {
                // one absorption of the public-input hash, wired the way CircuitBuilder::build does it: the chunk's public
                // inputs are this row's own cells (free: the PartialWitness sets them), every other input is copy-connected
                // to the previous absorption's output (overwrite-mode sponge; zero for the first), swap = 0
                const u64 k = pi_hash_row[r], lo = 8 * k, len = std::min<u64>(8, num_public_inputs - lo);
                for (u64 j = 0; j < 12; j++) {
                    if (j < len) { W(r, j) = pis[lo + j]; pack.pi_cells[lo + j] = own_cell(r, j); }
                    else if (k == 0) { W(r, j) = 0; const uint32_t a = find(cell(1, 0)), b = find(cell(r, j)); if (a != b) parent[b] = a; }
                    else { W(r, j) = W(r - 1, 12 + j); const uint32_t a = find(cell(r - 1, 12 + j)), b = find(cell(r, j)); if (a != b) parent[b] = a; }
                    in[j] = W(r, j);
                }
                poseidon_row(r, in, 0, st);
                for (int i = 0; i < 12; i++) W(r, 12 + i) = st[i];          // sponge state: not offered to the copy pool
                if (lo + len == num_public_inputs)
                    for (int i = 0; i < 4; i++) { W(0, i) = st[i]; const uint32_t a = find(cell(r, 12 + i)), b = find(cell(0, i)); if (a != b) parent[b] = a; }
            } else {
                // PoseidonGate row: inputs (some copied), swap bit, deltas, recorded S-box inputs, outputs
                for (int k = 0; k < 12; k++) in[k] = input(r, k);
                poseidon_row(r, in, rng.below(2), st);
                for (int i = 0; i < 12; i++) output(r, 12 + i, st[i]);
            }
        }
    }
    // sigma: cycle through each copy class
    std::vector<uint32_t> order(parent.size());
    std::iota(order.begin(), order.end(), 0u);
    std::vector<uint32_t> root(parent.size());
    for (uint32_t i = 0; i < parent.size(); i++) root[i] = find(i);
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return root[a] < root[b]; });
    std::vector<uint32_t> next(parent.size());
    for (size_t s = 0; s < order.size();) {
        size_t e = s;
        while (e < order.size() && root[order[e]] == root[order[s]]) e++;
        for (size_t k = s; k < e; k++) next[order[k]] = order[k + 1 < e ? k + 1 : s];
        s = e;
    }
    std::vector<u64> omega_pow(n);
    { u64 w = gl::root_of_unity(degree_bits), a = 1; for (u64 i = 0; i < n; i++) { omega_pow[i] = gl::canon(a); a = gl::mul(a, w); } }
    const u64 sig0 = pack.num_selectors + pack.num_constants;
    for (u64 r = 0; r < n; r++)
        for (u64 c = 0; c < num_routed; c++) {
            uint32_t t = next[cell(r, c)];
            CS(r, sig0 + c) = gl::canon(gl::mul(pack.k_is[t % num_routed], omega_pow[t / num_routed]));
        }
    // circuit_digest: any 4 elements bound to the shape (the real one comes from the builder)
    u64 shape[6] = {degree_bits, num_wires, num_routed, num_public_inputs, seed % gl::P, 0x51504350ull + flags};
    host_hash_no_pad(shape, 6, pack.circuit_digest);
    return pack.validate();
}
