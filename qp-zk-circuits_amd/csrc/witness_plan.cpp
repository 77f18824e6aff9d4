// witness_plan.cpp — stage s1: the witness-generation plan of a loaded circuit and its C ABI.
//
// plonky2 runs its witness generators to a fixpoint over a dependency DAG on the host. Here the
// DAG is resolved once per circuit: copy classes are recovered from the sigma polynomials (a class is a cycle of the wire
// permutation), every gate row contributes its gate-attached generator instances with the cells they read and write,
// and instances are sorted by dependency depth. Generation is then one kernel launch per level plus a copy-fill pass
// (witness_kernels.hip). Cells no generator produces are the caller's inputs (the PartialWitness): qpgpu_witness_free_mask
// reports them.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstring>
#include <sys/random.h>
#include <numeric>
#include <string>
#include <map>
#include <unordered_map>
#include <vector>
#include "circuit_state.hpp"
#include "witness.hpp"
#include "prover_kernels.hpp"

using gl::u64;

struct WitnessPlan {
    std::vector<WitnessInst> insts;          // sorted by level
    std::vector<uint32_t> level_start;       // level l = insts[level_start[l] .. level_start[l+1])
    std::vector<uint32_t> level_poseidon;    // first PoseidonGate instance of level l (they come last in their level)
    std::vector<std::pair<uint32_t, uint32_t>> segments;   // launches: [l0, l1) — one wide level, or a run of narrow ones
    bool pi_gate_from_hash = true;           // every PublicInputGate row takes its wires from a hash output through the copy pass
    // A plan built WITH the caller's assignment list (only when the plan without it is cyclic: a generator that needs a target the
    // caller sets, whose only other producer depends on that generator — split_low_high / div_extension of an input): the classes
    // the caller assigns are known from the start, as a PartialWitness's values are in plonky2; their producers are checked.
    bool assigned_aware = false;
    std::vector<u64> assigned_cells;         // the list (incl. the public-input cells when they are supplied) the plan was built for
    std::vector<u64> decided_for;            // the assignment list this plan (of either kind) was last chosen for
    std::vector<uint8_t> assigned_checked;   // [num_wires][n]: 1 = a caller-assigned class that generators produce too (scatter AND check)
    uint32_t *d_level_start = nullptr, *d_level_poseidon = nullptr;
    std::vector<uint8_t> free_mask;          // [num_wires][n]: 1 = supplied by the caller
    uint64_t num_free = 0;
    uint32_t *d_src_of = nullptr;
    WitnessInst *d_insts = nullptr;
    u64 *d_pi_hash = nullptr;
    u64 *d_hints = nullptr;
    uint32_t pi_cap = 0;                     // witnesses d_pi_hash has room for
    // public-input cells (pack trailer "PUBI1"): stage s1 writes the caller's public inputs there
    uint32_t *d_pi_idx = nullptr;            // flat cell (column * n + row) of public input i
    u64 *d_pi_vals = nullptr;                // [pi_cap][num_public_inputs]
    // partial-witness entry: host view of the copy classes and growable device staging
    std::vector<uint32_t> h_src_of;          // [n][num_routed] -> flat source cell of the copy class
    std::vector<uint8_t> generated;          // [num_wires][n]: 1 = a generator produces this cell (or its copy class)
    uint32_t *d_pair_idx = nullptr; u64 *d_pair_val = nullptr; size_t pair_cap = 0;
    // copy classes with more than one producer (plonky2 lets several generators set one partition as long as they agree:
    // `connect(computed, claimed)` where both sides are generated). One producer is the class's source; the cells the others
    // write are compared with the source before the copy pass overwrites them (check_own[i] vs check_src[i], flat cells).
    std::vector<uint32_t> h_check_own, h_check_src;
    uint32_t *d_check_own = nullptr, *d_check_src = nullptr;
    // generators that invert one of their inputs (QuotientGeneratorExtension's denominator, InterpolationGenerator's coset shift):
    // plonky2's Field::inverse panics on zero ("Tried to invert zero"), so a witness where one of them is zero is an error, not a
    // witness. nz_a[i] / nz_b[i]: the flat source cells of the two limbs (the same cell twice for a base-field value); nz_cell[i]: the
    // cell (row * num_wires + wire) named in the message
    std::vector<uint32_t> h_nz_a, h_nz_b;
    std::vector<u64> h_nz_cell;
    uint32_t *d_nz_a = nullptr, *d_nz_b = nullptr;
    uint32_t *d_err = nullptr; uint32_t err_cap = 0;     // [err_cap][4]: first failing index per check (secondary producers; assignments; zero inverses)
    // a prepared PartialWitness shape (the cells a caller assigns are the same for every proof of a circuit): see PartialPrep
    struct PartialPrep *prep = nullptr;
};

// The resolution of one assignment list against the plan: assignment i (public inputs first, then the caller's cells) either
// seeds a free copy class (scattered into its source cell), or lands in a class a generator produces (compared with the
// generated value afterwards), or repeats an earlier assignment's class (the two values must agree on the host).
struct PartialPrep {
    std::vector<u64> cells;                       // the caller's list this was prepared for
    std::vector<uint32_t> scatter_from, check_from;   // index into the combined value vector [public inputs..., values...]
    std::vector<uint32_t> scatter_idx, check_idx; // flat cells
    std::vector<std::pair<uint32_t, uint32_t>> same;   // (earlier, later) value indices that share a class
    std::vector<u64> check_cell;                  // cell (row * num_wires + wire) named in messages
    uint32_t *d_scatter_idx = nullptr, *d_check_idx = nullptr, *d_keys = nullptr;   // d_keys: [cap][8], ChaCha20 keys of device-drawn values
    u64 *d_scatter_val = nullptr, *d_check_val = nullptr;   // [cap][count]
    uint32_t cap = 0;
    bool valid = false, with_pis = true;          // with_pis: the public-input cells are assignments too (values supplied by the caller)
    void release() {
        for (void *q : {(void *)d_scatter_idx, (void *)d_check_idx, (void *)d_scatter_val, (void *)d_check_val, (void *)d_keys}) if (q) (void)hipFree(q);
        d_scatter_idx = d_check_idx = d_keys = nullptr; d_scatter_val = d_check_val = nullptr; cap = 0;
    }
};

void witness_plan_free(WitnessPlan *p) {
    if (!p) return;
    if (p->d_src_of) (void)hipFree(p->d_src_of);
    if (p->d_insts) (void)hipFree(p->d_insts);
    if (p->d_level_start) (void)hipFree(p->d_level_start);
    if (p->d_level_poseidon) (void)hipFree(p->d_level_poseidon);
    if (p->d_pi_hash) (void)hipFree(p->d_pi_hash);
    if (p->d_hints) (void)hipFree(p->d_hints);
    if (p->d_pi_idx) (void)hipFree(p->d_pi_idx);
    if (p->d_pi_vals) (void)hipFree(p->d_pi_vals);
    if (p->d_pair_idx) (void)hipFree(p->d_pair_idx);
    if (p->d_pair_val) (void)hipFree(p->d_pair_val);
    if (p->d_check_own) (void)hipFree(p->d_check_own);
    if (p->d_check_src) (void)hipFree(p->d_check_src);
    if (p->d_nz_a) (void)hipFree(p->d_nz_a);
    if (p->d_nz_b) (void)hipFree(p->d_nz_b);
    if (p->d_err) (void)hipFree(p->d_err);
    if (p->prep) { p->prep->release(); delete p->prep; }
    delete p;
}

namespace {

struct IO { std::vector<uint32_t> in, out; };                                   // wire columns of the instance's own row
struct Cell { uint32_t row, col; };
struct CellIO { std::vector<Cell> in, out; };

// the cells (wire columns of its row) a generator instance reads and writes; must match witness_level_kernel
void describe(const GateInfo &g, uint32_t op, const P2GateLayout &lay, IO &io) {
    io.in.clear(); io.out.clear();
    auto range = [](std::vector<uint32_t> &v, uint32_t a, uint32_t b) { for (uint32_t i = a; i < b; i++) v.push_back(i); };
    switch (g.type) {
    case GATE_CONSTANT: io.out = {op}; break;
    case GATE_PUBLIC_INPUT: range(io.out, 0, 4); break;
    case GATE_ARITHMETIC: range(io.in, 4 * op, 4 * op + 3); io.out = {4 * op + 3}; break;
    case GATE_ARITHMETIC_EXT: range(io.in, 8 * op, 8 * op + 6); range(io.out, 8 * op + 6, 8 * op + 8); break;
    case GATE_MUL_EXT: range(io.in, 6 * op, 6 * op + 4); range(io.out, 6 * op + 4, 6 * op + 6); break;
    case GATE_BASE_SUM: io.in = {0}; range(io.out, 1, 1 + (uint32_t)g.param0); break;
    case GATE_POSEIDON: range(io.in, 0, 12); io.in.push_back(24); range(io.out, 12, 24); range(io.out, 25, 135); break;
    case GATE_POSEIDON2:
        range(io.in, lay.w_input, lay.w_input + 12); range(io.out, lay.w_output, lay.w_output + 12);
        if (lay.has_swap()) { io.in.push_back(lay.w_swap); range(io.out, lay.w_delta, lay.w_delta + 4); }
        range(io.out, lay.w_full0, lay.w_full0 + 12 * lay.full0_rounds()); range(io.out, lay.w_partial, lay.w_partial + 22);
        range(io.out, lay.w_full1, lay.w_full1 + 48);
        break;
    case GATE_REDUCING: case GATE_REDUCING_EXT: {
        const uint32_t nc = (uint32_t)g.param0, ncw = g.type == GATE_REDUCING_EXT ? 2 * nc : nc;
        range(io.in, 2, 6 + ncw); range(io.out, 0, 2); range(io.out, 6 + ncw, 6 + ncw + 2 * (nc - 1));
        break;
    }
    case GATE_RANDOM_ACCESS: {
        const uint32_t bits = (uint32_t)g.param0, copies = (uint32_t)g.param1, extra = (uint32_t)g.param2, vec = 1u << bits;
        const uint32_t routed = (2 + vec) * copies + extra;
        if (op == copies) { range(io.out, (2 + vec) * copies, routed); break; }
        const uint32_t b0 = (2 + vec) * op;
        io.in.push_back(b0); range(io.in, b0 + 2, b0 + 2 + vec);
        io.out.push_back(b0 + 1); range(io.out, routed + op * bits, routed + (op + 1) * bits);
        break;
    }
    case GATE_EXPONENTIATION: {
        const uint32_t nb = (uint32_t)g.param0;
        range(io.in, 0, 1 + nb); range(io.out, 1 + nb, 2 + 2 * nb);
        break;
    }
    case GATE_POSEIDON_MDS: range(io.in, 0, 24); range(io.out, 24, 48); break;
    case GATE_COSET_INTERPOLATION: {
        const uint32_t np = 1u << g.param0, deg = (uint32_t)g.param1, ni = (np - 2) / (deg - 1), s_ep = 1 + 2 * np;
        range(io.in, 0, s_ep + 2); range(io.out, s_ep + 2, s_ep + 4 + 4 * ni + 2);
        break;
    }
    default: break;
    }
}
uint32_t num_instances(const GateInfo &g) {
    switch (g.type) {
    case GATE_NOOP: return 0;
    case GATE_CONSTANT: case GATE_ARITHMETIC: case GATE_ARITHMETIC_EXT: case GATE_MUL_EXT: return (uint32_t)g.param0;
    case GATE_RANDOM_ACCESS: return (uint32_t)g.param1 + (g.param2 ? 1 : 0);
    default: return 1;
    }
}

// cells of any generator instance: a gate generator works on its row, a hint on the cells it names
void describe_any(const CircuitPack &p, const WitnessInst &w, IO &tmp, CellIO &io) {
    io.in.clear(); io.out.clear();
    if (w.gate != WITNESS_HINT) {
        describe(p.gates[w.gate], w.op, p.p2_layout, tmp);
        for (uint32_t c : tmp.in) io.in.push_back({w.row, c});
        for (uint32_t c : tmp.out) io.out.push_back({w.row, c});
        return;
    }
    const HintOp &h = p.hints[w.row];
    auto cell = [&](uint64_t x) { return Cell{(uint32_t)(x / p.num_wires), (uint32_t)(x % p.num_wires)}; };
    switch (h.w[0]) {
    case HINT_COPY: io.out = {cell(h.w[1])}; io.in = {cell(h.w[2])}; break;
    case HINT_EQUALITY: io.in = {cell(h.w[1]), cell(h.w[2])}; io.out = {cell(h.w[3]), cell(h.w[4])}; break;
    case HINT_WIRE_SPLIT: io.in = {cell(h.w[1])}; io.out = {cell(h.w[2])}; break;
    case HINT_QUOTIENT_EXT: io.in = {cell(h.w[1]), cell(h.w[2]), cell(h.w[3]), cell(h.w[4])}; io.out = {cell(h.w[5]), cell(h.w[6])}; break;
    case HINT_CONSTANT: io.out = {cell(h.w[1])}; break;
    case HINT_NONZERO_TEST: io.in = {cell(h.w[1])}; io.out = {cell(h.w[2])}; break;
    case HINT_LOW_HIGH: io.in = {cell(h.w[1])}; io.out = {cell(h.w[2]), cell(h.w[3])}; break;
    default: break;
    }
}

std::string build_plan(qpgpu_circuit *c, WitnessPlan &plan, const std::vector<u64> *assigned = nullptr) {
    const CircuitPack &p = c->pack;
    const u64 n = p.n(), R = p.num_routed_wires, NW = p.num_wires;
    const size_t sig0 = p.num_selectors + p.num_constants;
    if (NW * n >= (1ull << 32)) return "witness plan: circuit too large for 32-bit cell indices";
    auto CS = [&](u64 row, u64 col) { return p.constants_sigmas[col * n + row]; };

    // ---- copy classes from sigma: sigma(row, col) = k_is[col'] * w^row' names the next cell of the cycle ----
    std::unordered_map<u64, uint32_t> row_of;           // w^r -> r
    row_of.reserve(n * 2);
    { const u64 w = gl::root_of_unity((unsigned)p.degree_bits); u64 a = 1; for (u64 r = 0; r < n; r++) { row_of[gl::canon(a)] = (uint32_t)r; a = gl::mul(a, w); } }
    std::unordered_map<u64, uint32_t> col_of;           // k_is[c]^n -> c (cosets of the subgroup are told apart by x^n)
    std::vector<u64> k_inv(R);
    for (u64 cidx = 0; cidx < R; cidx++) {
        u64 t = p.k_is[cidx];
        for (unsigned i = 0; i < p.degree_bits; i++) t = gl::mul(t, t);
        if (!col_of.emplace(gl::canon(t), (uint32_t)cidx).second) return "witness plan: k_is are not in distinct cosets";
        k_inv[cidx] = gl::inv(p.k_is[cidx]);
    }
    std::vector<uint32_t> parent(n * R);
    std::iota(parent.begin(), parent.end(), 0u);
    auto find = [&](uint32_t x) { while (parent[x] != x) { parent[x] = parent[parent[x]]; x = parent[x]; } return x; };
    for (u64 r = 0; r < n; r++)
        for (u64 cidx = 0; cidx < R; cidx++) {
            const u64 s = CS(r, sig0 + cidx);
            u64 t = s;
            for (unsigned i = 0; i < p.degree_bits; i++) t = gl::mul(t, t);
            auto ci = col_of.find(gl::canon(t));
            if (ci == col_of.end()) return "witness plan: a sigma value is outside the wire cosets";
            auto ri = row_of.find(gl::canon(gl::mul(s, k_inv[ci->second])));
            if (ri == row_of.end()) return "witness plan: a sigma value is outside the wire cosets";
            const uint32_t a = find((uint32_t)(r * R + cidx)), b = find((uint32_t)((u64)ri->second * R + ci->second));
            if (a != b) parent[std::max(a, b)] = std::min(a, b);
        }

    // ---- generator instances per row ----
    std::vector<int32_t> gate_of_row(n, -1);
    for (u64 r = 0; r < n; r++)
        for (size_t gi = 0; gi < p.gates.size(); gi++)
            if (CS(r, p.gates[gi].selector_index) == gi) { gate_of_row[r] = (int32_t)gi; break; }
    std::vector<WitnessInst> insts;
    for (u64 r = 0; r < n; r++) {
        if (gate_of_row[r] < 0) return "witness plan: row " + std::to_string(r) + " selects no gate";
        const GateInfo &g = p.gates[gate_of_row[r]];
        for (uint32_t op = 0, k = num_instances(g); op < k; op++) insts.push_back({(uint32_t)r, (uint32_t)gate_of_row[r], op});
    }
    for (size_t hi = 0; hi < p.hints.size(); hi++) insts.push_back({(uint32_t)hi, WITNESS_HINT, (uint32_t)p.hints[hi].w[0]});
    // producers of each copy class (by class root). plonky2 lets any number of generators set one partition as long as they
    // agree; which of them becomes the class's source (the cell every member reads) is decided below, by who can run first.
    std::vector<std::vector<uint32_t>> producers(n * R);   // indexed by class root: instance ids
    plan.free_mask.assign(NW * n, 1);
    IO tmp;
    CellIO io;
    // PublicInputGate has no generator in plonky2: CircuitBuilder::build hashes the public-input targets with PoseidonGate
    // rows and copy-connects that hash to the gate's wires, so in an exported circuit those four cells are filled by the copy
    // pass. Only where nothing else produces them (the first synthetic circuits) does the row take the host-computed hash.
    auto is_pi = [&](const WitnessInst &w) { return w.gate != WITNESS_HINT && p.gates[w.gate].type == GATE_PUBLIC_INPUT; };
    std::vector<uint8_t> dropped(insts.size(), 0);
    for (int pass = 0; pass < 2; pass++)
        for (size_t id = 0; id < insts.size(); id++) {
            if ((is_pi(insts[id]) ? 1 : 0) != pass) continue;
            describe_any(p, insts[id], tmp, io);
            if (pass == 1) {
                size_t taken = 0;
                for (const Cell &oc : io.out) if (oc.col < R && !producers[find((uint32_t)((u64)oc.row * R + oc.col))].empty()) taken++;
                if (taken == io.out.size()) { dropped[id] = 1; continue; }
                plan.pi_gate_from_hash = false;
                if (taken) return "witness plan: PublicInputGate wires are only partly connected to a hash output (row " + std::to_string(insts[id].row) + ")";
            }
            for (const Cell &oc : io.out) {
                if (oc.col >= NW) return "witness plan: generator output beyond num_wires";
                plan.free_mask[(u64)oc.col * n + oc.row] = 0;
                if (oc.col >= R) continue;
                producers[find((uint32_t)((u64)oc.row * R + oc.col))].push_back((uint32_t)id);
            }
        }
    if (std::count(dropped.begin(), dropped.end(), 1)) {   // compact, keeping producer ids valid
        std::vector<int32_t> remap(insts.size(), -1);
        std::vector<WitnessInst> kept;
        for (size_t id = 0; id < insts.size(); id++) if (!dropped[id]) { remap[id] = (int32_t)kept.size(); kept.push_back(insts[id]); }
        for (auto &pl : producers) for (auto &pr : pl) pr = (uint32_t)remap[pr];
        insts.swap(kept);
    }

    // the caller's assignments (assignment-aware plans only): class root -> the cell the caller names first
    std::unordered_map<uint32_t, uint32_t> assigned_root;
    if (assigned)
        for (u64 cell : *assigned) {
            if (cell >= NW * n) continue;
            const u64 row = cell / NW, col = cell % NW;
            if (col >= R) continue;
            assigned_root.emplace(find((uint32_t)(row * R + col)), (uint32_t)(row * R + col));
        }
    // ---- levels, the way generate_partial_witness gets there: a generator runs once every target it watches is set, a
    // partition is set by the first of its producers to run. Level of an instance = 1 + the latest level at which one of its
    // input classes becomes known; a class becomes known at the level of its earliest producer (level 0 = caller-supplied).
    // Instances that never become runnable sit on a dependency cycle. ----
    std::vector<std::vector<uint32_t>> in_classes(insts.size());     // distinct producer-backed input classes (roots)
    std::vector<std::vector<uint32_t>> readers(n * R);               // class root -> instances waiting for it
    for (size_t id = 0; id < insts.size(); id++) {
        describe_any(p, insts[id], tmp, io);
        std::vector<uint32_t> &ic = in_classes[id];
        for (const Cell &c_ : io.in) {
            if (c_.col >= R) continue;
            const uint32_t root = find((uint32_t)((u64)c_.row * R + c_.col));
            const auto &pl = producers[root];
            if (pl.empty() || (pl.size() == 1 && pl[0] == id) || assigned_root.count(root)) continue;      // free, produced by this very instance, or set by the caller
            ic.push_back(root);
        }
        std::sort(ic.begin(), ic.end());
        ic.erase(std::unique(ic.begin(), ic.end()), ic.end());
        for (uint32_t root : ic) readers[root].push_back((uint32_t)id);
    }
    std::vector<int32_t> level(insts.size(), 0), class_level(n * R, -1);
    std::vector<int32_t> primary(n * R, -1);            // class root -> the producer that is its source
    std::vector<uint32_t> source(n * R);                // class root -> source cell (row * R + col)
    for (uint32_t i = 0; i < n * R; i++) source[i] = i;
    for (const auto &ar : assigned_root) { class_level[ar.first] = 0; source[ar.first] = ar.second; }
    {
        std::vector<uint32_t> waiting(insts.size());
        std::vector<std::vector<uint32_t>> bucket(2);   // bucket[l] = instances that run at level l (levels are reached in order)
        for (size_t id = 0; id < insts.size(); id++) { waiting[id] = (uint32_t)in_classes[id].size(); if (!waiting[id]) { level[id] = 1; bucket[1].push_back((uint32_t)id); } }
        for (size_t l = 1; l < bucket.size(); l++) {
            for (size_t k = 0; k < bucket[l].size(); k++) {
                const uint32_t id = bucket[l][k];
                describe_any(p, insts[id], tmp, io);
                for (const Cell &oc : io.out) {
                    if (oc.col >= R) continue;
                    const uint32_t cell = (uint32_t)((u64)oc.row * R + oc.col), root = find(cell);
                    if (class_level[root] >= 0) continue;
                    class_level[root] = (int32_t)l; primary[root] = (int32_t)id; source[root] = cell;
                    for (uint32_t rd : readers[root]) {
                        if (level[rd] > 0) continue;
                        if (--waiting[rd] == 0) {
                            level[rd] = (int32_t)l + 1;
                            if (bucket.size() <= l + 1) bucket.resize(l + 2);
                            bucket[l + 1].push_back(rd);
                        }
                    }
                }
            }
        }
        for (size_t id = 0; id < insts.size(); id++) if (level[id] == 0) return "witness plan: cyclic generator dependency";
    }
    // cells written by a producer that is not its class's source: compared with the source before the copy pass
    for (size_t id = 0; id < insts.size(); id++) {
        describe_any(p, insts[id], tmp, io);
        for (const Cell &oc : io.out) {
            if (oc.col >= R) continue;
            const uint32_t cell = (uint32_t)((u64)oc.row * R + oc.col), s = source[find(cell)];
            if (s == cell) continue;
            plan.h_check_own.push_back((uint32_t)((u64)oc.col * n + oc.row));
            plan.h_check_src.push_back((uint32_t)((u64)(s % R) * n + s / R));
        }
    }
    std::vector<uint32_t> src_of(n * R);
    for (uint32_t cell = 0; cell < n * R; cell++) {
        const uint32_t s = source[find(cell)];
        src_of[cell] = (uint32_t)((u64)(s % R) * n + s / R);
        if (s != cell) plan.free_mask[(u64)(cell % R) * n + cell / R] = 0;      // filled by the copy pass
    }
    plan.num_free = 0;
    for (uint8_t m : plan.free_mask) plan.num_free += m;
    plan.generated.assign(NW * n, 0);
    plan.assigned_checked.assign(assigned ? NW * n : 0, 0);
    for (u64 col = 0; col < NW; col++)
        for (u64 r = 0; r < n; r++) {
            if (col >= R) { plan.generated[col * n + r] = plan.free_mask[col * n + r] == 0; continue; }
            const uint32_t root = find((uint32_t)(r * R + col));
            const bool produced = !producers[root].empty(), by_caller = assigned_root.count(root) != 0;
            plan.generated[col * n + r] = produced && !by_caller;
            if (produced && by_caller) plan.assigned_checked[col * n + r] = 1;
        }
    plan.assigned_aware = assigned != nullptr;
    if (assigned) plan.assigned_cells = *assigned;
    plan.h_src_of = src_of;
    for (const WitnessInst &w : insts) {
        if (w.gate == WITNESS_HINT) {
            const HintOp &h = p.hints[w.row];
            if (h.w[0] != 4) continue;
            plan.h_nz_a.push_back(src_of[(h.w[3] / NW) * R + h.w[3] % NW]);
            plan.h_nz_b.push_back(src_of[(h.w[4] / NW) * R + h.w[4] % NW]);
            plan.h_nz_cell.push_back(h.w[3]);
        } else if (p.gates[w.gate].type == GATE_COSET_INTERPOLATION) {
            plan.h_nz_a.push_back(src_of[(u64)w.row * R]);
            plan.h_nz_b.push_back(src_of[(u64)w.row * R]);
            plan.h_nz_cell.push_back((u64)w.row * NW);
        }
    }

    int32_t max_level = 0;
    for (int32_t l : level) max_level = std::max(max_level, l);
    std::vector<uint32_t> order(insts.size());
    std::iota(order.begin(), order.end(), 0u);
    auto hash_gate = [&](const WitnessInst &w) { return w.gate != WITNESS_HINT && (p.gates[w.gate].type == GATE_POSEIDON || p.gates[w.gate].type == GATE_POSEIDON2); };
    auto is_pos = [&](uint32_t i) { return hash_gate(insts[i]) ? 1 : 0; };   // rows of the two hash gates: lane-cooperative generators, last in their level
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return level[a] != level[b] ? level[a] < level[b] : is_pos(a) < is_pos(b); });
    plan.insts.resize(insts.size());
    plan.level_start.assign(max_level + 1, 0);
    for (size_t i = 0; i < order.size(); i++) { plan.insts[i] = insts[order[i]]; plan.level_start[level[order[i]]]++; }
    {   // counts per level (index 1..max) -> start offsets; level_start[l] = first instance of level l+1
        uint32_t acc = 0;
        for (int32_t l = 1; l <= max_level; l++) { const uint32_t cnt = plan.level_start[l]; plan.level_start[l - 1] = acc; acc += cnt; }
        plan.level_start[max_level] = acc;
    }
    plan.level_poseidon.assign(max_level, 0);
    for (int32_t l = 0; l < max_level; l++) {
        uint32_t k = plan.level_start[l];
        while (k < plan.level_start[l + 1] && !hash_gate(plan.insts[k])) k++;
        plan.level_poseidon[l] = k;
    }

    // ---- device copies ----
    qpgpu_ctx *ctx = c->ctx;
    auto up = [&](const void *src, size_t bytes, void **dst) -> bool {
        if (hipMalloc(dst, std::max<size_t>(bytes, 8)) != hipSuccess) return false;
        return hipMemcpyAsync(*dst, src, bytes, hipMemcpyHostToDevice, ctx->stream) == hipSuccess && hipStreamSynchronize(ctx->stream) == hipSuccess;
    };
    if (!up(src_of.data(), src_of.size() * 4, (void **)&plan.d_src_of)) return "witness plan: device allocation failed";
    if (!up(plan.insts.data(), plan.insts.size() * sizeof(WitnessInst), (void **)&plan.d_insts)) return "witness plan: device allocation failed";
    if (!up(plan.level_start.data(), plan.level_start.size() * 4, (void **)&plan.d_level_start)) return "witness plan: device allocation failed";
    if (!up(plan.level_poseidon.data(), plan.level_poseidon.size() * 4, (void **)&plan.d_level_poseidon)) return "witness plan: device allocation failed";
    // launches: consecutive levels narrow enough for one workgroup go out as one run. Measured both ways (tools/witness_time.py):
    // a run keeps a witness on one CU, which saves the launch gap and the cross-XCD L2 misses of a launch per level (2^16 rows,
    // 2 562 levels: 108 -> 95 ms) but funnels every level's scattered loads through one CU's address unit (2^13 rows, 294 wider
    // levels: 5.7 -> 6.7 ms). Default: runs for deep plans only. QPGPU_WITNESS_FUSE=0 never, =1 always.
    {
        const char *e = getenv("QPGPU_WITNESS_FUSE");
        const bool fuse = e && (*e == '0' || *e == '1') ? *e == '1' : max_level >= 1024;
        const uint32_t L = (uint32_t)max_level;
        auto narrow = [&](uint32_t l) { return fuse && plan.level_poseidon[l] - plan.level_start[l] <= WITNESS_RUN_GENERIC_CAP &&
                                               plan.level_start[l + 1] - plan.level_poseidon[l] <= WITNESS_RUN_POSEIDON_CAP; };
        for (uint32_t l = 0; l < L;) {
            uint32_t e1 = l + 1;
            if (narrow(l)) while (e1 < L && narrow(e1)) e1++;
            plan.segments.push_back({l, e1});
            l = e1;
        }
    }
    if (getenv("QPGPU_WITNESS_DUMP")) {
        for (uint32_t l = 0; l < (uint32_t)max_level; l++) {
            std::map<int, int> hist;
            for (uint32_t k = plan.level_start[l]; k < plan.level_start[l + 1]; k++) {
                const WitnessInst &in = plan.insts[k];
                hist[in.gate == WITNESS_HINT ? 1000 + (int)in.op : (int)p.gates[in.gate].type]++;
            }
            fprintf(stderr, "LEVEL %u n=%u pos=%u :", l, plan.level_start[l + 1] - plan.level_start[l], plan.level_start[l + 1] - plan.level_poseidon[l]);
            for (auto &kv : hist) fprintf(stderr, " t%d=%d", kv.first, kv.second);
            fprintf(stderr, "\n");
        }
        fprintf(stderr, "SEGMENTS %zu\n", plan.segments.size());
    }
    if (!p.hints.empty() && !up(p.hints.data(), p.hints.size() * sizeof(HintOp), (void **)&plan.d_hints)) return "witness plan: device allocation failed";
    if (!p.pi_cells.empty()) {
        std::vector<uint32_t> flat(p.pi_cells.size());
        // a public input is written where its copy class is read from (the class's source cell)
        for (size_t i = 0; i < flat.size(); i++) flat[i] = src_of[(p.pi_cells[i] / NW) * R + p.pi_cells[i] % NW];
        if (!up(flat.data(), flat.size() * 4, (void **)&plan.d_pi_idx)) return "witness plan: device allocation failed";
    }
    if (!plan.h_check_own.empty()) {
        if (!up(plan.h_check_own.data(), plan.h_check_own.size() * 4, (void **)&plan.d_check_own)) return "witness plan: device allocation failed";
        if (!up(plan.h_check_src.data(), plan.h_check_src.size() * 4, (void **)&plan.d_check_src)) return "witness plan: device allocation failed";
    }
    if (!plan.h_nz_a.empty()) {
        if (!up(plan.h_nz_a.data(), plan.h_nz_a.size() * 4, (void **)&plan.d_nz_a)) return "witness plan: device allocation failed";
        if (!up(plan.h_nz_b.data(), plan.h_nz_b.size() * 4, (void **)&plan.d_nz_b)) return "witness plan: device allocation failed";
    }
    return "";
}

// The plan of a circuit. Without an assignment list: the plan every entry point shares. With one (the PartialWitness entries): the
// plan built WITHOUT the list, unless (a) the circuit's generators form a cycle that only the caller's values break, or (b) the list
// sets targets that generators produce too (beyond the public inputs: hash hints of a front-end, include/qpgpu_leaf.h) and the plan
// built FOR the list — those classes known from the start, their producers checked against the caller's values, as a PartialWitness's
// values are there first in plonky2 — has fewer dependency levels. The choice is remembered for the list (rebuilt when it changes).
int ensure_plan(qpgpu_circuit *c, const std::vector<u64> *assigned = nullptr, size_t pi_prefix = 0) {
    // (an entry point without a list — a witness from the free cells — cannot run on a plan built for somebody's assignments)
    if (c->wplan && (assigned ? c->wplan->decided_for == *assigned : !c->wplan->assigned_aware)) return QPGPU_OK;
    if (c->wplan && assigned && !c->wplan->assigned_aware) {
        // a plan built without a list is in place: it stays unless this list carries hints
        const WitnessPlan &d = *c->wplan;
        bool hints = false;
        const u64 NW = c->pack.num_wires, n = c->pack.n();
        for (size_t i = pi_prefix; i < assigned->size() && !hints; i++) { const u64 cell = (*assigned)[i]; hints = cell < NW * n && d.generated[(cell % NW) * n + cell / NW]; }
        if (!hints) { c->wplan->decided_for = *assigned; return QPGPU_OK; }
    }
    if (c->wplan) { QP_HIP(c->ctx, hipStreamSynchronize(c->ctx->stream)); witness_plan_free(c->wplan); c->wplan = nullptr; }
    WitnessPlan *plan = new WitnessPlan();
    std::string err = build_plan(c, *plan);
    if (assigned) {
        bool want_aware = !err.empty() && err.find("cyclic generator dependency") != std::string::npos;
        if (err.empty()) {
            const u64 NW = c->pack.num_wires, n = c->pack.n();
            for (size_t i = pi_prefix; i < assigned->size() && !want_aware; i++) { const u64 cell = (*assigned)[i]; want_aware = cell < NW * n && plan->generated[(cell % NW) * n + cell / NW]; }
        }
        if (want_aware) {
            WitnessPlan *aware = new WitnessPlan();
            const std::string err2 = build_plan(c, *aware, assigned);
            if (err2.empty() && (!err.empty() || aware->level_start.size() < plan->level_start.size())) { witness_plan_free(plan); plan = aware; err.clear(); }
            else { witness_plan_free(aware); if (!err.empty()) err = err2.empty() ? err : err2; }
        }
    }
    if (!err.empty()) { witness_plan_free(plan); return c->ctx->fail(QPGPU_EINVAL, err); }
    if (assigned) plan->decided_for = *assigned;
    c->wplan = plan;
    return QPGPU_OK;
}

void host_pi_hash(const hasher::Config &h, const u64 *pis, size_t n, u64 out[4]) {
    u64 st[12] = {0};
    for (size_t i = 0; i < n; i += 8) {
        const size_t len = std::min<size_t>(8, n - i);
        for (size_t k = 0; k < len; k++) st[k] = gl::canon(pis[i + k]);
        h.permute(st);
    }
    std::memcpy(out, st, 32);
}

}  // namespace

extern "C" {

int qpgpu_witness_info(qpgpu_circuit *c, uint64_t *num_generators, uint64_t *num_levels, uint64_t *num_free_cells) {
    if (!c) return QPGPU_EINVAL;
    QP_DEV(c->ctx);
    if (!c->wplan) QP_TRY(ensure_plan(c));        // (reports the plan in place — of either kind — and builds the shared one only if there is none)
    if (num_generators) *num_generators = c->wplan->insts.size();
    if (num_levels) *num_levels = c->wplan->level_start.empty() ? 0 : c->wplan->level_start.size() - 1;
    if (num_free_cells) *num_free_cells = c->wplan->num_free;
    return QPGPU_OK;
}

int qpgpu_witness_free_mask(qpgpu_circuit *c, uint8_t *mask, size_t mask_len) {
    if (!c || !mask) return QPGPU_EINVAL;
    QP_DEV(c->ctx);
    if (!c->wplan) QP_TRY(ensure_plan(c));        // (of the plan in place: under a plan built for an assignment list the assigned classes count as supplied)
    if (mask_len < c->wplan->free_mask.size()) return c->ctx->fail(QPGPU_EBUFSIZE, "witness_free_mask: buffer too small");
    std::memcpy(mask, c->wplan->free_mask.data(), c->wplan->free_mask.size());
    return QPGPU_OK;
}

}  // extern "C"

namespace {
// Resolve an assignment list against the plan (see PartialPrep). Public inputs come first in the combined value vector.
std::string prepare_partial(const CircuitPack &p, const WitnessPlan &plan, const uint64_t *cells, size_t count, PartialPrep &pp, bool with_pis) {
    const u64 n = p.n(), NW = p.num_wires, R = p.num_routed_wires;
    const uint32_t npis = (uint32_t)p.num_public_inputs;
    pp.cells.assign(cells, cells + count);
    pp.scatter_from.clear(); pp.check_from.clear(); pp.scatter_idx.clear(); pp.check_idx.clear(); pp.same.clear(); pp.check_cell.clear();
    std::unordered_map<uint32_t, uint32_t> first;       // class key -> value index of the first assignment
    first.reserve((count + npis) * 2);
    auto assign = [&](u64 cell, uint32_t from) -> std::string {
        if (cell >= NW * n) return "generate_witness_partial: cell " + std::to_string(cell) + " is outside the trace";
        const u64 row = cell / NW, col = cell % NW;
        const uint32_t own = (uint32_t)(col * n + row), key = col < R ? plan.h_src_of[row * R + col] : own;
        auto it = first.find(key);
        if (it != first.end()) { pp.same.push_back({it->second, from}); return ""; }
        first.emplace(key, from);
        if (plan.generated[own]) { pp.check_idx.push_back(key); pp.check_from.push_back(from); pp.check_cell.push_back(cell); }
        else { pp.scatter_idx.push_back(key); pp.scatter_from.push_back(from); }   // a free class is read through its source cell
        if (!plan.assigned_checked.empty() && plan.assigned_checked[own]) {       // set by the caller AND produced: what the generators leave there must be the caller's value
            pp.check_idx.push_back(key); pp.check_from.push_back(from); pp.check_cell.push_back(cell);
        }
        return "";
    };
    pp.with_pis = with_pis;
    for (size_t i = 0; with_pis && i < p.pi_cells.size(); i++) { const std::string e = assign(p.pi_cells[i], (uint32_t)i); if (!e.empty()) return e; }
    for (size_t i = 0; i < count; i++) { const std::string e = assign(cells[i], npis + (uint32_t)i); if (!e.empty()) return e; }
    return "";
}
int prep_device(qpgpu_ctx *ctx, PartialPrep &pp, uint32_t batch) {
    if (pp.cap >= batch && pp.d_scatter_idx) return QPGPU_OK;
    QP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    pp.release();
    const size_t ns = std::max<size_t>(pp.scatter_idx.size(), 1), nc = std::max<size_t>(pp.check_idx.size(), 1);
    QP_HIP(ctx, hipMalloc((void **)&pp.d_scatter_idx, ns * 4));
    QP_HIP(ctx, hipMalloc((void **)&pp.d_check_idx, nc * 4));
    QP_HIP(ctx, hipMalloc((void **)&pp.d_scatter_val, ns * 8 * batch));
    QP_HIP(ctx, hipMalloc((void **)&pp.d_check_val, nc * 8 * batch));
    QP_HIP(ctx, hipMalloc((void **)&pp.d_keys, (size_t)32 * batch));
    if (!pp.scatter_idx.empty()) QP_HIP(ctx, hipMemcpyAsync(pp.d_scatter_idx, pp.scatter_idx.data(), pp.scatter_idx.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    if (!pp.check_idx.empty()) QP_HIP(ctx, hipMemcpyAsync(pp.d_check_idx, pp.check_idx.data(), pp.check_idx.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    QP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    pp.cap = batch;
    return QPGPU_OK;
}

// Stage s1 for `batch` witnesses. With `pp`: the wire matrices are cleared and seeded with the prepared assignments first
// (values: [batch][npis + count] is assembled from public_inputs and part_values). status (may be null): per witness
// QPGPU_OK / QPGPU_EUNSAT. Returns QPGPU_EUNSAT when any witness failed (the first one's reason in last_error).
// n_blind: the LAST n_blind cells of the prepared list take values drawn on the device (ChaCha20 under seeds[b], or under 32 bytes
// of OS entropy per witness when seeds is null); part_values then holds count - n_blind values per witness.
int generate_batch(qpgpu_circuit *c, uint64_t *d_wires, uint32_t batch, const uint64_t *public_inputs, PartialPrep *pp,
                          const uint64_t *part_values, int *status, size_t n_blind = 0, const uint8_t *seeds = nullptr) {
    qpgpu_ctx *ctx = c->ctx;
    WitnessPlan &plan = *c->wplan;
    const CircuitPack &p = c->pack;
    const size_t npis = p.num_public_inputs, all_cells = pp ? pp->cells.size() : 0, count = all_cells - n_blind;
    // the device-drawn cells must be the tail of the scatter list: free classes nothing else assigns
    size_t ns_host = pp ? pp->scatter_idx.size() : 0;
    if (n_blind) {
        const uint32_t first_blind = (uint32_t)(npis + count);
        ns_host = (size_t)(std::lower_bound(pp->scatter_from.begin(), pp->scatter_from.end(), first_blind) - pp->scatter_from.begin());
        bool ok = pp->scatter_idx.size() - ns_host == n_blind;
        for (uint32_t f : pp->check_from) ok = ok && f < first_blind;
        for (const auto &sm : pp->same) ok = ok && sm.first < first_blind && sm.second < first_blind;
        if (!ok) return ctx->fail(QPGPU_EINVAL, "generate_witness_partial: a blinding cell is not a free cell of its own (it is generated, copy-connected to another assigned cell, or listed twice)");
    }
    const u64 NW = p.num_wires, stride = p.num_wires * p.n();
    auto name = [&](u64 cell) { return "target (row " + std::to_string(cell / NW) + ", wire " + std::to_string(cell % NW) + ")"; };
    if (status) for (uint32_t b = 0; b < batch; b++) status[b] = QPGPU_OK;
    int overall = QPGPU_OK;
    auto flag = [&](uint32_t b, const std::string &why) {
        if (status) status[b] = QPGPU_EUNSAT;
        if (overall == QPGPU_OK) { overall = QPGPU_EUNSAT; ctx->err = (batch > 1 ? "witness " + std::to_string(b) + ": " : std::string()) + why; }
    };
    // two assignments of one copy class must agree (host check; the later one is not written)
    if (pp)
        for (uint32_t b = 0; b < batch; b++) {
            auto val = [&](uint32_t from) { return gl::canon(from < npis ? public_inputs[(size_t)b * npis + from] : part_values[(size_t)b * count + (from - npis)]); };
            auto cell_of = [&](uint32_t from) { return from < npis ? p.pi_cells[from] : pp->cells[from - npis]; };
            for (const auto &sm : pp->same)
                if (val(sm.first) != val(sm.second)) {
                    flag(b, name(cell_of(sm.second)) + " set twice with different values (" + std::to_string(val(sm.second)) + " here, " + std::to_string(val(sm.first)) +
                                " through " + name(cell_of(sm.first)) + ")");
                    break;
                }
        }
    const bool pair_checks = !plan.h_check_own.empty(), val_checks = pp && !pp->check_idx.empty(), nz_checks = !plan.h_nz_a.empty();
    if (plan.pi_cap < batch || (plan.err_cap < batch)) {
        QP_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (plan.d_pi_hash) { (void)hipFree(plan.d_pi_hash); plan.d_pi_hash = nullptr; }
        if (plan.d_pi_vals) { (void)hipFree(plan.d_pi_vals); plan.d_pi_vals = nullptr; }
        if (plan.d_err) { (void)hipFree(plan.d_err); plan.d_err = nullptr; }
        plan.pi_cap = 0; plan.err_cap = 0;
        QP_HIP(ctx, hipMalloc((void **)&plan.d_pi_hash, (size_t)batch * 32));
        QP_HIP(ctx, hipMalloc((void **)&plan.d_pi_vals, std::max<size_t>((size_t)batch * npis * 8, 8)));
        QP_HIP(ctx, hipMalloc((void **)&plan.d_err, (size_t)batch * 16));
        plan.pi_cap = batch; plan.err_cap = batch;
    }
    if (pp) QP_TRY(prep_device(ctx, *pp, batch));
    // everything the host sends goes up through the context's pinned bounce buffer and a copy kernel (ctx.hpp: read_back)
    const size_t ns = pp ? pp->scatter_idx.size() : 0, nc = pp ? pp->check_idx.size() : 0;
    const size_t pi_bytes = plan.d_pi_idx && public_inputs ? (size_t)batch * npis * 8 : 0, hash_bytes = (size_t)batch * 32;
    const size_t sc_bytes = (size_t)batch * ns_host * 8, ck_bytes = (size_t)batch * nc * 8, key_bytes = n_blind ? (size_t)batch * 32 : 0;
    QP_TRY(ctx->reserve_read_back(pi_bytes + hash_bytes + sc_bytes + ck_bytes + key_bytes + (size_t)batch * 8));
    QP_HIP(ctx, hipStreamSynchronize(ctx->stream));       // nothing of an earlier call still reads the bounce buffer
    u64 *bounce = (u64 *)ctx->h_pin;
    if (pp) QP_HIP(ctx, hipMemsetAsync(d_wires, 0, (size_t)batch * stride * 8, ctx->stream));
    if (pi_bytes) std::memcpy(bounce, public_inputs, pi_bytes);
    u64 *pih = bounce + pi_bytes / 8;
    for (uint32_t b = 0; b < batch; b++) {
        if (public_inputs) host_pi_hash(ctx->hasher, public_inputs + (size_t)b * npis, npis, pih + 4 * b);
        else std::memset(pih + 4 * b, 0, 32);        // derived public inputs: no PublicInputGate generator reads it (pi_gate_from_hash)
    }
    u64 *scv = pih + hash_bytes / 8, *ckv = scv + sc_bytes / 8;
    if (pp)
        for (uint32_t b = 0; b < batch; b++) {
            auto val = [&](uint32_t from) { return gl::canon(from < npis ? public_inputs[(size_t)b * npis + from] : part_values[(size_t)b * count + (from - npis)]); };
            for (size_t i = 0; i < ns_host; i++) scv[(size_t)b * ns_host + i] = val(pp->scatter_from[i]);
            for (size_t i = 0; i < nc; i++) ckv[(size_t)b * nc + i] = val(pp->check_from[i]);
        }
    u64 *keyv = ckv + ck_bytes / 8;
    if (n_blind) {
        if (seeds) std::memcpy(keyv, seeds, key_bytes);
        else {
            size_t got = 0;
            while (got < key_bytes) {
                const ssize_t r = getrandom((uint8_t *)keyv + got, key_bytes - got, 0);
                if (r <= 0) return ctx->fail(QPGPU_EDEVICE, "generate_witness_partial: the OS entropy source failed");
                got += (size_t)r;
            }
        }
    }
    if (pi_bytes) {   // PartialWitness::set_target for every public-input target
        QP_HIP(ctx, pk_copy(plan.d_pi_vals, bounce, pi_bytes, ctx->stream));
        QP_HIP(ctx, wk_scatter(d_wires, plan.d_pi_idx, plan.d_pi_vals, (uint32_t)npis, batch, stride, (uint32_t)npis, ctx->stream));
    }
    QP_HIP(ctx, pk_copy(plan.d_pi_hash, pih, hash_bytes, ctx->stream));
    if (ns) {
        if (ns_host == ns) QP_HIP(ctx, pk_copy(pp->d_scatter_val, scv, sc_bytes, ctx->stream));
        else {
            if (ns_host) QP_HIP(ctx, pk_unpack_rows(scv, ns, ns_host, batch, pp->d_scatter_val, ctx->stream));
            QP_HIP(ctx, pk_copy(pp->d_keys, keyv, key_bytes, ctx->stream));
            QP_HIP(ctx, pk_random_felts(pp->d_keys, n_blind, pp->d_scatter_val + ns_host, ns, batch, ctx->stream));
        }
        QP_HIP(ctx, wk_scatter(d_wires, pp->d_scatter_idx, pp->d_scatter_val, (uint32_t)ns, batch, stride, (uint32_t)ns, ctx->stream));
    }
    if (nc) QP_HIP(ctx, pk_copy(pp->d_check_val, ckv, ck_bytes, ctx->stream));
    if (pair_checks || val_checks || nz_checks) QP_HIP(ctx, hipMemsetAsync(plan.d_err, 0xFF, (size_t)batch * 16, ctx->stream));
    WitnessArgs a{};
    a.wires = d_wires; a.src_of = plan.d_src_of; a.insts = plan.d_insts; a.gates = c->d_gates; a.cs = c->d_cs_values;
    a.poseidon_rc = c->d_poseidon_rc; a.poseidon_fast = c->d_poseidon_fast; a.pi_hash = plan.d_pi_hash;
    a.p2_gate = c->has_p2_gate ? ctx->d_p2_app : nullptr; a.p2_layout = c->pack.p2_layout;
    a.hints = plan.d_hints; a.num_wires = (uint32_t)c->pack.num_wires;
    a.n = c->pack.n(); a.batch_stride = stride;
    a.num_routed = (uint32_t)c->pack.num_routed_wires; a.num_selectors = (uint32_t)c->pack.num_selectors;
    // one launch per dependency level (ordinary instances and PoseidonGate rows side by side); QPGPU_WITNESS_COMBINED=0: two
    static const bool combined = [] { const char *e = getenv("QPGPU_WITNESS_COMBINED"); return !(e && *e == '0'); }();
    ctx->prof_begin("witness_generate");
    for (const auto &seg : plan.segments) {
        if (seg.second - seg.first > 1) { QP_HIP(ctx, wk_run_levels(a, plan.d_level_start, plan.d_level_poseidon, seg.first, seg.second, batch, ctx->stream)); continue; }
        const size_t l = seg.first;
        const uint32_t lo = plan.level_start[l], mid = plan.level_poseidon[l], hi = plan.level_start[l + 1];
        if (combined) QP_HIP(ctx, wk_run_combined(a, lo, mid - lo, hi - mid, batch, ctx->stream));
        else {
            QP_HIP(ctx, wk_run_level(a, lo, mid - lo, batch, ctx->stream));
            QP_HIP(ctx, wk_run_poseidon(a, mid, hi - mid, batch, ctx->stream));
        }
    }
    // a partition set twice: the cells of producers that are not their class's source, and the caller's assignments inside
    // generated classes, against the value the class's source holds (before the copy pass makes them equal)
    if (pair_checks) QP_HIP(ctx, wk_check_pairs(d_wires, plan.d_check_own, plan.d_check_src, (uint32_t)plan.h_check_own.size(), batch, stride, plan.d_err, 0, ctx->stream));
    if (val_checks) QP_HIP(ctx, wk_check_vals(d_wires, pp->d_check_idx, pp->d_check_val, (uint32_t)nc, batch, stride, (uint32_t)nc, plan.d_err, 1, ctx->stream));
    if (nz_checks) QP_HIP(ctx, wk_check_nonzero(d_wires, plan.d_nz_a, plan.d_nz_b, (uint32_t)plan.h_nz_a.size(), batch, stride, plan.d_err, 2, ctx->stream));
    QP_HIP(ctx, wk_fill_copies(a, batch, ctx->stream));
    ctx->prof_end();
    if (pair_checks || val_checks || nz_checks) {
        std::vector<uint32_t> err((size_t)batch * 4);
        QP_TRY(ctx->read_back(err.data(), plan.d_err, err.size() * 4));
        const u64 n = p.n();
        for (uint32_t b = 0; b < batch; b++) {
            if (nz_checks && err[4 * b + 2] != 0xFFFFFFFFu) {      // first: the value such a generator wrote is what the other two checks would trip over
                flag(b, "a generator inverts " + name(plan.h_nz_cell[err[4 * b + 2]]) + ", which is zero (a quotient's denominator or an interpolation's coset shift)");
            } else if (pair_checks && err[4 * b] != 0xFFFFFFFFu) {
                const uint32_t own = plan.h_check_own[err[4 * b]];
                flag(b, name((u64)(own % n) * NW + own / n) + " set twice with different values (two generators of one copy class disagree)");
            } else if (val_checks && err[4 * b + 1] != 0xFFFFFFFFu) {
                const uint32_t k = err[4 * b + 1], from = pp->check_from[k];
                const u64 v = gl::canon(from < npis ? public_inputs[(size_t)b * npis + from] : part_values[(size_t)b * count + (from - npis)]);
                flag(b, name(pp->check_cell[k]) + " set twice with different values (" + std::to_string(v) + " supplied, another value generated)");
            }
        }
    }
    if (n_blind) {   // the keys are secrets: the device has read the bounce buffer (the launches above are ordered before this sync)
        QP_HIP(ctx, hipStreamSynchronize(ctx->stream));
        volatile u64 *kz = keyv;
        for (size_t i = 0; i < key_bytes / 8; i++) kz[i] = 0;
    }
    return overall;
}

// the plan's prepared assignment list for `cells` (rebuilt when the caller's list changes)
int ensure_prep(qpgpu_circuit *c, const uint64_t *cells, size_t count, uint32_t batch, bool with_pis = true) {
    qpgpu_ctx *ctx = c->ctx;
    WitnessPlan &plan = *c->wplan;
    if (!plan.prep) plan.prep = new PartialPrep();
    PartialPrep &pp = *plan.prep;
    if (!pp.valid || pp.with_pis != with_pis || pp.cells.size() != count || (count && std::memcmp(pp.cells.data(), cells, count * 8) != 0)) {
        QP_HIP(ctx, hipStreamSynchronize(ctx->stream));
        pp.release();
        const std::string err = prepare_partial(c->pack, plan, cells, count, pp, with_pis);
        if (!err.empty()) { pp.valid = false; return ctx->fail(QPGPU_EINVAL, err); }
        pp.valid = true;
    }
    return prep_device(ctx, pp, batch);
}

// the PartialWitness entries: the shared plan, or — for a circuit whose generators only the caller's values untangle — one built for
// this assignment list (the public-input cells belong to it when the public inputs are supplied)
int ensure_plan_for(qpgpu_circuit *c, const uint64_t *cells, size_t count, bool with_pis) {
    if (c->wplan && c->wplan->prep) {      // the list this plan's assignments were prepared for: the plan was chosen for it
        const PartialPrep &pp = *c->wplan->prep;
        if (pp.valid && pp.with_pis == with_pis && pp.cells.size() == count && (!count || std::memcmp(pp.cells.data(), cells, count * 8) == 0)) return QPGPU_OK;
    }
    std::vector<u64> assigned;
    if (with_pis) assigned.assign(c->pack.pi_cells.begin(), c->pack.pi_cells.end());
    const size_t pi_prefix = assigned.size();
    assigned.insert(assigned.end(), cells, cells + count);
    return ensure_plan(c, &assigned, pi_prefix);
}

}  // namespace

extern "C" {

// the trace rows a gate type occupies (introspection for tools and tests: the gate of a row is the one its selector column names)
int qpgpu_circuit_gate_rows(const qpgpu_circuit *c, unsigned gate_type, uint32_t *rows_out, size_t cap, size_t *count) {
    if (!c || !count) return QPGPU_EINVAL;
    const CircuitPack &p = c->pack;
    const u64 n = p.n();
    size_t k = 0;
    for (u64 r = 0; r < n; r++)
        for (size_t gi = 0; gi < p.gates.size(); gi++)
            if (p.constants_sigmas[(u64)p.gates[gi].selector_index * n + r] == gi) {
                if (p.gates[gi].type == gate_type) { if (rows_out && k < cap) rows_out[k] = (uint32_t)r; k++; }
                break;
            }
    *count = k;
    return rows_out && k > cap ? QPGPU_EBUFSIZE : QPGPU_OK;
}

int qpgpu_generate_witness_batch_dev(qpgpu_circuit *c, uint64_t *d_wires, uint32_t batch, const uint64_t *public_inputs) {
    if (!c) return QPGPU_EINVAL;
    qpgpu_ctx *ctx = c->ctx;
    QP_DEV(ctx);
    if (!d_wires || batch == 0 || batch > 65535 || (!public_inputs && c->pack.num_public_inputs)) return ctx->fail(QPGPU_EINVAL, "generate_witness: bad argument");
    QP_TRY(ensure_plan(c));
    return generate_batch(c, d_wires, batch, public_inputs, nullptr, nullptr, nullptr);
}

int qpgpu_generate_witness_dev(qpgpu_circuit *c, uint64_t *d_wires, const uint64_t *public_inputs) {
    return qpgpu_generate_witness_batch_dev(c, d_wires, 1, public_inputs);
}

// plonky2's generate_partial_witness on sparse PartialWitnesses: (cell, value) assignments instead of wire matrices.
// A target that is set twice with different values — two assignments in one copy class, an assignment that disagrees
// with what a generator (or the public-input argument) produces for that target, or two generators that disagree — is the
// reference's "set twice with different values" panic (wormhole/tests/src/circuit/block_header_tests.rs:34-95); here
// QPGPU_EUNSAT.
int qpgpu_witness_partial_prepare(qpgpu_circuit *c, const uint64_t *cells, size_t count, uint32_t max_batch) {
    if (!c) return QPGPU_EINVAL;
    qpgpu_ctx *ctx = c->ctx;
    QP_DEV(ctx);
    if ((count && !cells) || max_batch == 0 || max_batch > 65535) return ctx->fail(QPGPU_EINVAL, "witness_partial_prepare: bad argument");
    QP_TRY(ensure_plan_for(c, cells, count, true));
    QP_TRY(ensure_prep(c, cells, count, max_batch));
    WitnessPlan &plan = *c->wplan;
    // size what generation will need, so that the calls themselves neither allocate nor free
    if (plan.pi_cap < max_batch || plan.err_cap < max_batch) {
        QP_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (plan.d_pi_hash) { (void)hipFree(plan.d_pi_hash); plan.d_pi_hash = nullptr; }
        if (plan.d_pi_vals) { (void)hipFree(plan.d_pi_vals); plan.d_pi_vals = nullptr; }
        if (plan.d_err) { (void)hipFree(plan.d_err); plan.d_err = nullptr; }
        plan.pi_cap = 0; plan.err_cap = 0;
        QP_HIP(ctx, hipMalloc((void **)&plan.d_pi_hash, (size_t)max_batch * 32));
        QP_HIP(ctx, hipMalloc((void **)&plan.d_pi_vals, std::max<size_t>((size_t)max_batch * c->pack.num_public_inputs * 8, 8)));
        QP_HIP(ctx, hipMalloc((void **)&plan.d_err, (size_t)max_batch * 16));
        plan.pi_cap = max_batch; plan.err_cap = max_batch;
    }
    const PartialPrep &pp = *plan.prep;
    return ctx->reserve_read_back((size_t)max_batch * (c->pack.num_public_inputs + 4 + pp.scatter_idx.size() + pp.check_idx.size() + 1) * 8);
}

int qpgpu_generate_witness_partial_batch_dev(qpgpu_circuit *c, const uint64_t *cells, size_t count, const uint64_t *values,
                                             const uint64_t *public_inputs, uint32_t batch, uint64_t *d_wires, int *status) {
    if (!c) return QPGPU_EINVAL;
    qpgpu_ctx *ctx = c->ctx;
    QP_DEV(ctx);
    const CircuitPack &p = c->pack;
    if (!d_wires || batch == 0 || batch > 65535 || (count && (!cells || !values)))
        return ctx->fail(QPGPU_EINVAL, "generate_witness_partial: null argument");
    QP_TRY(ensure_plan_for(c, cells, count, public_inputs != nullptr));
    if (!public_inputs && p.num_public_inputs && !c->wplan->pi_gate_from_hash)
        return ctx->fail(QPGPU_EINVAL, "generate_witness_partial: public inputs can only be derived in a circuit whose PublicInputGate wires are copy-connected to the in-circuit hash of the public-input targets");
    QP_TRY(ensure_prep(c, cells, count, batch, public_inputs != nullptr));
    return generate_batch(c, d_wires, batch, public_inputs, c->wplan->prep, values, status);
}

int qpgpu_generate_witness_partial_batch_blinded_dev(qpgpu_circuit *c, const uint64_t *cells, size_t count, size_t n_blinding, const uint64_t *values,
                                                     const uint8_t *seeds, const uint64_t *public_inputs, uint32_t batch, uint64_t *d_wires, int *status) {
    if (!c) return QPGPU_EINVAL;
    qpgpu_ctx *ctx = c->ctx;
    QP_DEV(ctx);
    const CircuitPack &p = c->pack;
    if (!d_wires || batch == 0 || batch > 65535 || !cells || n_blinding > count || (count > n_blinding && !values))
        return ctx->fail(QPGPU_EINVAL, "generate_witness_partial_blinded: null argument or more blinding cells than cells");
    QP_TRY(ensure_plan_for(c, cells, count, public_inputs != nullptr));
    if (!public_inputs && p.num_public_inputs && !c->wplan->pi_gate_from_hash)
        return ctx->fail(QPGPU_EINVAL, "generate_witness_partial_blinded: public inputs can only be derived in a circuit whose PublicInputGate wires are copy-connected to the in-circuit hash of the public-input targets");
    QP_TRY(ensure_prep(c, cells, count, batch, public_inputs != nullptr));
    return generate_batch(c, d_wires, batch, public_inputs, c->wplan->prep, values, status, n_blinding, seeds);
}

int qpgpu_generate_witness_partial_dev(qpgpu_circuit *c, const uint64_t *cells, const uint64_t *values, size_t count,
                                       const uint64_t *public_inputs, uint64_t *d_wires) {
    return qpgpu_generate_witness_partial_batch_dev(c, cells, count, values, public_inputs, 1, d_wires, nullptr);
}

// ProverCircuitData::prove reads the public inputs out of the partition witness (`get_targets(&prover_data.public_inputs)`): the
// values at the public-input targets' cells of `batch` resident witnesses, [batch][num_public_inputs]
int qpgpu_witness_public_inputs_dev(qpgpu_circuit *c, const uint64_t *d_wires, uint32_t batch, uint64_t *public_inputs_out) {
    if (!c) return QPGPU_EINVAL;
    qpgpu_ctx *ctx = c->ctx;
    QP_DEV(ctx);
    const CircuitPack &p = c->pack;
    const size_t npis = p.num_public_inputs;
    if (!d_wires || batch == 0 || batch > 65535 || (npis && !public_inputs_out)) return ctx->fail(QPGPU_EINVAL, "witness_public_inputs: bad argument");
    if (npis == 0) return QPGPU_OK;
    if (p.pi_cells.size() != npis) return ctx->fail(QPGPU_EINVAL, "witness_public_inputs: the circuit pack carries no public-input cell trailer");
    if (!c->wplan) QP_TRY(ensure_plan(c));        // (the public-input cells are the same in a plan of either kind)
    WitnessPlan &plan = *c->wplan;
    if (plan.pi_cap < batch) {
        QP_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (plan.d_pi_vals) { (void)hipFree(plan.d_pi_vals); plan.d_pi_vals = nullptr; }
        if (plan.d_pi_hash) { (void)hipFree(plan.d_pi_hash); plan.d_pi_hash = nullptr; }
        plan.pi_cap = 0;
        QP_HIP(ctx, hipMalloc((void **)&plan.d_pi_hash, (size_t)batch * 32));
        QP_HIP(ctx, hipMalloc((void **)&plan.d_pi_vals, (size_t)batch * npis * 8));
        plan.pi_cap = batch;
    }
    const u64 stride = p.num_wires * p.n();
    for (uint32_t b = 0; b < batch; b++) QP_HIP(ctx, wk_gather(d_wires + (size_t)b * stride, plan.d_pi_idx, plan.d_pi_vals + (size_t)b * npis, (uint32_t)npis, ctx->stream));
    return ctx->read_back(public_inputs_out, plan.d_pi_vals, (size_t)batch * npis * 8);
}

int qpgpu_generate_witness(qpgpu_circuit *c, uint64_t *wires, const uint64_t *public_inputs) {
    if (!c) return QPGPU_EINVAL;
    qpgpu_ctx *ctx = c->ctx;
    QP_DEV(ctx);
    if (!wires) return ctx->fail(QPGPU_EINVAL, "generate_witness: null argument");
    const size_t bytes = (size_t)c->pack.num_wires * c->pack.n() * 8;
    QP_HIP(ctx, hipMemcpyAsync(c->d_wires_vals, wires, bytes, hipMemcpyHostToDevice, ctx->stream));
    QP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    QP_TRY(qpgpu_generate_witness_dev(c, c->d_wires_vals, public_inputs));
    QP_HIP(ctx, hipMemcpyAsync(wires, c->d_wires_vals, bytes, hipMemcpyDeviceToHost, ctx->stream));
    QP_HIP(ctx, hipMemsetAsync(c->d_wires_vals, 0, bytes, ctx->stream));   // the witness carries the spend secret
    QP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return QPGPU_OK;
}

}  // extern "C"
