// witness_kernels.hip — stage s1 (`generate_partial_witness`) on the device: the witness generators of the gates this
// backend evaluates, run level by level over the circuit's dependency order.
//
// Replaces, inside qp-plonky2 1.5.5 `plonk::prover::prove` (reference call site wormhole/prover/src/lib.rs:171-175), the
// generator loop of `iop::generator::generate_partial_witness` for the gate-attached generators: ConstantGenerator,
// ArithmeticBaseGenerator, ArithmeticExtensionGenerator, MulExtensionGenerator, BaseSplitGenerator, PoseidonGenerator,
// ReducingGenerator (both), RandomAccessGenerator, ExponentiationGenerator, PoseidonMdsGenerator,
// InterpolationGenerator. One thread runs one generator instance; copy constraints are resolved by reading every routed
// input through the source cell of its copy class (src_of), and a final pass copies sources to all class members.
#include <hip/hip_runtime.h>
#include "gl64.hpp"
#include "poseidon.hpp"
#include "witness.hpp"

using gl::e2;
using gl::u32;
using gl::u64;

namespace {

__device__ __forceinline__ void witness_level_body(WitnessArgs a, u32 first, u32 count, const u32 t) {
    if (t >= count) return;
    const WitnessInst in = a.insts[first + t];
    const u64 n = a.n;
    a.wires += (u64)blockIdx.y * a.batch_stride;        // one wire matrix per witness of the batch
    a.pi_hash += 4 * blockIdx.y;
    if (in.gate == WITNESS_HINT) {
        // a generator that is not attached to a gate: cells are named explicitly (routed wires), read through their copy class
        const u64 *h = a.hints + (u64)in.row * 8;
        const u32 NWc = a.num_wires, Rr = a.num_routed;
        auto RDc = [&](u64 cell) -> u64 { const u32 r = (u32)(cell / NWc), c = (u32)(cell % NWc); return gl::canon(a.wires[a.src_of[(u64)r * Rr + c]]); };
        auto WRc = [&](u64 cell, u64 v) { const u32 r = (u32)(cell / NWc), c = (u32)(cell % NWc); a.wires[(u64)c * n + r] = gl::canon(v); };
        switch (h[0]) {
        case 1: WRc(h[1], RDc(h[2])); break;                                                          // CopyGenerator
        case 2: { const u64 x = RDc(h[1]), y = RDc(h[2]);                                             // EqualityGenerator
                  WRc(h[3], x == y ? 1 : 0); WRc(h[4], x == y ? 0 : gl::inv(gl::sub(x, y))); break; }
        case 3: WRc(h[2], (RDc(h[1]) >> h[3]) & ((1ull << h[4]) - 1)); break;                         // WireSplitGenerator (one gate)
        case 4: { const e2 num = gl::e2_make(RDc(h[1]), RDc(h[2])), den = gl::e2_make(RDc(h[3]), RDc(h[4]));   // QuotientGeneratorExtension
                  const e2 q = gl::e2_mul(num, gl::e2_inv(den)); WRc(h[5], q.a); WRc(h[6], q.b); break; }
        case 5: WRc(h[1], h[2]); break;                                                               // ConstantGenerator
        case 6: { const u64 x = RDc(h[1]); WRc(h[2], x == 0 ? 1 : gl::inv(x)); break; }               // NonzeroTestGenerator
        case 7: { const u64 x = RDc(h[1]); WRc(h[2], x & ((1ull << h[4]) - 1)); WRc(h[3], x >> h[4]); break; }   // LowHighGenerator
        default: break;
        }
        return;
    }
    const GateDev g = a.gates[in.gate];
    const u32 R = a.num_routed, row = in.row, op = in.op;
    auto RD = [&](u32 col) -> u64 { return col < R ? a.wires[a.src_of[(u64)row * R + col]] : a.wires[(u64)col * n + row]; };
    auto WR = [&](u32 col, u64 v) { a.wires[(u64)col * n + row] = gl::canon(v); };
    auto RD2 = [&](u32 col) { const u64 x = RD(col), y = RD(col + 1); return gl::e2_make(x, y); };
    auto WR2 = [&](u32 col, e2 v) { WR(col, v.a); WR(col + 1, v.b); };
    const u64 *consts = a.cs + (u64)a.num_selectors * n + row;   // constant i of this row at consts[i * n]
    switch (g.type) {
    case 1: WR(op, consts[(u64)op * n]); break;                                         // ConstantGate
    case 2: for (u32 i = 0; i < 4; i++) WR(i, a.pi_hash[i]); break;                    // PublicInputGate
    case 3: {                                                                           // ArithmeticGate, one operation
        const u64 m0 = RD(4 * op), m1 = RD(4 * op + 1), ad = RD(4 * op + 2);
        WR(4 * op + 3, gl::add(gl::mul(gl::mul(m0, m1), consts[0]), gl::mul(ad, consts[n])));
        break;
    }
    case 6: {                                                                           // ArithmeticExtensionGate
        const e2 m0 = RD2(8 * op), m1 = RD2(8 * op + 2), ad = RD2(8 * op + 4);
        WR2(8 * op + 6, gl::e2_add(gl::e2_scale(gl::e2_mul(m0, m1), consts[0]), gl::e2_scale(ad, consts[n])));
        break;
    }
    case 7: WR2(6 * op + 4, gl::e2_scale(gl::e2_mul(RD2(6 * op), RD2(6 * op + 2)), consts[0])); break;   // MulExtensionGate
    case 5: {                                                                           // BaseSumGate<2>: limbs of the sum
        const u64 v = gl::canon(RD(0));
        for (u32 i = 0; i < g.param0; i++) WR(1 + i, (v >> i) & 1);
        break;
    }
    case 4: {                                                                           // PoseidonGate
        const u64 *rcs = a.poseidon_rc, *fpt = a.poseidon_fast;
        u64 st[12], inp[12];
        for (int i = 0; i < 12; i++) inp[i] = RD(i);
        const u64 swap = gl::canon(RD(24));
        for (int i = 0; i < 4; i++) {
            const u64 delta = swap ? gl::canon(gl::sub(inp[i + 4], inp[i])) : 0;
            WR(25 + i, delta);
            st[i] = gl::add(inp[i], delta); st[i + 4] = gl::sub(inp[i + 4], delta);
        }
        for (int i = 8; i < 12; i++) st[i] = inp[i];
        int rc = 0;
        for (int r = 0; r < 4; r++, rc++) {
            for (int i = 0; i < 12; i++) st[i] = gl::canon(gl::add(st[i], rcs[rc * 12 + i]));
            if (r) for (int i = 0; i < 12; i++) WR(29 + 12 * (r - 1) + i, st[i]);
            for (int i = 0; i < 12; i++) st[i] = poseidon::sbox7(st[i]);
            poseidon::mds_layer(st);
        }
        poseidon::fast_partial_enter(st, fpt);
        for (int r = 0; r < 22; r++) {
            st[0] = gl::canon(st[0]);
            WR(65 + r, st[0]);
            st[0] = poseidon::sbox7(st[0]);
            poseidon::fast_partial_linear(st, fpt, r);
        }
        rc += 22;
        for (int r = 0; r < 4; r++, rc++) {
            for (int i = 0; i < 12; i++) st[i] = gl::canon(gl::add(st[i], rcs[rc * 12 + i]));
            for (int i = 0; i < 12; i++) WR(87 + 12 * r + i, st[i]);
            for (int i = 0; i < 12; i++) st[i] = poseidon::sbox7(st[i]);
            poseidon::mds_layer(st);
        }
        for (int i = 0; i < 12; i++) WR(12 + i, st[i]);
        break;
    }
    case 8: case 9: {                                                                   // ReducingGate / ReducingExtensionGate
        const bool ext = g.type == 9;
        const u32 nc = g.param0, start_accs = 6 + (ext ? 2 * nc : nc);
        const e2 alpha = RD2(2);
        e2 acc = RD2(4);
        // the coefficients are fetched eight at a time before the chain that consumes them: a load issued inside the chain
        // waits behind the previous step's store (same array), and its latency would be paid once per coefficient
        for (u32 i0 = 0; i0 < nc; i0 += 8) {
            e2 cf[8];
#pragma unroll
            for (u32 k = 0; k < 8; k++) cf[k] = i0 + k < nc ? (ext ? RD2(6 + 2 * (i0 + k)) : gl::e2_from(RD(6 + i0 + k))) : gl::e2_from(0);
#pragma unroll
            for (u32 k = 0; k < 8; k++) {
                const u32 i = i0 + k;
                if (i < nc) {
                    acc = gl::e2_canon(gl::e2_add(gl::e2_mul(acc, alpha), cf[k]));
                    WR2(i == nc - 1 ? 0 : start_accs + 2 * i, acc);
                }
            }
        }
        break;
    }
    case 10: {                                                                          // RandomAccessGate: one copy, or the extra constants
        const u32 bits = g.param0, copies = g.param1, extra = g.param2, vec = 1u << bits, routed = (2 + vec) * copies + extra;
        if (op == copies) { for (u32 i = 0; i < extra; i++) WR((2 + vec) * copies + i, consts[(u64)i * n]); break; }
        const u32 b0 = (2 + vec) * op;
        const u64 idx = gl::canon(RD(b0)) & (vec - 1);
        WR(b0 + 1, RD(b0 + 2 + (u32)idx));
        for (u32 i = 0; i < bits; i++) WR(routed + op * bits + i, (idx >> i) & 1);
        break;
    }
    case 11: {                                                                          // ExponentiationGate
        const u32 nb = g.param0;
        const u64 base = RD(0);
        // all power bits first (independent loads, up to 128 of them), then the square-and-multiply chain without a load in it
        u64 bits_lo = 0, bits_hi = 0;
#pragma unroll 8
        for (u32 i = 0; i < nb; i++) {
            const u64 b = gl::canon(RD(1 + i)) & 1;
            if (i < 64) bits_lo |= b << i; else bits_hi |= b << (i - 64);
        }
        u64 cur = 1;
        for (u32 i = 0; i < nb; i++) {
            const u32 src = nb - 1 - i;
            const u64 prev = i == 0 ? 1 : gl::mul(cur, cur), bit = src < 64 ? (bits_lo >> src) & 1 : (bits_hi >> (src - 64)) & 1;
            cur = gl::canon(bit ? gl::mul(prev, base) : prev);
            WR(2 + nb + i, cur);
        }
        WR(1 + nb, cur);
        break;
    }
    case 12: {                                                                          // PoseidonMdsGate
        for (u32 comp = 0; comp < 2; comp++) {
            u64 st[12];
            for (int i = 0; i < 12; i++) st[i] = RD(2 * i + comp);
            poseidon::mds_layer(st);
            for (int i = 0; i < 12; i++) WR(24 + 2 * i + comp, st[i]);
        }
        break;
    }
    case 13: {                                                                          // CosetInterpolationGate
        const u32 bits = g.param0, deg = g.param1, np = 1u << bits, ni = (np - 2) / (deg - 1);
        const u32 s_ep = 1 + 2 * np, s_ev = s_ep + 2, s_int = s_ev + 2;
        const u64 shift = RD(0);
        const e2 sp = gl::e2_canon(gl::e2_scale(RD2(s_ep), gl::inv(shift)));
        WR2(s_int + 4 * ni, sp);
        const u64 omega = 1ull << (192u >> bits), inv_n = gl::P - ((1ull << (64 - bits)) - (1ull << (32 - bits)));
        e2 ev = gl::e2_from(0), pr = gl::e2_from(1);
        u64 x = 1;
        u32 lo = 0, hi = deg;
        for (u32 c = 0; c <= ni; c++) {
            e2 vals[8];                          // the chunk's values up front (a chunk has `deg` points, then deg - 1; longer chunks read the rest in place)
            const u32 cnt = hi - lo;
#pragma unroll
            for (u32 k = 0; k < 8; k++) vals[k] = k < cnt ? RD2(1 + 2 * (lo + k)) : gl::e2_from(0);
            auto step = [&](e2 v) {
                e2 term = sp; term.a = gl::sub(term.a, x);
                const e2 tv = gl::e2_scale(gl::e2_mul(v, pr), gl::mul(x, inv_n));
                ev = gl::e2_add(gl::e2_mul(ev, term), tv);
                pr = gl::e2_mul(pr, term);
                x = gl::mul(x, omega);
            };
#pragma unroll
            for (u32 k = 0; k < 8; k++) if (k < cnt) step(vals[k]);
            for (u32 q = lo + 8; q < hi; q++) step(RD2(1 + 2 * q));
            ev = gl::e2_canon(ev); pr = gl::e2_canon(pr);
            if (c == ni) break;
            WR2(s_int + 2 * c, ev); WR2(s_int + 2 * (ni + c), pr);
            lo = 1 + (deg - 1) * (c + 1); hi = lo + deg - 1 < np ? lo + deg - 1 : np;
        }
        WR2(s_ev, ev);
        break;
    }
    default: break;
    }
}

// PoseidonGate generator, lane-cooperative: one instance per 16 lanes (state element g in lane g, 12 used), S-box layer in
// parallel, MDS gathered with wave shuffles. A single thread needs ~100 us for the 30 rounds, and a dependency level ends
// when its slowest generator does. The S-box inputs recorded for the partial rounds are those of the textbook schedule:
// the fast-basis formulation the gate's constraints use feeds the same values to the S-box.
// rc: the 360 round constants staged in LDS by the calling kernel (stage_round_constants): the round loop is one dependent
// chain per row, and a global load per round would put an L2 round trip on it thirty times.
__device__ __forceinline__ void stage_round_constants(const WitnessArgs &a, u64 *rc_lds) {
    for (u32 i = threadIdx.x; i < poseidon::ROUNDS * 12; i += blockDim.x) rc_lds[i] = a.poseidon_rc[i];
    if (a.p2_gate) {   // the Poseidon2 gate's parameter block behind the Poseidon constants (same reason)
        const u64 *src = reinterpret_cast<const u64 *>(a.p2_gate);
        for (u32 i = threadIdx.x; i < poseidon2::PARAM_WORDS; i += blockDim.x) rc_lds[poseidon::ROUNDS * 12 + i] = src[i];
    }
    __syncthreads();
}
constexpr u32 WITNESS_CONST_WORDS = poseidon::ROUNDS * 12 + poseidon2::PARAM_WORDS;

// Poseidon2 gate generator (gate type 14), lane-cooperative like the PoseidonGate one: state element g in lane g of a 16-lane
// group. External layer: every lane takes its 4x4 block row over its group of four (4 shuffles), then adds the column sum over
// the three blocks (3 shuffles). Internal layer: the state sum by a 4-step butterfly over the 16 lanes (lanes 12..15 hold 0),
// then s * diag + sum. The S-box inputs go to the wires the layout names; they are what the gate's constraints pin.
__device__ __forceinline__ void witness_poseidon2_row(WitnessArgs a, const WitnessInst in, const bool live, const u64 *p2w) {
    const P2GateLayout &lay = a.p2_layout;
    const u64 *rc_ext = p2w, *rc_int = p2w + 96, *diag = p2w + 118;
    const int g = threadIdx.x & 15, lane_base = (threadIdx.x & 63) & ~15;
    const u64 n = a.n;
    const u32 R = a.num_routed, row = in.row;
    a.wires += (u64)blockIdx.y * a.batch_stride;
    auto RD = [&](u32 col) -> u64 { return col < R ? a.wires[a.src_of[(u64)row * R + col]] : a.wires[(u64)col * n + row]; };
    auto WR = [&](u32 col, u64 v) { if (live) a.wires[(u64)col * n + row] = gl::canon(v); };
    auto shfl64 = [&](u64 v, int src) { return ((u64)(u32)__shfl((int)(v >> 32), src, 64) << 32) | (u32)__shfl((int)(u32)v, src, 64); };
    const bool act = g < 12;
    u64 s = act ? RD(lay.w_input + g) : 0;
    if (lay.has_swap()) {
        const u64 swap = gl::canon(RD(lay.w_swap));
        const u64 partner = shfl64(s, lane_base + (g < 4 ? g + 4 : (g < 8 ? g - 4 : g)));
        if (g < 4) { const u64 delta = swap ? gl::canon(gl::sub(partner, s)) : 0; WR(lay.w_delta + g, delta); s = gl::add(s, delta); }
        else if (g < 8) { const u64 delta = swap ? gl::canon(gl::sub(s, partner)) : 0; s = gl::sub(s, delta); }
    }
    // cross-lane moves inside a 16-lane group as DPP operands of the vector ALU (quad permutes, rotations of a row of 16) instead of
    // ds_bpermute round trips through the LDS crossbar: a row generator is one dependent chain, and a crossbar shuffle puts ~100 cycles
    // of latency into it where a DPP move costs an instruction slot. Lanes 12..15 of a group hold zero throughout.
    constexpr int XOR1 = 0xB1, XOR2 = 0x4E, NEXT_IN_QUAD = 0x39, ROR4 = 0x124, ROR8 = 0x128, ROR12 = 0x12C;
#define P2_DPP64(v, ctrl) (((u64)(u32)__builtin_amdgcn_update_dpp(0, (int)((v) >> 32), ctrl, 0xF, 0xF, false) << 32) | (u32)__builtin_amdgcn_update_dpp(0, (int)(u32)(v), ctrl, 0xF, 0xF, false))
    // external layer with qp-poseidon-core's block circ(2, 3, 1, 1): row `col` of a block is (sum of the four) + own + 2 * next
    auto ext = [&](u64 v) -> u64 {
        u64 sum = gl::add(v, P2_DPP64(v, XOR1));
        sum = gl::add(sum, P2_DPP64(sum, XOR2));
        const u64 nx = P2_DPP64(v, NEXT_IN_QUAD);
        const u64 t = gl::add(gl::add(sum, v), gl::add(nx, nx));
        const u64 cs = gl::add(gl::add(t, P2_DPP64(t, ROR4)), gl::add(P2_DPP64(t, ROR8), P2_DPP64(t, ROR12)));   // column sum over the three blocks (+ the idle block's zero)
        return act ? gl::add(t, cs) : 0;
    };
    s = ext(s);
    u32 wf = lay.w_full0;
#pragma unroll 1
    for (int r = 0; r < 4; r++) {
        s = gl::canon(gl::add(s, act ? rc_ext[r * 12 + g] : 0));
        if (r || lay.first_round_wires) { if (act) WR(wf + g, s); wf += 12; }
        s = ext(poseidon::sbox7(s));
    }
    const u64 dg = act ? diag[g] : 0;
#pragma unroll 1
    for (int r = 0; r < 22; r++) {
        if (g == 0) { s = gl::canon(gl::add(s, rc_int[r])); WR(lay.w_partial + r, s); s = poseidon::sbox7(s); }
        u64 sum = gl::add(s, P2_DPP64(s, XOR1));
        sum = gl::add(sum, P2_DPP64(sum, XOR2));
        sum = gl::add(sum, P2_DPP64(sum, ROR4));
        sum = gl::add(sum, P2_DPP64(sum, ROR8));
        s = act ? gl::add(gl::mul(s, dg), sum) : 0;
    }
#pragma unroll 1
    for (int r = 0; r < 4; r++) {
        s = gl::canon(gl::add(s, act ? rc_ext[(4 + r) * 12 + g] : 0));
        if (act) WR(lay.w_full1 + 12 * r + g, s);
        s = ext(poseidon::sbox7(s));
    }
    if (act) WR(lay.w_output + g, s);
#undef P2_DPP64
}
__device__ __forceinline__ void witness_poseidon_body(WitnessArgs a, u32 first, u32 count, const u32 tid, const u64 *rc) {
    constexpr u32 C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    const u32 slot = tid >> 4;
    const int g = threadIdx.x & 15, lane_base = (threadIdx.x & 63) & ~15;
    const bool live = slot < count;
    const WitnessInst in = a.insts[first + (live ? slot : count - 1)];   // idle groups shadow the last instance, without stores
    // a wave's four 16-lane groups may hold rows of either hash gate; both bodies use wave shuffles, so the wave runs
    // them one after the other with all of its lanes (a group whose row is of the other kind computes on it without storing)
    const bool is_p2 = a.gates[in.gate].type == 14;
    if (__any(is_p2)) witness_poseidon2_row(a, in, live && is_p2, rc + poseidon::ROUNDS * 12);
    if (!__any(!is_p2)) return;
    const bool live1 = live && !is_p2;
    const u64 n = a.n;
    const u32 R = a.num_routed, row = in.row;
    a.wires += (u64)blockIdx.y * a.batch_stride;
    auto RD = [&](u32 col) -> u64 { return col < R ? a.wires[a.src_of[(u64)row * R + col]] : a.wires[(u64)col * n + row]; };
    auto WR = [&](u32 col, u64 v) { if (live1) a.wires[(u64)col * n + row] = gl::canon(v); };
    auto shfl64 = [&](u64 v, int src) { return ((u64)(u32)__shfl((int)(v >> 32), src, 64) << 32) | (u32)__shfl((int)(u32)v, src, 64); };
    int src[12];
#pragma unroll
    for (int i = 0; i < 12; i++) { int e = g + i; e -= e >= 12 ? 12 : 0; src[i] = lane_base + (g < 12 ? e : i); }
    u64 s = g < 12 ? RD(g) : 0;
    const u64 swap = gl::canon(RD(24));
    {
        const u64 partner = shfl64(s, lane_base + (g < 4 ? g + 4 : (g < 8 ? g - 4 : g)));
        if (g < 4) { const u64 delta = swap ? gl::canon(gl::sub(partner, s)) : 0; WR(25 + g, delta); s = gl::add(s, delta); }
        else if (g < 8) { const u64 delta = swap ? gl::canon(gl::sub(s, partner)) : 0; s = gl::sub(s, delta); }
    }
#pragma unroll 1
    for (int r = 0; r < poseidon::ROUNDS; r++) {
        s = gl::canon(gl::add(s, g < 12 ? rc[r * 12 + g] : 0));
        const bool full = r < poseidon::HALF_FULL || r >= poseidon::HALF_FULL + poseidon::PARTIAL;
        if (g < 12) {
            if (full && r >= 1 && r < poseidon::HALF_FULL) WR(29 + 12 * (r - 1) + g, s);
            if (full && r >= poseidon::HALF_FULL + poseidon::PARTIAL) WR(87 + 12 * (r - poseidon::HALF_FULL - poseidon::PARTIAL) + g, s);
            if (!full && g == 0) WR(65 + (r - poseidon::HALF_FULL), s);
        }
        const u64 sb = poseidon::sbox7(s);
        s = (full || g == 0) ? sb : s;
        const u32 lo = (u32)s, hi = (u32)(s >> 32);
        u64 al = 0, ah = 0;
#pragma unroll
        for (int i = 0; i < 12; i++) {
            const u32 l = (u32)__shfl((int)lo, src[i], 64), h = (u32)__shfl((int)hi, src[i], 64);
            al += (u64)l * C[i];
            ah += (u64)h * C[i];
        }
        if (g == 0) { al += (u64)lo * 8u; ah += (u64)hi * 8u; }
        const u64 low = al + (ah << 32);
        const u32 top = (u32)(ah >> 32) + (low < al ? 1u : 0u);
        s = gl::reduce96(low, top);
    }
    if (g < 12) WR(12 + g, s);
}

__global__ void __launch_bounds__(128) witness_level_kernel(WitnessArgs a, u32 first, u32 count) {
    witness_level_body(a, first, count, blockIdx.x * blockDim.x + threadIdx.x);
}
__global__ void __launch_bounds__(256) witness_poseidon_kernel(WitnessArgs a, u32 first, u32 count) {
    __shared__ u64 rc_lds[WITNESS_CONST_WORDS];
    stage_round_constants(a, rc_lds);
    witness_poseidon_body(a, first, count, blockIdx.x * blockDim.x + threadIdx.x, rc_lds);
}
// One dependency level in one launch: the first `generic_blocks` workgroups run the level's ordinary generator instances (a
// thread each), the rest its PoseidonGate rows (16 lanes each). Both halves are latency-bound (a level ends when its slowest
// generator does), so running them side by side instead of back to back takes the longer of the two, not the sum.
__global__ void __launch_bounds__(256) witness_combined_kernel(WitnessArgs a, u32 first, u32 n_generic, u32 n_poseidon, u32 generic_blocks) {
    __shared__ u64 rc_lds[WITNESS_CONST_WORDS];
    if (blockIdx.x < generic_blocks) { witness_level_body(a, first, n_generic, blockIdx.x * blockDim.x + threadIdx.x); return; }
    stage_round_constants(a, rc_lds);
    witness_poseidon_body(a, first + n_generic, n_poseidon, (blockIdx.x - generic_blocks) * blockDim.x + threadIdx.x, rc_lds);
}

// A run of narrow dependency levels in one launch. A level of a deep circuit is a handful of generator instances (a serial
// chain of reducing / arithmetic operations, a Merkle path's hashes), and one launch per level costs 20-40 us of launch and
// drain latency against a few us of work. Here one workgroup per witness walks levels [l0, l1): waves 0-7 take the level's
// ordinary instances (a thread each, strided), waves 8-15 its PoseidonGate rows (16 lanes each), then a workgroup barrier —
// all waves of a workgroup share one L1, so the barrier's memory fence makes the level's stores visible to the next level.
__global__ void __launch_bounds__(1024) witness_run_kernel(WitnessArgs a, const u32 *level_start, const u32 *level_poseidon, u32 l0, u32 l1) {
    __shared__ u64 rc_lds[WITNESS_CONST_WORDS];
    stage_round_constants(a, rc_lds);
    const u32 t = threadIdx.x;
    for (u32 l = l0; l < l1; l++) {
        const u32 lo = level_start[l], mid = level_poseidon[l], hi = level_start[l + 1];
        if (t < 512) {
            for (u32 i = t; i < mid - lo; i += 512) witness_level_body(a, lo, mid - lo, i);
        } else if (hi > mid) {
            for (u32 base = 0; base < (hi - mid) * 16; base += 512) witness_poseidon_body(a, mid, hi - mid, base + (t - 512), rc_lds);
        }
        __threadfence_block();
        __syncthreads();
    }
}

// every routed cell takes the value of its copy class's source cell
__global__ void __launch_bounds__(256) witness_fill_kernel(WitnessArgs a) {
    const u64 cell = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (cell >= a.n * a.num_routed) return;
    a.wires += (u64)blockIdx.y * a.batch_stride;
    const u64 row = cell / a.num_routed, col = cell % a.num_routed, own = col * a.n + row;
    const u32 src = a.src_of[cell];
    if (src != own) a.wires[own] = a.wires[src];
}

// wires[b][idx[i]] = vals[b][i]: the caller's assignments (PartialWitness::set_target) and the public-input cells
__global__ void __launch_bounds__(256) witness_scatter_kernel(u64 *wires, const u32 *idx, const u64 *vals, u32 count, u64 batch_stride, u32 val_stride) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    wires[(u64)blockIdx.y * batch_stride + idx[i]] = gl::canon(vals[(u64)blockIdx.y * val_stride + i]);
}
__global__ void __launch_bounds__(256) witness_gather_kernel(const u64 *wires, const u32 *idx, u64 *out, u32 count) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) out[i] = gl::canon(wires[idx[i]]);
}

// a partition set twice (see witness.hpp)
__global__ void __launch_bounds__(256) witness_check_pairs_kernel(const u64 *wires, const u32 *own, const u32 *src, u32 count, u64 batch_stride, u32 *err, u32 slot) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const u64 *w = wires + (u64)blockIdx.y * batch_stride;
    if (gl::canon(w[own[i]]) != gl::canon(w[src[i]])) atomicMin(&err[4 * blockIdx.y + slot], i);
}
// a generator's divisor that is zero (see WitnessPlan::h_nz_a)
__global__ void __launch_bounds__(256) witness_check_nonzero_kernel(const u64 *wires, const u32 *a, const u32 *b, u32 count, u64 batch_stride, u32 *err, u32 slot) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const u64 *w = wires + (u64)blockIdx.y * batch_stride;
    if (gl::canon(w[a[i]]) == 0 && gl::canon(w[b[i]]) == 0) atomicMin(&err[4 * blockIdx.y + slot], i);
}
__global__ void __launch_bounds__(256) witness_check_vals_kernel(const u64 *wires, const u32 *idx, const u64 *vals, u32 count, u64 batch_stride, u32 val_stride, u32 *err, u32 slot) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    if (gl::canon(wires[(u64)blockIdx.y * batch_stride + idx[i]]) != gl::canon(vals[(u64)blockIdx.y * val_stride + i])) atomicMin(&err[4 * blockIdx.y + slot], i);
}

}  // namespace

hipError_t wk_check_nonzero(const uint64_t *wires, const uint32_t *a, const uint32_t *b, uint32_t count, uint32_t batch, uint64_t batch_stride, uint32_t *err, uint32_t slot, hipStream_t st) {
    if (!count || !batch) return hipSuccess;
    hipLaunchKernelGGL(witness_check_nonzero_kernel, dim3((count + 255) / 256, batch), dim3(256), 0, st, wires, a, b, count, batch_stride, err, slot);
    return hipGetLastError();
}
hipError_t wk_check_pairs(const uint64_t *wires, const uint32_t *own, const uint32_t *src, uint32_t count, uint32_t batch, uint64_t batch_stride, uint32_t *err, uint32_t slot, hipStream_t st) {
    if (count == 0 || batch == 0) return hipSuccess;
    hipLaunchKernelGGL(witness_check_pairs_kernel, dim3((count + 255) / 256, batch), dim3(256), 0, st, wires, own, src, count, batch_stride, err, slot);
    return hipGetLastError();
}
hipError_t wk_check_vals(const uint64_t *wires, const uint32_t *idx, const uint64_t *vals, uint32_t count, uint32_t batch, uint64_t batch_stride, uint32_t val_stride, uint32_t *err, uint32_t slot, hipStream_t st) {
    if (count == 0 || batch == 0) return hipSuccess;
    hipLaunchKernelGGL(witness_check_vals_kernel, dim3((count + 255) / 256, batch), dim3(256), 0, st, wires, idx, vals, count, batch_stride, val_stride, err, slot);
    return hipGetLastError();
}

hipError_t wk_scatter(uint64_t *wires, const uint32_t *idx, const uint64_t *vals, uint32_t count, uint32_t batch, uint64_t batch_stride, uint32_t val_stride, hipStream_t st) {
    if (count == 0 || batch == 0) return hipSuccess;
    hipLaunchKernelGGL(witness_scatter_kernel, dim3((count + 255) / 256, batch), dim3(256), 0, st, wires, idx, vals, count, batch_stride, val_stride);
    return hipGetLastError();
}
hipError_t wk_gather(const uint64_t *wires, const uint32_t *idx, uint64_t *out, uint32_t count, hipStream_t st) {
    if (count == 0) return hipSuccess;
    hipLaunchKernelGGL(witness_gather_kernel, dim3((count + 255) / 256), dim3(256), 0, st, wires, idx, out, count);
    return hipGetLastError();
}
hipError_t wk_run_level(const WitnessArgs &a, uint32_t first, uint32_t count, uint32_t batch, hipStream_t st) {
    if (count == 0 || batch == 0) return hipSuccess;
    hipLaunchKernelGGL(witness_level_kernel, dim3((count + 127) / 128, batch), dim3(128), 0, st, a, first, count);
    return hipGetLastError();
}
hipError_t wk_run_poseidon(const WitnessArgs &a, uint32_t first, uint32_t count, uint32_t batch, hipStream_t st) {
    if (count == 0 || batch == 0) return hipSuccess;
    hipLaunchKernelGGL(witness_poseidon_kernel, dim3((count + 15) / 16, batch), dim3(256), 0, st, a, first, count);
    return hipGetLastError();
}
hipError_t wk_run_combined(const WitnessArgs &a, uint32_t first, uint32_t n_generic, uint32_t n_poseidon, uint32_t batch, hipStream_t st) {
    if (batch == 0 || (n_generic == 0 && n_poseidon == 0)) return hipSuccess;
    if (n_poseidon == 0) return wk_run_level(a, first, n_generic, batch, st);
    if (n_generic == 0) return wk_run_poseidon(a, first, n_poseidon, batch, st);
    const uint32_t gb = (n_generic + 255) / 256, pb = (n_poseidon + 15) / 16;
    hipLaunchKernelGGL(witness_combined_kernel, dim3(gb + pb, batch), dim3(256), 0, st, a, first, n_generic, n_poseidon, gb);
    return hipGetLastError();
}
hipError_t wk_run_levels(const WitnessArgs &a, const uint32_t *d_level_start, const uint32_t *d_level_poseidon, uint32_t l0, uint32_t l1, uint32_t batch, hipStream_t st) {
    if (batch == 0 || l1 <= l0) return hipSuccess;
    hipLaunchKernelGGL(witness_run_kernel, dim3(1, batch), dim3(1024), 0, st, a, d_level_start, d_level_poseidon, l0, l1);
    return hipGetLastError();
}
hipError_t wk_fill_copies(const WitnessArgs &a, uint32_t batch, hipStream_t st) {
    const uint64_t cells = a.n * a.num_routed;
    if (batch == 0) return hipSuccess;
    hipLaunchKernelGGL(witness_fill_kernel, dim3((unsigned)((cells + 255) / 256), batch), dim3(256), 0, st, a);
    return hipGetLastError();
}
