// prover_kernels.hip — stages s5..s11 of plonky2's prove() as gfx950 kernels.
//
// Replaces, inside qp-plonky2 1.5.5 `plonk::prover::prove` (reference call site
// wormhole/prover/src/lib.rs:171-175):
//   s5  all_wires_permutation_partial_products   -> pp_rows_kernel, pp_scan_kernel, pp_finish_kernel
//   s6  compute_quotient_polys                   -> quotient_perm_kernel, quotient_gates_kernel, quotient_poseidon_kernel
//   s7  OpeningSet::new (poly evaluation at zeta) -> poly_eval_kernel
//   s8  prove_openings (batch reduce, /(X - z))  -> reduce_polys_kernel, divide_linear_kernel
//   s9  fri_committed_trees (fold)               -> fri_fold_kernel, interleave_ext_kernel
//   s10 fri_proof_of_work                        -> pow_kernel
//   s11 fri_prover_query_rounds (gathers)        -> gather kernels
// Layout: polynomial batches column-major; LDEs in leaf order (slot j = point bitrev(j)), so every
// per-point kernel reads one slot of every column with unit-stride across the wave.
#include <hip/hip_runtime.h>
#include "gl64.hpp"
#include "poseidon.hpp"
#include "prover_kernels.hpp"

using gl::e2;
using gl::u32;
using gl::u64;

namespace {

__device__ __forceinline__ u32 brev32(u32 x, u32 bits) { return bits ? __brev(x) >> (32 - bits) : 0; }

// ---------------------------------------------------------------- s5
// Row i: quotient chunk products prod_{j in chunk} (w_j + beta k_j x + gamma) / (w_j + beta sigma_j + gamma).
// qcp layout: [challenge][chunk][row]; rowprod: [challenge][row].
__global__ void __launch_bounds__(256) pp_rows_kernel(PpArgs a) {
    const u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const u32 k = blockIdx.y;
    { const u64 pr = blockIdx.z;   // proof of the batch
      a.wires += pr * a.ps_wires; a.betas += pr * a.ps_small; a.gammas += pr * a.ps_small; a.beta_k_is += pr * a.ps_small;
      a.qcp += pr * a.ps_qcp; a.rowprod += pr * a.ps_rowprod; }
    const u64 beta = a.betas[k], gamma = a.gammas[k];
    const u64 x = a.omega_pows[i];
    const u32 R = a.num_routed, chunk = a.chunk, nchunks = a.nchunks;
    u64 nums[16], dens[16], pref[16];   // nchunks <= 16
    for (u32 cc = 0; cc < nchunks; cc++) {
        u64 pn = 1, pd = 1;
        for (u32 j = cc * chunk; j < (cc + 1) * chunk && j < R; j++) {
            const u64 w = a.wires[(u64)j * a.n + i];
            const u64 sid = gl::mul(a.beta_k_is[k * R + j], x);          // beta * k_j * x
            const u64 ssg = gl::mul(beta, a.sigmas[(u64)j * a.n + i]);   // beta * sigma_j(x)
            pn = gl::mul(pn, gl::add(gl::add(w, sid), gamma));
            pd = gl::mul(pd, gl::add(gl::add(w, ssg), gamma));
        }
        nums[cc] = pn; dens[cc] = pd;
    }
    // Montgomery batch inversion of the chunk denominators
    u64 acc = 1;
    for (u32 cc = 0; cc < nchunks; cc++) { pref[cc] = acc; acc = gl::mul(acc, dens[cc]); }
    u64 inv = gl::inv(acc);
    u64 rowp = 1;
    for (u32 cc = nchunks; cc-- > 0;) {
        const u64 dinv = gl::mul(inv, pref[cc]);
        inv = gl::mul(inv, dens[cc]);
        nums[cc] = gl::mul(nums[cc], dinv);
    }
    for (u32 cc = 0; cc < nchunks; cc++) {
        rowp = gl::mul(rowp, nums[cc]);
        a.qcp[((u64)k * nchunks + cc) * a.n + i] = gl::canon(nums[cc]);
    }
    a.rowprod[(u64)k * a.n + i] = gl::canon(rowp);
}

// Exclusive prefix product over rows (Z(x_0) = 1). One workgroup of 1024 threads per challenge.
__global__ void __launch_bounds__(1024) pp_scan_kernel(const u64 *rowprod, u64 *z_out, u64 n) {
    __shared__ u64 part[1024];
    const u32 t = threadIdx.x, T = blockDim.x;
    const u64 *rp = rowprod + (u64)blockIdx.x * n;
    u64 *z = z_out + (u64)blockIdx.x * n;
    const u64 per = (n + T - 1) / T, lo = (u64)t * per, hi = lo + per < n ? lo + per : n;
    u64 acc = 1;
    for (u64 i = lo; i < hi; i++) acc = gl::mul(acc, rp[i]);
    part[t] = acc;
    __syncthreads();
    for (u32 off = 1; off < T; off <<= 1) {   // inclusive Hillis-Steele scan of the partial products
        u64 v = part[t];
        if (t >= off) v = gl::mul(v, part[t - off]);
        __syncthreads();
        part[t] = v;
        __syncthreads();
    }
    acc = t == 0 ? 1 : part[t - 1];
    for (u64 i = lo; i < hi; i++) { z[i] = gl::canon(acc); acc = gl::mul(acc, rp[i]); }
}

// zs_pp columns: [Z_0..Z_{nch-1}, pp_{0,*}, pp_{1,*}, ...]; pp_{k,c}(x_i) = Z_k(x_i) * prod_{c' <= c} qcp
__global__ void __launch_bounds__(256) pp_finish_kernel(PpArgs a, const u64 *z, u64 *zs_pp, u64 ps_z, u64 ps_zs) {
    const u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    { const u64 pr = blockIdx.z; a.qcp += pr * a.ps_qcp; z += pr * ps_z; zs_pp += pr * ps_zs; }
    const u32 k = blockIdx.y, npp = a.nchunks - 1;
    u64 acc = z[(u64)k * a.n + i];
    zs_pp[(u64)k * a.n + i] = acc;
    for (u32 cc = 0; cc < npp; cc++) {
        acc = gl::mul(acc, a.qcp[((u64)k * a.nchunks + cc) * a.n + i]);
        zs_pp[((u64)a.nch + (u64)k * npp + cc) * a.n + i] = gl::canon(acc);
    }
}

// ---------------------------------------------------------------- s6
__device__ __forceinline__ u64 gate_filter(const QuotientArgs &a, u32 gi, u64 s) {
    const GateDev g = a.gates[gi];
    u64 f = 1;
    for (u32 j = g.group_start; j < g.group_end; j++)
        if (j != gi) f = gl::mul(f, gl::sub((u64)j, s));
    if (a.num_selectors > 1) f = gl::mul(f, gl::sub(0xFFFFFFFFull, s));
    return f;
}

// ---- s6 is three kernels over the LDE slots (thread = slot j, point index i = bitrev(j)), each adding its
// alpha-weighted terms into acc[c][j] (slot order, unit stride); the last one multiplies by 1/Z_H(x) and stores the
// quotient values in natural order for the inverse NTT. NCH (number of challenges) is a compile-time constant so the
// per-challenge accumulators live in registers.

// proof blockIdx.z of a lockstep batch: move the per-proof pointers
__device__ __forceinline__ void quotient_select_proof(QuotientArgs &a) {
    const u64 pr = blockIdx.z;
    a.wires += pr * a.ps_wires; a.zs_pp += pr * a.ps_zs;
    a.alpha_pows += pr * a.ps_small; a.beta_k_is += pr * a.ps_small; a.betas += pr * a.ps_small; a.gammas += pr * a.ps_small; a.pi_hash += pr * a.ps_small;
    a.acc += pr * a.ps_acc; a.out += pr * a.ps_out;
}

// (1) L_0(x)(Z(x) - 1) and the partial-product checks
template <int NCH>
__global__ void __launch_bounds__(256) quotient_perm_kernel(QuotientArgs a) {
    const u64 j = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (j >= a.q_n) return;
    quotient_select_proof(a);
    const u32 logL = a.log_lde, R = a.num_routed, chunk = a.chunk, nchunks = a.nchunks, npp = nchunks - 1;
    const u64 i = brev32((u32)j, logL);
    const u64 jn = brev32((u32)((i + a.rate) & (a.lde_n - 1)), logL);   // slot of the next row g*x
    const u64 x = a.x_coset[j], l0 = a.l0_coset[j], S = a.lde_n;
    gl::Acc192 acc[NCH];      // alpha-weighted sums as unreduced 192-bit accumulators (gl64.hpp): one reduction per challenge at the end
#pragma unroll
    for (int c = 0; c < NCH; c++) acc[c] = gl::acc_zero();
    u32 t = 0;
#pragma unroll
    for (int k = 0; k < NCH; k++, t++) {
        const u64 term = gl::mul(l0, gl::sub(a.zs_pp[(u64)k * S + j], 1));
#pragma unroll
        for (int c = 0; c < NCH; c++) gl::acc_mul(acc[c], term, a.alpha_pows[(u64)c * a.nterms + t]);
    }
#pragma unroll
    for (int k = 0; k < NCH; k++) {
        const u64 beta = a.betas[k], gamma = a.gammas[k];
        u64 prev = a.zs_pp[(u64)k * S + j];
        for (u32 cc = 0; cc < nchunks; cc++, t++) {
            u64 pn = 1, pd = 1;
            for (u32 r = cc * chunk; r < (cc + 1) * chunk && r < R; r++) {
                const u64 w = a.wires[(u64)r * S + j];
                const u64 sid = gl::mul(a.beta_k_is[k * R + r], x);
                const u64 ssg = gl::mul(beta, a.cs[(u64)(a.sig0 + r) * S + j]);
                pn = gl::mul(pn, gl::add(gl::add(w, sid), gamma));
                pd = gl::mul(pd, gl::add(gl::add(w, ssg), gamma));
            }
            const u64 next = cc == nchunks - 1 ? a.zs_pp[(u64)k * S + jn] : a.zs_pp[((u64)NCH + (u64)k * npp + cc) * S + j];
            const u64 term = gl::sub(gl::mul(prev, pn), gl::mul(next, pd));
#pragma unroll
            for (int c = 0; c < NCH; c++) gl::acc_mul(acc[c], term, a.alpha_pows[(u64)c * a.nterms + t]);
            prev = next;
        }
    }
#pragma unroll
    for (int c = 0; c < NCH; c++) a.acc[(u64)c * S + j] = gl::acc_reduce(acc[c]);
}

// One copy of a RandomAccessGate with 2^BITS list entries: out[0..BITS) the bit constraints, out[BITS] the index
// reconstruction, out[BITS+1] the list folded by the bits against the claimed element. BITS is a compile-time constant so
// the item array stays in registers (a dynamically indexed array went to scratch). The 32-entry form, which standard
// configurations do not use (arity 16, cap height 4), lives in its own kernel instance (WIDE) so that it does not set the
// register count of the common one.
template <int BITS>
__device__ __forceinline__ void random_access_values(const u64 *cw, const u64 *bw, u64 S, u64 *out) {
    constexpr int VEC = 1 << BITS;
    u64 items[VEC], bit[BITS];
#pragma unroll
    for (int i = 0; i < VEC; i++) items[i] = cw[(u64)(2 + i) * S];
#pragma unroll
    for (int i = 0; i < BITS; i++) { bit[i] = bw[(u64)i * S]; out[i] = gl::mul(bit[i], gl::sub(bit[i], 1)); }
    u64 idx = 0;
#pragma unroll
    for (int i = BITS - 1; i >= 0; i--) idx = gl::add(gl::add(idx, idx), bit[i]);
    out[BITS] = gl::sub(idx, cw[0]);
#pragma unroll
    for (int b = 0; b < BITS; b++) {
#pragma unroll
        for (int i = 0; i < (VEC >> (b + 1)); i++) items[i] = gl::add(items[2 * i], gl::mul(bit[b], gl::sub(items[2 * i + 1], items[2 * i])));
    }
    out[BITS + 1] = gl::sub(items[0], cw[S]);
}

// (2) every gate except PoseidonGate: Constant, PublicInput, BaseSum<2>, Arithmetic, the extension-arithmetic pair and the
// recursion set (Reducing*, RandomAccess, Exponentiation, PoseidonMds, CosetInterpolation). t0 = index of the first gate constraint.
template <int NCH, bool WIDE>
__global__ void __launch_bounds__(256) quotient_gates_kernel(QuotientArgs a, u32 t0, int finalize) {
    const u64 j = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (j >= a.q_n) return;
    quotient_select_proof(a);
    const u64 S = a.lde_n;
    u64 acc[NCH];
#pragma unroll
    for (int c = 0; c < NCH; c++) acc[c] = a.acc[(u64)c * S + j];
    const u64 *consts_base = a.cs + (u64)a.num_selectors * S + j;
    const u64 *ap = a.alpha_pows + t0;
    for (u32 gi = 0; gi < a.num_gates; gi++) {
        const GateDev g = a.gates[gi];
        if (g.num_constraints == 0 || g.type == 4 || g.type == 14) continue;   // the hash gates have kernels of their own
        const u64 f = gate_filter(a, gi, a.cs[(u64)g.selector_index * S + j]);
        gl::Acc192 sum[NCH];
#pragma unroll
        for (int c = 0; c < NCH; c++) sum[c] = gl::acc_zero();
        auto emit = [&](u32 q, u64 cst) {
#pragma unroll
            for (int c = 0; c < NCH; c++) gl::acc_mul(sum[c], cst, ap[(u64)c * a.nterms + q]);
        };
        if (g.type == 1) {            // ConstantGate: const_i - wire_i
            for (u32 q = 0; q < g.param0; q++) emit(q, gl::sub(consts_base[(u64)q * S], a.wires[(u64)q * S + j]));
        } else if (g.type == 2) {     // PublicInputGate: wire_i - pi_hash_i
            for (u32 q = 0; q < 4; q++) emit(q, gl::sub(a.wires[(u64)q * S + j], a.pi_hash[q]));
        } else if (g.type == 3) {     // ArithmeticGate: out - (c0 m0 m1 + c1 addend)
            const u64 c0 = consts_base[0], c1 = consts_base[S];
            for (u32 q = 0; q < g.param0; q++) {
                const u64 m0 = a.wires[(u64)(4 * q) * S + j], m1 = a.wires[(u64)(4 * q + 1) * S + j];
                const u64 ad = a.wires[(u64)(4 * q + 2) * S + j], out = a.wires[(u64)(4 * q + 3) * S + j];
                emit(q, gl::sub(out, gl::add(gl::mul(gl::mul(m0, m1), c0), gl::mul(ad, c1))));
            }
        } else if (g.type == 6) {     // ArithmeticExtensionGate<2>: out - (c0 m0 m1 + c1 addend) over F[x]/(x^2-7)
            const u64 c0 = consts_base[0], c1 = consts_base[S];
            for (u32 q = 0; q < g.param0; q++) {
                const u64 *w = a.wires + (u64)(8 * q) * S + j;
                const e2 m0 = gl::e2_make(w[0], w[S]), m1 = gl::e2_make(w[2 * S], w[3 * S]), ad = gl::e2_make(w[4 * S], w[5 * S]);
                const e2 out = gl::e2_make(w[6 * S], w[7 * S]);
                const e2 d = gl::e2_sub(out, gl::e2_add(gl::e2_scale(gl::e2_mul(m0, m1), c0), gl::e2_scale(ad, c1)));
                emit(2 * q, d.a); emit(2 * q + 1, d.b);
            }
        } else if (g.type == 7) {     // MulExtensionGate<2>: out - c0 m0 m1
            const u64 c0 = consts_base[0];
            for (u32 q = 0; q < g.param0; q++) {
                const u64 *w = a.wires + (u64)(6 * q) * S + j;
                const e2 d = gl::e2_sub(gl::e2_make(w[4 * S], w[5 * S]),
                                        gl::e2_scale(gl::e2_mul(gl::e2_make(w[0], w[S]), gl::e2_make(w[2 * S], w[3 * S])), c0));
                emit(2 * q, d.a); emit(2 * q + 1, d.b);
            }
        } else if (g.type == 8 || g.type == 9) {   // ReducingGate / ReducingExtensionGate: acc*alpha + coeff_i - acc_i, chained
            const bool ext = g.type == 9;
            const u32 nc = g.param0, start_accs = 6 + (ext ? 2 * nc : nc);
            const u64 *w = a.wires + j;
            const e2 alpha = gl::e2_make(w[2 * S], w[3 * S]);
            e2 acc = gl::e2_make(w[4 * S], w[5 * S]);
            for (u32 q = 0; q < nc; q++) {
                const u32 nx = q == nc - 1 ? 0 : start_accs + 2 * q;
                const e2 next = gl::e2_make(w[(u64)nx * S], w[(u64)(nx + 1) * S]);
                e2 t = gl::e2_mul(acc, alpha);
                if (ext) t = gl::e2_add(t, gl::e2_make(w[(u64)(6 + 2 * q) * S], w[(u64)(7 + 2 * q) * S]));
                else t.a = gl::add(t.a, w[(u64)(6 + q) * S]);
                emit(2 * q, gl::sub(t.a, next.a)); emit(2 * q + 1, gl::sub(t.b, next.b));
                acc = next;
            }
        } else if (g.type == 10) {    // RandomAccessGate(bits, copies, extra constants)
            const u32 bits = g.param0, copies = g.param1, extra = g.param2, vec = 1u << bits;
            const u32 routed = (2 + vec) * copies + extra;
            u32 q = 0;
            for (u32 cp = 0; cp < copies; cp++) {
                const u64 *cw = a.wires + (u64)((2 + vec) * cp) * S + j, *bw = a.wires + (u64)(routed + cp * bits) * S + j;
                u64 vals[7];
                switch (bits) {
                    case 1: random_access_values<1>(cw, bw, S, vals); break;
                    case 2: random_access_values<2>(cw, bw, S, vals); break;
                    case 3: random_access_values<3>(cw, bw, S, vals); break;
                    case 4: random_access_values<4>(cw, bw, S, vals); break;
                    default: if constexpr (WIDE) random_access_values<5>(cw, bw, S, vals); break;
                }
                for (u32 i = 0; i < bits + 2; i++) emit(q++, vals[i]);
            }
            for (u32 i = 0; i < extra; i++) emit(q++, gl::sub(consts_base[(u64)i * S], a.wires[(u64)((2 + vec) * copies + i) * S + j]));
        } else if (g.type == 11) {    // ExponentiationGate: square-and-multiply chain over the power bits (big-endian walk)
            const u32 n = g.param0;
            const u64 *w = a.wires + j;
            const u64 base = w[0];
            for (u32 q = 0; q < n; q++) {
                const u64 pi = q == 0 ? 1 : w[(u64)(2 + n + q - 1) * S];
                const u64 prev = q == 0 ? 1 : gl::mul(pi, pi);
                const u64 bit = w[(u64)(1 + (n - 1 - q)) * S];
                const u64 computed = gl::mul(prev, gl::add(gl::mul(bit, base), gl::sub(1, bit)));
                emit(q, gl::sub(computed, w[(u64)(2 + n + q) * S]));
            }
            emit(n, gl::sub(w[(u64)(1 + n) * S], w[(u64)(2 + n + n - 1) * S]));
        } else if (g.type == 12) {    // PoseidonMdsGate: out - MDS(in) on 12 extension-algebra elements, component by component
            const u64 *w = a.wires + j;
            for (u32 comp = 0; comp < 2; comp++) {
                u64 st[12];
#pragma unroll
                for (int i = 0; i < 12; i++) st[i] = w[(u64)(2 * i + comp) * S];
                poseidon::mds_layer(st);
#pragma unroll
                for (int i = 0; i < 12; i++) emit(2 * i + comp, gl::sub(w[(u64)(24 + 2 * i + comp) * S], st[i]));
            }
        } else if (g.type == 13) {    // CosetInterpolationGate(subgroup_bits, degree): chunked barycentric interpolation
            const u32 bits = g.param0, deg = g.param1, np = 1u << bits, ni = (np - 2) / (deg - 1);
            const u32 s_ep = 1 + 2 * np, s_ev = s_ep + 2, s_int = s_ev + 2;
            const u64 *w = a.wires + j;
            auto ld = [&](u32 c) { return gl::e2_make(w[(u64)c * S], w[(u64)(c + 1) * S]); };
            const u64 shift = w[0];
            const e2 ep = ld(s_ep), sp = ld(s_int + 4 * ni);
            emit(0, gl::sub(ep.a, gl::mul(sp.a, shift))); emit(1, gl::sub(ep.b, gl::mul(sp.b, shift)));
            // subgroup of order 2^bits: generator 2^(192 >> bits); barycentric weight of x_i is x_i / 2^bits, and
            // 1 / 2^bits = -2^(96 - bits) = p - (2^(64-bits) - 2^(32-bits))
            const u64 omega = 1ull << (192u >> bits), inv_n = gl::P - ((1ull << (64 - bits)) - (1ull << (32 - bits)));
            e2 ev = gl::e2_from(0), pr = gl::e2_from(1);
            u64 x = 1;
            u32 lo = 0, hi = deg, q_out = 2;
            for (u32 c = 0; c <= ni; c++) {
                for (u32 q = lo; q < hi; q++) {
                    e2 term = sp; term.a = gl::sub(term.a, x);
                    const e2 t = gl::e2_scale(gl::e2_mul(ld(1 + 2 * q), pr), gl::mul(x, inv_n));
                    ev = gl::e2_add(gl::e2_mul(ev, term), t);
                    pr = gl::e2_mul(pr, term);
                    x = gl::mul(x, omega);
                }
                if (c == ni) break;
                const e2 ie = ld(s_int + 2 * c), ip = ld(s_int + 2 * (ni + c));
                emit(q_out++, gl::sub(ie.a, ev.a)); emit(q_out++, gl::sub(ie.b, ev.b));
                emit(q_out++, gl::sub(ip.a, pr.a)); emit(q_out++, gl::sub(ip.b, pr.b));
                ev = ie; pr = ip;
                lo = 1 + (deg - 1) * (c + 1); hi = lo + deg - 1 < np ? lo + deg - 1 : np;
            }
            const e2 val = ld(s_ev);
            emit(q_out++, gl::sub(val.a, ev.a)); emit(q_out++, gl::sub(val.b, ev.b));
        } else if (g.type == 5) {     // BaseSumGate<2>: sum - sum_i 2^i limb_i, and limb_i (limb_i - 1)
            u64 s2 = 0;
            for (u32 q = g.param0; q-- > 0;) s2 = gl::add(gl::add(s2, s2), a.wires[(u64)(1 + q) * S + j]);
            emit(0, gl::sub(s2, a.wires[j]));
            for (u32 q = 0; q < g.param0; q++) {
                const u64 limb = a.wires[(u64)(1 + q) * S + j];
                emit(1 + q, gl::mul(limb, gl::sub(limb, 1)));
            }
        }
#pragma unroll
        for (int c = 0; c < NCH; c++) acc[c] = gl::add(acc[c], gl::mul(f, gl::acc_reduce(sum[c])));
    }
    if (finalize) {
        const u64 i = brev32((u32)j, a.log_lde);
        const u64 zi = a.zh_inv[i & (a.rate - 1)];
#pragma unroll
        for (int c = 0; c < NCH; c++) a.out[(u64)c * a.q_n + (i >> a.q_shift)] = gl::canon(gl::mul(acc[c], zi));
    } else {
#pragma unroll
        for (int c = 0; c < NCH; c++) a.acc[(u64)c * S + j] = acc[c];
    }
}

// (3) PoseidonGate (plonky2::gates::poseidon) at one point: wires 0..11 input, 12..23 output, 24 swap, 25..28 delta,
// 29..64 / 65..86 / 87..134 S-box inputs of the full / partial / full rounds; 123 constraints, each weighted by
// alpha_c^(t0+q) on the fly.
template <int NCH>
__global__ void __launch_bounds__(256) quotient_poseidon_kernel(QuotientArgs a, u32 gi, u32 t0, int finalize) {
    const u64 j = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (j >= a.q_n) return;
    quotient_select_proof(a);
    const u64 S = a.lde_n;
    const u64 *ap = a.alpha_pows + t0;
    auto W = [&](u32 i) -> u64 { return a.wires[(u64)i * S + j]; };
    gl::Acc192 wsum[NCH];
#pragma unroll
    for (int c = 0; c < NCH; c++) wsum[c] = gl::acc_zero();
    u32 q = 0;
    auto emit = [&](u64 cst) {
#pragma unroll
        for (int c = 0; c < NCH; c++) gl::acc_mul(wsum[c], cst, ap[(u64)c * a.nterms + q]);
        q++;
    };
    const u64 swap = W(24);
    emit(gl::mul(swap, gl::sub(swap, 1)));
    u64 st[12];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const u64 lhs = W(i), rhs = W(i + 4), delta = W(25 + i);
        emit(gl::sub(gl::mul(swap, gl::sub(rhs, lhs)), delta));
        st[i] = gl::add(lhs, delta); st[i + 4] = gl::sub(rhs, delta);
    }
#pragma unroll
    for (int i = 8; i < 12; i++) st[i] = W(i);
    int rc = 0;
    for (int r = 0; r < 4; r++, rc++) {
#pragma unroll
        for (int i = 0; i < 12; i++) st[i] = gl::add(st[i], a.poseidon_rc[rc * 12 + i]);
        if (r) {
#pragma unroll
            for (int i = 0; i < 12; i++) { const u64 in = W(29 + 12 * (r - 1) + i); emit(gl::sub(st[i], in)); st[i] = in; }
        }
        poseidon::sbox7_layer(st);
        poseidon::mds_layer(st);
    }
    // partial rounds in the textbook schedule: the value fed to the S-box is the same in upstream's fast-partial basis (that
    // change of basis leaves lane 0 alone), and the multiplication-free MDS layer is cheaper here than the fast basis's 23 full
    // products per round plus its 11x11 entry matrix
    for (int r = 0; r < 22; r++, rc++) {
#pragma unroll
        for (int i = 0; i < 12; i++) st[i] = gl::add(st[i], a.poseidon_rc[rc * 12 + i]);
        const u64 in = W(65 + r);
        emit(gl::sub(st[0], in));
        st[0] = poseidon::sbox7_lane(in);
        poseidon::mds_layer(st);
    }
    for (int r = 0; r < 4; r++, rc++) {
#pragma unroll
        for (int i = 0; i < 12; i++) st[i] = gl::add(st[i], a.poseidon_rc[rc * 12 + i]);
#pragma unroll
        for (int i = 0; i < 12; i++) { const u64 in = W(87 + 12 * r + i); emit(gl::sub(st[i], in)); st[i] = in; }
        poseidon::sbox7_layer(st);
        poseidon::mds_layer(st);
    }
#pragma unroll
    for (int i = 0; i < 12; i++) emit(gl::sub(st[i], W(12 + i)));
    const u64 f = gate_filter(a, gi, a.cs[(u64)a.gates[gi].selector_index * S + j]);
    u64 sum[NCH];
#pragma unroll
    for (int c = 0; c < NCH; c++) sum[c] = gl::acc_reduce(wsum[c]);
    if (finalize) {
        const u64 i = brev32((u32)j, a.log_lde);
        const u64 zi = a.zh_inv[i & (a.rate - 1)];
#pragma unroll
        for (int c = 0; c < NCH; c++) a.out[(u64)c * a.q_n + (i >> a.q_shift)] = gl::canon(gl::mul(gl::add(a.acc[(u64)c * S + j], gl::mul(f, sum[c])), zi));
    } else {
#pragma unroll
        for (int c = 0; c < NCH; c++) a.acc[(u64)c * S + j] = gl::add(a.acc[(u64)c * S + j], gl::mul(f, sum[c]));
    }
}

// (4) the qp fork's Poseidon2 gate (type 14; the gate behind `hash_n_to_hash_no_pad_p2`, reference call sites
// wormhole/circuit/src/zk_merkle_proof.rs:482,504,606, nullifier.rs:298-299, unspendable_account.rs:229-231,
// block_header/mod.rs:66) at one point. The permutation is qp-poseidon-core's Poseidon2 (pinned by the reference's seven
// known-answer vectors); the wire layout comes from the pack (P2GateLayout, default = upstream PoseidonGate's layout carried
// over: LAYOUT UNPINNED). Structure: optional swap of the first two 4-element groups (boolean + 4 delta constraints), the
// initial external layer, then per round `add constants -> S-box input equals its wire -> S-box -> linear layer` with the
// S-box inputs of full rounds [first_round_wires ? 0 : 1 .. 3], the 22 partial rounds (lane 0) and the last four full rounds on
// wires, and the 12 outputs: 123 constraints of degree 7 with the default layout. The external layers use the multiplication-free
// form of qp-poseidon-core's block circ(2, 3, 1, 1) (poseidon2::ext_layer_qp).
template <int NCH>
__global__ void __launch_bounds__(256) quotient_poseidon2_kernel(QuotientArgs a, u32 gi, u32 t0, int finalize) {
    const u64 j = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (j >= a.q_n) return;
    quotient_select_proof(a);
    const u64 S = a.lde_n;
    const u64 *ap = a.alpha_pows + t0;
    const P2GateLayout &lay = a.p2_layout;
    const poseidon2::Params &P2 = *a.p2_gate;
    auto W = [&](u32 i) -> u64 { return a.wires[(u64)i * S + j]; };
    gl::Acc192 wsum[NCH];
#pragma unroll
    for (int c = 0; c < NCH; c++) wsum[c] = gl::acc_zero();
    u32 q = 0;
    auto emit = [&](u64 cst) {
#pragma unroll
        for (int c = 0; c < NCH; c++) gl::acc_mul(wsum[c], cst, ap[(u64)c * a.nterms + q]);
        q++;
    };
    u64 st[12];
    if (lay.has_swap()) {
        const u64 swap = W(lay.w_swap);
        emit(gl::mul(swap, gl::sub(swap, 1)));
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const u64 lhs = W(lay.w_input + i), rhs = W(lay.w_input + 4 + i), delta = W(lay.w_delta + i);
            emit(gl::sub(gl::mul(swap, gl::sub(rhs, lhs)), delta));
            st[i] = gl::add(lhs, delta); st[i + 4] = gl::sub(rhs, delta);
        }
    } else {
#pragma unroll
        for (int i = 0; i < 8; i++) st[i] = W(lay.w_input + i);
    }
#pragma unroll
    for (int i = 8; i < 12; i++) st[i] = W(lay.w_input + i);
    poseidon2::ext_layer_qp(st);
    u32 wf = lay.w_full0;
    for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int i = 0; i < 12; i++) st[i] = gl::add(st[i], P2.rc_ext[r * 12 + i]);
        if (r || lay.first_round_wires) {
#pragma unroll
            for (int i = 0; i < 12; i++) { const u64 in = W(wf + i); emit(gl::sub(st[i], in)); st[i] = in; }
            wf += 12;
        }
        poseidon::sbox7_layer(st);
        poseidon2::ext_layer_qp(st);
    }
    for (int r = 0; r < 22; r++) {
        const u64 in = W(lay.w_partial + r);
        emit(gl::sub(gl::add(st[0], P2.rc_int[r]), in));
        st[0] = poseidon::sbox7_lane(in);
        poseidon2::int_layer(st, P2);
    }
    for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int i = 0; i < 12; i++) st[i] = gl::add(st[i], P2.rc_ext[(4 + r) * 12 + i]);
#pragma unroll
        for (int i = 0; i < 12; i++) { const u64 in = W(lay.w_full1 + 12 * r + i); emit(gl::sub(st[i], in)); st[i] = in; }
        poseidon::sbox7_layer(st);
        poseidon2::ext_layer_qp(st);
    }
#pragma unroll
    for (int i = 0; i < 12; i++) emit(gl::sub(st[i], W(lay.w_output + i)));
    const u64 f = gate_filter(a, gi, a.cs[(u64)a.gates[gi].selector_index * S + j]);
    u64 sum[NCH];
#pragma unroll
    for (int c = 0; c < NCH; c++) sum[c] = gl::acc_reduce(wsum[c]);
    if (finalize) {
        const u64 i = brev32((u32)j, a.log_lde);
        const u64 zi = a.zh_inv[i & (a.rate - 1)];
#pragma unroll
        for (int c = 0; c < NCH; c++) a.out[(u64)c * a.q_n + (i >> a.q_shift)] = gl::canon(gl::mul(gl::add(a.acc[(u64)c * S + j], gl::mul(f, sum[c])), zi));
    } else {
#pragma unroll
        for (int c = 0; c < NCH; c++) a.acc[(u64)c * S + j] = gl::add(a.acc[(u64)c * S + j], gl::mul(f, sum[c]));
    }
}

// Witness check on the trace rows (optional): acc holds the alpha-weighted gate-constraint sums per row (the gate kernels
// run on the value arrays, S = n). result[0] = smallest row with a non-zero sum, result[1] = 1 when a permutation
// product does not close (Z(g x_{n-1}) != 1), i.e. a copy constraint is violated.
__global__ void __launch_bounds__(256) witness_check_kernel(const u64 *acc, u64 n, u32 nch, const u64 *z, const u64 *rowprod, u64 *result) {
    const u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (i >= n) return;
    { const u64 pr = blockIdx.z; acc += pr * nch * n; z += pr * nch * n; rowprod += pr * nch * n; result += pr * 2; }
    bool bad = false;
    for (u32 c = 0; c < nch; c++) bad |= gl::canon(acc[(u64)c * n + i]) != 0;
    if (bad) atomicMin((unsigned long long *)&result[0], (unsigned long long)i);
    if (i == n - 1)
        for (u32 c = 0; c < nch; c++)
            if (gl::canon(gl::mul(z[(u64)c * n + i], rowprod[(u64)c * n + i])) != 1) atomicMax((unsigned long long *)&result[1], 1ull);
}

// out[i] = in[i] * shift_inv^i (coset_ifft tail), two-level power table
__global__ void __launch_bounds__(256) scale_powers_kernel(u64 *data, u64 n, u64 ncols, const u64 *pw_lo, const u64 *pw_hi, u32 lo_bits) {
    const u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u64 s = gl::mul(pw_hi[i >> lo_bits], pw_lo[i & ((1u << lo_bits) - 1)]);
    for (u64 c = 0; c < ncols; c++) data[c * n + i] = gl::canon(gl::mul(data[c * n + i], s));
}

// ---------------------------------------------------------------- s7
// Evaluate polynomial p (n base-field coefficients) at an extension point. One workgroup per (poly, point).
__global__ void __launch_bounds__(256) poly_eval_kernel(const u64 *coeffs, u64 n, const e2 *points, e2 *out, u64 ps_coeffs, u64 ps_points, u64 ps_out) {
    __shared__ e2 part[256];
    const u32 t = threadIdx.x, T = blockDim.x;
    { const u64 pr = blockIdx.z; coeffs += pr * ps_coeffs; points += pr * ps_points; out += pr * ps_out; }
    const u64 p = blockIdx.x;
    const e2 z = points[blockIdx.y];
    const u64 *f = coeffs + p * n;
    // thread t takes the coefficients t, t + T, t + 2T, ... (unit stride across the wave): Horner in w = z^T, then times z^t.
    // z^(2^b) on the way to w serve as the factors of z^t.
    e2 zp[8];                                   // T = 256 = 2^8
    zp[0] = z;
#pragma unroll
    for (int b = 1; b < 8; b++) zp[b] = gl::e2_mul(zp[b - 1], zp[b - 1]);
    const e2 w = gl::e2_mul(zp[7], zp[7]);
    e2 acc = gl::e2_from(0);
    if (t < n) {
        const u64 terms = (n - t + T - 1) / T;
        for (u64 k = terms; k-- > 0;) acc = gl::e2_add(gl::e2_mul(acc, w), gl::e2_from(f[t + k * T]));
        e2 zt = gl::e2_from(1);
#pragma unroll
        for (int b = 0; b < 8; b++) if ((t >> b) & 1) zt = gl::e2_mul(zt, zp[b]);
        acc = gl::e2_mul(acc, zt);
    }
    part[t] = acc;
    __syncthreads();
    for (u32 off = T >> 1; off > 0; off >>= 1) {
        if (t < off) part[t] = gl::e2_add(part[t], part[t + off]);
        __syncthreads();
    }
    if (t == 0) out[(u64)blockIdx.y * gridDim.x + blockIdx.x] = gl::e2_canon(part[0]);
}

// ---------------------------------------------------------------- s8
// comp[i] = sum_p alpha^p f_p[i] over the listed polynomials (several source batches)
__global__ void __launch_bounds__(256) reduce_polys_kernel(const ReduceArgs a) {
    const u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const u64 pr = blockIdx.z;
    // the argument block stays read-only: its arrays are indexed by the loop variable, and a modified copy would live in scratch
    const e2 *alpha_pows = a.alpha_pows + pr * a.ps_alpha;
    e2 acc = gl::e2_from(0);
    u32 p = 0;
    for (u32 s = 0; s < a.nsrc; s++) {
        const u64 *base = a.src[s] + pr * a.ps_src[s];
        for (u32 c = 0; c < a.ncols[s]; c++, p++) acc = gl::e2_add(acc, gl::e2_scale(alpha_pows[p], base[(u64)c * a.n + i]));
    }
    a.comp_a[pr * a.ps_comp + i] = gl::canon(acc.a);
    a.comp_b[pr * a.ps_comp + i] = gl::canon(acc.b);
}

// q = comp / (X - z) by synthetic division: b_{i-1} = b_i z + comp_i (from the top), quotient coefficient
// q_{i-1} = b_i ... one workgroup, blocked linear-recurrence scan. final += handled by the caller's mode:
// mode 0: final = q ; mode 1: final = final * shift + q.
__global__ void __launch_bounds__(1024) divide_linear_kernel(const u64 *comp_a, const u64 *comp_b, u64 n, const e2 *zs, const e2 *shifts, int mode, u64 *fin_a, u64 *fin_b,
                                                             u64 ps_comp, u64 ps_fin) {
    __shared__ e2 carry[1024];
    const u32 t = threadIdx.x, T = blockDim.x;
    const u64 pr = blockIdx.x;   // one workgroup per proof of the batch
    comp_a += pr * ps_comp; comp_b += pr * ps_comp; fin_a += pr * ps_fin; fin_b += pr * ps_fin;
    const e2 z = zs[pr], shift = shifts[pr];
    const u64 per = (n + T - 1) / T;
    // thread t owns indices [lo, hi) counted from the TOP: index i = n-1-r
    const u64 rlo = (u64)t * per, rhi = rlo + per < n ? rlo + per : n;
    // local Horner with zero incoming carry
    e2 acc = gl::e2_from(0);
    for (u64 r = rlo; r < rhi; r++) { const u64 i = n - 1 - r; acc = gl::e2_add(gl::e2_mul(acc, z), gl::e2_make(comp_a[i], comp_b[i])); }
    const e2 zper = gl::e2_pow(z, rhi > rlo ? rhi - rlo : 0);
    // sequential-in-log combine: carry_in(t) = value of the recurrence after all elements owned by threads < t
    carry[t] = acc;
    __syncthreads();
    // Hillis-Steele over the affine maps (acc, zper): combined(t) = carry[t - off] * zpow[t] + carry[t]
    __shared__ e2 zp[1024];
    zp[t] = zper;
    __syncthreads();
    for (u32 off = 1; off < T; off <<= 1) {
        e2 c = carry[t], m = zp[t];
        if (t >= off) { c = gl::e2_add(gl::e2_mul(carry[t - off], zp[t]), carry[t]); m = gl::e2_mul(zp[t - off], zp[t]); }
        __syncthreads();
        carry[t] = c; zp[t] = m;
        __syncthreads();
    }
    e2 cin = t == 0 ? gl::e2_from(0) : carry[t - 1];
    // replay with the true incoming value: b after consuming index i is the quotient coefficient q_{i-1}
    acc = cin;
    for (u64 r = rlo; r < rhi; r++) {
        const u64 i = n - 1 - r;
        acc = gl::e2_add(gl::e2_mul(acc, z), gl::e2_make(comp_a[i], comp_b[i]));
        if (i > 0) {
            e2 q = acc;
            if (mode) q = gl::e2_add(gl::e2_mul(gl::e2_make(fin_a[i - 1], fin_b[i - 1]), shift), q);
            fin_a[i - 1] = gl::canon(q.a); fin_b[i - 1] = gl::canon(q.b);
        }
    }
    if (t == 0) {   // the padded top coefficient q_{n-1} = 0
        e2 q = gl::e2_from(0);
        if (mode) q = gl::e2_mul(gl::e2_make(fin_a[n - 1], fin_b[n - 1]), shift);
        fin_a[n - 1] = gl::canon(q.a); fin_b[n - 1] = gl::canon(q.b);
    }
}

// ---------------------------------------------------------------- s9
// rows[j] = (va[j], vb[j]) interleaved: extension values in leaf order -> row-major leaves of 2*arity felts
__global__ void __launch_bounds__(256) interleave_ext_kernel(const u64 *va, const u64 *vb, u64 n, u64 *rows, u64 ps_vals, u64 ps_rows) {
    const u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (i >= n) return;
    { const u64 pr = blockIdx.z; va += pr * ps_vals; vb += pr * ps_vals; rows += pr * ps_rows; }
    reinterpret_cast<ulonglong2 *>(rows)[i] = make_ulonglong2(va[i], vb[i]);
}
// new[i] = sum_{k < arity} beta^k coeffs[arity*i + k]   (in place is safe: i <= arity*i; done out of place here)
__global__ void __launch_bounds__(256) fri_fold_kernel(const u64 *ca, const u64 *cb, u64 new_n, u32 arity, const e2 *betas, u64 *oa, u64 *ob, u64 ps_in, u64 ps_out) {
    const u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (i >= new_n) return;
    const u64 pr = blockIdx.z;
    ca += pr * ps_in; cb += pr * ps_in; oa += pr * ps_out; ob += pr * ps_out;
    const e2 beta = betas[pr];
    e2 acc = gl::e2_from(0);
    for (u32 k = arity; k-- > 0;) acc = gl::e2_add(gl::e2_mul(acc, beta), gl::e2_make(ca[(u64)arity * i + k], cb[(u64)arity * i + k]));
    oa[i] = gl::canon(acc.a); ob[i] = gl::canon(acc.b);
}

// ---------------------------------------------------------------- s11
// out[q][c] = cols[c*stride + idx[q]]
__global__ void gather_rows_kernel(const u64 *cols, u64 stride, u32 ncols, const u64 *idx, u32 nq, u64 *out, u64 ps_cols, u64 ps_out) {
    const u32 q = blockIdx.x;
    { const u64 pr = blockIdx.z; cols += pr * ps_cols; idx += pr * nq; out += pr * ps_out; }
    for (u32 c = threadIdx.x; c < ncols; c += blockDim.x) out[(u64)q * ncols + c] = cols[(u64)c * stride + idx[q]];
}
// Merkle authentication paths: out[q][lvl] = digests[level lvl][ (idx[q] >> lvl) ^ 1 ]
__global__ void gather_paths_kernel(const u64 *digests, u64 n_leaves, u32 path_len, const u64 *idx, u32 shift, u64 *out, u32 nq, u64 ps_digests, u64 ps_out) {
    const u32 q = blockIdx.x, t = threadIdx.x;
    { const u64 pr = blockIdx.z; digests += pr * ps_digests; idx += pr * nq; out += pr * ps_out; }
    if (t >= path_len * 4) return;
    const u32 lvl = t >> 2, e = t & 3;
    u64 off = 0, cnt = n_leaves;
    for (u32 l = 0; l < lvl; l++) { off += cnt; cnt >>= 1; }
    const u64 node = ((idx[q] >> shift) >> lvl) ^ 1;
    out[((u64)q * path_len + lvl) * 4 + e] = digests[(off + node) * 4 + e];
}
// out[q][e] = rows[(idx[q] >> shift) * width + e]
__global__ void gather_leaf_rows_kernel(const u64 *rows, u32 width, const u64 *idx, u32 shift, u64 *out, u32 nq, u64 ps_rows, u64 ps_out) {
    const u32 q = blockIdx.x;
    { const u64 pr = blockIdx.z; rows += pr * ps_rows; idx += pr * nq; out += pr * ps_out; }
    for (u32 e = threadIdx.x; e < width; e += blockDim.x) out[(u64)q * width + e] = rows[(idx[q] >> shift) * width + e];
}

// salt columns of a blinded oracle (PolynomialBatch::from_coeffs with blinding: SALT_SIZE = 4 random columns in every
// leaf). The reference draws them from thread_rng, a CSPRNG; every opened leaf publishes its salts, so they must not be
// predictable from one another: ChaCha20 (RFC 8439 block function, 64-bit block counter) keyed with 256 bits per proof
// from the OS entropy source (or from an injected seed, which makes a proof reproducible: SURVEY.md section 0.5).
// Stream layout: nonce = (oracle_index, column); block (leaf >> 1) holds the candidates of leaves 2k and 2k+1, four 64-bit
// words each; a leaf takes its first candidate below p (rejection sampling: uniform on [0, p)).
__device__ __forceinline__ u32 rotl32(u32 x, int k) { return (x << k) | (x >> (32 - k)); }
__device__ __forceinline__ void chacha20_block(const u32 *key, u64 counter, u32 n0, u32 n1, u32 (&out)[16]) {
    u32 x[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key[0], key[1], key[2], key[3], key[4], key[5], key[6], key[7],
                 (u32)counter, (u32)(counter >> 32), n0, n1};
    u32 w[16];
#pragma unroll
    for (int i = 0; i < 16; i++) w[i] = x[i];
#define QR(a, b, c, d) w[a] += w[b]; w[d] = rotl32(w[d] ^ w[a], 16); w[c] += w[d]; w[b] = rotl32(w[b] ^ w[c], 12); \
                       w[a] += w[b]; w[d] = rotl32(w[d] ^ w[a], 8);  w[c] += w[d]; w[b] = rotl32(w[b] ^ w[c], 7);
#pragma unroll 1
    for (int r = 0; r < 10; r++) {
        QR(0, 4, 8, 12) QR(1, 5, 9, 13) QR(2, 6, 10, 14) QR(3, 7, 11, 15)
        QR(0, 5, 10, 15) QR(1, 6, 11, 12) QR(2, 7, 8, 13) QR(3, 4, 9, 14)
    }
#undef QR
#pragma unroll
    for (int i = 0; i < 16; i++) out[i] = w[i] + x[i];
}
__global__ void __launch_bounds__(256) salt_kernel(const u32 *keys, u32 oracle_index, u64 lde_n, u64 *out) {
    const u64 j = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (j >= lde_n) return;
    const u32 *key = keys + 8 * blockIdx.z;
    out += (u64)blockIdx.z * 4 * lde_n;
    for (u32 c = 0; c < 4; c++) {
        u32 blk[16];
        chacha20_block(key, j >> 1, oracle_index, c, blk);
        const u32 h = (u32)(j & 1) * 8;
        u64 v = 0;
        bool found = false;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const u64 cand = ((u64)blk[h + 2 * k + 1] << 32) | blk[h + 2 * k];
            if (!found && cand < gl::P) { v = cand; found = true; }
        }
        if (!found) v = (((u64)blk[h + 7] << 32) | blk[h + 6]) - gl::P;   // four rejections in a row: probability 2^-128
        out[(u64)c * lde_n + j] = v;
    }
}

// RandomValueGenerator on the device (CircuitBuilder::blind's random wires of a zero-knowledge circuit): value j of witness b from
// the ChaCha20 stream of that witness's key, nonce ("BLND", 0); block (j >> 1) holds the candidates of values 2k and 2k+1, the
// first of four below p is taken (uniform on [0, p)). out: [batch][pitch], `count` values each.
__global__ void __launch_bounds__(256) random_felts_kernel(const u32 *keys, u64 count, u64 *out, u64 pitch) {
    const u64 j = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (j >= count) return;
    u32 blk[16];
    chacha20_block(keys + 8 * blockIdx.z, j >> 1, 0x444E4C42u, 0, blk);
    const u32 h = (u32)(j & 1) * 8;
    u64 v = 0;
    bool found = false;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const u64 cand = ((u64)blk[h + 2 * k + 1] << 32) | blk[h + 2 * k];
        if (!found && cand < gl::P) { v = cand; found = true; }
    }
    if (!found) v = (((u64)blk[h + 7] << 32) | blk[h + 6]) - gl::P;
    out[(u64)blockIdx.z * pitch + j] = v;
}

// x_coset[j] = g * w^bitrev(j), l0_coset[j] = zh(i) / (n (x - 1))
__global__ void __launch_bounds__(256) coset_tables_kernel(u64 lde_n, u32 log_lde, const u64 *pw_lo, const u64 *pw_hi, u32 lo_bits,
                                                             const u64 *zh, u32 rate, u64 n_field, u64 *x_coset, u64 *l0_coset) {
    const u64 j = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (j >= lde_n) return;
    const u64 i = brev32((u32)j, log_lde);
    const u64 x = gl::canon(gl::mul(gl::MULT_GEN, gl::mul(pw_hi[i >> lo_bits], pw_lo[i & ((1u << lo_bits) - 1)])));
    x_coset[j] = x;
    l0_coset[j] = gl::canon(gl::mul(zh[i & (rate - 1)], gl::inv(gl::mul(n_field, gl::sub(x, 1)))));
}

}  // namespace

// rows of `width` words, `pitch` words apart -> one dense block (a lockstep batch's per-proof results, so that one
// contiguous copy brings them to the host; a 2D copy is issued by the runtime as one small copy per row)
__global__ void __launch_bounds__(256) pack_rows_kernel(const u64 *src, u64 pitch, u64 width, u64 total, u64 *dst) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const u64 r = i / width, c = i - r * width;
    dst[i] = src[r * pitch + c];
}

// the inverse: a dense block of `total / width` rows -> rows `pitch` words apart (one upload for the per-proof tables of a lockstep
// batch instead of one copy launch per proof)
__global__ void __launch_bounds__(256) unpack_rows_kernel(const u64 *src, u64 pitch, u64 width, u64 total, u64 *dst) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const u64 r = i / width, c = i - r * width;
    dst[r * pitch + c] = src[i];
}

// dst[0..bytes) = src[0..bytes): the small transfers of the proving path between pinned host memory and device memory (either
// side may be the pinned one: the device reads and writes it in place). 8-byte words when both ends are aligned, bytes otherwise.
__global__ void __launch_bounds__(256) copy_words_kernel(u64 *dst, const u64 *src, u64 words, u64 tail_bytes) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < words) dst[i] = src[i];
    if (i == 0) for (u64 k = 0; k < tail_bytes; k++) ((unsigned char *)(dst + words))[k] = ((const unsigned char *)(src + words))[k];
}
__global__ void __launch_bounds__(256) copy_bytes_kernel(unsigned char *dst, const unsigned char *src, u64 bytes) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < bytes) dst[i] = src[i];
}

#define LAUNCH_1D(kern, count, threads, st, ...) \
    do { if ((count) > 0) { dim3 b(threads), g((unsigned)(((count) + (threads)-1) / (threads))); hipLaunchKernelGGL(kern, g, b, 0, st, __VA_ARGS__); } } while (0)
// the same over a lockstep batch: grid.z = proof
#define LAUNCH_1D_B(kern, count, threads, nbatch, st, ...) \
    do { if ((count) > 0 && (nbatch) > 0) { dim3 b(threads), g((unsigned)(((count) + (threads)-1) / (threads)), 1, (unsigned)(nbatch)); hipLaunchKernelGGL(kern, g, b, 0, st, __VA_ARGS__); } } while (0)

hipError_t pk_pp_rows(const PpArgs &a, hipStream_t st) {
    dim3 b(256), g((unsigned)((a.n + 255) / 256), a.nch, a.batch);
    hipLaunchKernelGGL(pp_rows_kernel, g, b, 0, st, a);
    return hipGetLastError();
}
hipError_t pk_pp_scan(const u64 *rowprod, u64 *z, u64 n, u32 nch_total, hipStream_t st) {
    hipLaunchKernelGGL(pp_scan_kernel, dim3(nch_total), dim3(1024), 0, st, rowprod, z, n);
    return hipGetLastError();
}
hipError_t pk_pp_finish(const PpArgs &a, const u64 *z, u64 *zs_pp, u64 ps_z, u64 ps_zs, hipStream_t st) {
    dim3 b(256), g((unsigned)((a.n + 255) / 256), a.nch, a.batch);
    hipLaunchKernelGGL(pp_finish_kernel, g, b, 0, st, a, z, zs_pp, ps_z, ps_zs);
    return hipGetLastError();
}
static bool wide_random_access(const QuotientArgs &a, const GateDev *host_gates) {
    for (u32 i = 0; i < a.num_gates; i++) if (host_gates[i].type == 10 && host_gates[i].param0 > 4) return true;
    return false;
}
template <int NCH>
static hipError_t quotient_launch(const QuotientArgs &a, const GateDev *host_gates, hipStream_t st) {
    dim3 b(256), g((unsigned)((a.q_n + 255) / 256), 1, a.batch);
    const u32 t0 = a.nch + a.nch * a.nchunks;
    // Poseidon gates (heavy, one launch each) come last; the final launch also applies 1/Z_H and stores
    auto is_hash_gate = [&](u32 i) { return (host_gates[i].type == 4 || host_gates[i].type == 14) && host_gates[i].num_constraints; };
    int n_pos = 0;
    for (u32 i = 0; i < a.num_gates; i++) if (is_hash_gate(i)) n_pos++;
    hipLaunchKernelGGL((quotient_perm_kernel<NCH>), g, b, 0, st, a);
    if (wide_random_access(a, host_gates)) hipLaunchKernelGGL((quotient_gates_kernel<NCH, true>), g, b, 0, st, a, t0, n_pos == 0 ? 1 : 0);
    else hipLaunchKernelGGL((quotient_gates_kernel<NCH, false>), g, b, 0, st, a, t0, n_pos == 0 ? 1 : 0);
    int seen = 0;
    for (u32 i = 0; i < a.num_gates; i++)
        if (is_hash_gate(i)) {
            seen++;
            if (host_gates[i].type == 4) hipLaunchKernelGGL((quotient_poseidon_kernel<NCH>), g, b, 0, st, a, i, t0, seen == n_pos ? 1 : 0);
            else hipLaunchKernelGGL((quotient_poseidon2_kernel<NCH>), g, b, 0, st, a, i, t0, seen == n_pos ? 1 : 0);
        }
    return hipGetLastError();
}
// gate kernels only (no permutation terms, no 1/Z_H): used by the witness check on the trace rows
template <int NCH>
static hipError_t gates_only_launch(const QuotientArgs &a, const GateDev *host_gates, hipStream_t st) {
    dim3 b(256), g((unsigned)((a.q_n + 255) / 256), 1, a.batch);
    const u32 t0 = a.nch + a.nch * a.nchunks;
    if (wide_random_access(a, host_gates)) hipLaunchKernelGGL((quotient_gates_kernel<NCH, true>), g, b, 0, st, a, t0, 0);
    else hipLaunchKernelGGL((quotient_gates_kernel<NCH, false>), g, b, 0, st, a, t0, 0);
    for (u32 i = 0; i < a.num_gates; i++) {
        if (host_gates[i].type == 4 && host_gates[i].num_constraints) hipLaunchKernelGGL((quotient_poseidon_kernel<NCH>), g, b, 0, st, a, i, t0, 0);
        if (host_gates[i].type == 14 && host_gates[i].num_constraints) hipLaunchKernelGGL((quotient_poseidon2_kernel<NCH>), g, b, 0, st, a, i, t0, 0);
    }
    return hipGetLastError();
}
hipError_t pk_gate_sums(const QuotientArgs &a, const GateDev *host_gates, hipStream_t st) {
    switch (a.nch) {
        case 1: return gates_only_launch<1>(a, host_gates, st);
        case 2: return gates_only_launch<2>(a, host_gates, st);
        case 3: return gates_only_launch<3>(a, host_gates, st);
        case 4: return gates_only_launch<4>(a, host_gates, st);
        default: return hipErrorInvalidValue;
    }
}
hipError_t pk_witness_check(const u64 *acc, u64 n, u32 nch, const u64 *z, const u64 *rowprod, u64 *result, u32 batch, hipStream_t st) {
    LAUNCH_1D_B(witness_check_kernel, n, 256, batch, st, acc, n, nch, z, rowprod, result);
    return hipGetLastError();
}
hipError_t pk_quotient(const QuotientArgs &a, const GateDev *host_gates, hipStream_t st) {
    if (a.q_n == 0 || a.batch == 0) return hipSuccess;
    switch (a.nch) {
        case 1: return quotient_launch<1>(a, host_gates, st);
        case 2: return quotient_launch<2>(a, host_gates, st);
        case 3: return quotient_launch<3>(a, host_gates, st);
        case 4: return quotient_launch<4>(a, host_gates, st);
        default: return hipErrorInvalidValue;
    }
}
hipError_t pk_copy(void *dst, const void *src, size_t bytes, hipStream_t st) {
    if (bytes == 0) return hipSuccess;
    if ((((uintptr_t)dst | (uintptr_t)src) & 7) == 0) {
        const u64 words = bytes / 8, tail = bytes % 8;
        LAUNCH_1D(copy_words_kernel, words ? words : 1, 256, st, (u64 *)dst, (const u64 *)src, words, tail);
    } else {
        LAUNCH_1D(copy_bytes_kernel, bytes, 256, st, (unsigned char *)dst, (const unsigned char *)src, (u64)bytes);
    }
    return hipGetLastError();
}
hipError_t pk_pack_rows(const u64 *src, u64 pitch_words, u64 width_words, u64 rows, u64 *dst, hipStream_t st) {
    LAUNCH_1D(pack_rows_kernel, width_words * rows, 256, st, src, pitch_words, width_words, width_words * rows, dst);
    return hipGetLastError();
}
hipError_t pk_unpack_rows(const u64 *src, u64 pitch_words, u64 width_words, u64 rows, u64 *dst, hipStream_t st) {
    if (width_words * rows == 0) return hipSuccess;
    LAUNCH_1D(unpack_rows_kernel, width_words * rows, 256, st, src, pitch_words, width_words, width_words * rows, dst);
    return hipGetLastError();
}
hipError_t pk_scale_powers(u64 *data, u64 n, u64 ncols, const u64 *pw_lo, const u64 *pw_hi, u32 lo_bits, hipStream_t st) {
    LAUNCH_1D(scale_powers_kernel, n, 256, st, data, n, ncols, pw_lo, pw_hi, lo_bits);
    return hipGetLastError();
}
hipError_t pk_poly_eval(const u64 *coeffs, u64 n, u32 npolys, const e2 *points, u32 npoints, e2 *out, u32 batch, u64 ps_coeffs, u64 ps_points, u64 ps_out, hipStream_t st) {
    if (npolys == 0 || batch == 0) return hipSuccess;
    hipLaunchKernelGGL(poly_eval_kernel, dim3(npolys, npoints, batch), dim3(256), 0, st, coeffs, n, points, out, ps_coeffs, ps_points, ps_out);
    return hipGetLastError();
}
hipError_t pk_reduce_polys(const ReduceArgs &a, hipStream_t st) {
    LAUNCH_1D_B(reduce_polys_kernel, a.n, 256, a.batch, st, a);
    return hipGetLastError();
}
hipError_t pk_divide_linear(const u64 *comp_a, const u64 *comp_b, u64 n, const e2 *zs, const e2 *shifts, int mode, u64 *fin_a, u64 *fin_b,
                            u32 batch, u64 ps_comp, u64 ps_fin, hipStream_t st) {
    if (batch == 0) return hipSuccess;
    unsigned threads = n >= 1024 ? 1024 : (n >= 64 ? (unsigned)n : 64);
    hipLaunchKernelGGL(divide_linear_kernel, dim3(batch), dim3(threads), 0, st, comp_a, comp_b, n, zs, shifts, mode, fin_a, fin_b, ps_comp, ps_fin);
    return hipGetLastError();
}
hipError_t pk_interleave_ext(const u64 *va, const u64 *vb, u64 n, u64 *rows, u32 batch, u64 ps_vals, u64 ps_rows, hipStream_t st) {
    LAUNCH_1D_B(interleave_ext_kernel, n, 256, batch, st, va, vb, n, rows, ps_vals, ps_rows);
    return hipGetLastError();
}
hipError_t pk_fri_fold(const u64 *ca, const u64 *cb, u64 new_n, u32 arity, const e2 *betas, u64 *oa, u64 *ob, u32 batch, u64 ps_in, u64 ps_out, hipStream_t st) {
    LAUNCH_1D_B(fri_fold_kernel, new_n, 256, batch, st, ca, cb, new_n, arity, betas, oa, ob, ps_in, ps_out);
    return hipGetLastError();
}
hipError_t pk_gather_rows(const u64 *cols, u64 stride, u32 ncols, const u64 *idx, u32 nq, u64 *out, u32 batch, u64 ps_cols, u64 ps_out, hipStream_t st) {
    if (nq == 0 || batch == 0) return hipSuccess;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(nq, 1, batch), dim3(128), 0, st, cols, stride, ncols, idx, nq, out, ps_cols, ps_out);
    return hipGetLastError();
}
hipError_t pk_gather_paths(const u64 *digests, u64 n_leaves, u32 path_len, const u64 *idx, u32 shift, u32 nq, u64 *out, u32 batch, u64 ps_digests, u64 ps_out, hipStream_t st) {
    if (path_len == 0 || nq == 0 || batch == 0) return hipSuccess;
    hipLaunchKernelGGL(gather_paths_kernel, dim3(nq, 1, batch), dim3(256), 0, st, digests, n_leaves, path_len, idx, shift, out, nq, ps_digests, ps_out);
    return hipGetLastError();
}
hipError_t pk_gather_leaf_rows(const u64 *rows, u32 width, const u64 *idx, u32 shift, u32 nq, u64 *out, u32 batch, u64 ps_rows, u64 ps_out, hipStream_t st) {
    if (nq == 0 || batch == 0) return hipSuccess;
    hipLaunchKernelGGL(gather_leaf_rows_kernel, dim3(nq, 1, batch), dim3(64), 0, st, rows, width, idx, shift, out, nq, ps_rows, ps_out);
    return hipGetLastError();
}
hipError_t pk_salt(const u32 *keys, u32 oracle_index, u64 lde_n, u64 *out, u32 batch, hipStream_t st) {
    LAUNCH_1D_B(salt_kernel, lde_n, 256, batch, st, keys, oracle_index, lde_n, out);
    return hipGetLastError();
}
hipError_t pk_random_felts(const u32 *keys, u64 count, u64 *out, u64 pitch, u32 batch, hipStream_t st) {
    if (count == 0 || batch == 0) return hipSuccess;
    LAUNCH_1D_B(random_felts_kernel, count, 256, batch, st, keys, count, out, pitch);
    return hipGetLastError();
}
hipError_t pk_coset_tables(u64 lde_n, u32 log_lde, const u64 *pw_lo, const u64 *pw_hi, u32 lo_bits, const u64 *zh, u32 rate,
                           u64 n_field, u64 *x_coset, u64 *l0_coset, hipStream_t st) {
    LAUNCH_1D(coset_tables_kernel, lde_n, 256, st, lde_n, log_lde, pw_lo, pw_hi, lo_bits, zh, rate, n_field, x_coset, l0_coset);
    return hipGetLastError();
}
