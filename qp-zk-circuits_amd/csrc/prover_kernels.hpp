// prover_kernels.hpp — argument blocks and launchers of prover_kernels.hip.
//
// Lockstep batches: every launcher takes `batch` proofs of one circuit at once (grid.z or a folded leading dimension =
// proof index). A per-proof buffer X of one proof's size |X| is laid out as [batch][|X|]; the `ps_*` fields are those
// per-proof strides in words (0 = the buffer is shared by all proofs, e.g. the constants/sigmas oracle).
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include "circuit.hpp"
#include "gl64.hpp"

namespace poseidon2 { struct Params; }

struct GateDev { uint32_t type, param0, param1, selector_index, group_start, group_end, num_constraints, param2; };

struct PpArgs {
    const uint64_t *wires;      // [num_wires][n] values, natural order
    const uint64_t *sigmas;     // [num_routed][n] sigma values
    const uint64_t *omega_pows; // [n]
    const uint64_t *beta_k_is;  // [nch][num_routed] beta_k * k_j
    const uint64_t *betas, *gammas;
    uint64_t *qcp;              // [nch][nchunks][n]
    uint64_t *rowprod;          // [nch][n]
    uint64_t n;
    uint32_t num_routed, chunk, nchunks, nch;
    uint32_t batch;
    uint64_t ps_wires, ps_small, ps_qcp, ps_rowprod;   // per-proof strides: wires; betas/gammas/beta_k_is; qcp; rowprod
};

struct QuotientArgs {
    const uint64_t *wires, *cs, *zs_pp;   // LDE, column-major, leaf order, stride lde_n
    const uint64_t *x_coset, *l0_coset;   // [lde_n] slot order
    const uint64_t *zh_inv;               // [rate]
    const uint64_t *alpha_pows;           // [nch][nterms]
    const uint64_t *beta_k_is, *betas, *gammas, *pi_hash;
    const GateDev *gates;
    const uint64_t *poseidon_rc;          // 360 round constants (PoseidonGate)
    const uint64_t *poseidon_fast;        // FAST_PARTIAL_* tables, poseidon::FP_WORDS entries
    const poseidon2::Params *p2_gate;     // constants of the Poseidon2 gate (qp-poseidon-core's set), device block
    P2GateLayout p2_layout;               // its wire layout (circuit.hpp)
    uint64_t *acc;                        // [nch][lde_n] slot order: running alpha-weighted sums between the s6 kernels
    uint64_t *out;                        // [nch][lde_n] natural order
    uint64_t lde_n;                       // column stride of the LDE batches (slots)
    uint64_t q_n;                         // points the quotient is evaluated on: the first q_n slots = the coset g<w_{n*qdf}>
    uint32_t q_shift;                     // rate_bits - log2(quotient_degree_factor): natural LDE index >> q_shift = quotient index
    uint32_t log_lde, rate, nch, num_routed, chunk, nchunks, sig0, num_selectors, num_gates, nterms;
    uint32_t batch;
    uint64_t ps_wires, ps_zs, ps_small, ps_acc, ps_out;   // per-proof strides (alpha_pows, beta_k_is, betas, gammas, pi_hash share ps_small)
};

struct ReduceArgs {
    const uint64_t *src[8];
    uint64_t ps_src[8];
    uint32_t ncols[8];
    uint32_t nsrc;
    const gl::e2 *alpha_pows;
    uint64_t *comp_a, *comp_b;
    uint64_t n;
    uint32_t batch;
    uint64_t ps_alpha, ps_comp;     // alpha_pows in e2 units, comp in words
};

// proof of work for a batch: proof b tries nonces base[b] .. base[b] + count (count 0: already found)
struct PowArgs {
    const uint64_t *states;         // [batch][12] duplex state with the buffered inputs written in
    const uint64_t *bases;          // [batch]
    uint64_t *results;              // [batch], atomicMin
    uint32_t pos, pow_bits, batch;
    uint64_t count;
};

hipError_t pk_pp_rows(const PpArgs &a, hipStream_t st);
hipError_t pk_pp_scan(const uint64_t *rowprod, uint64_t *z, uint64_t n, uint32_t nch_total, hipStream_t st);   // [batch * nch][n], contiguous
hipError_t pk_pp_finish(const PpArgs &a, const uint64_t *z, uint64_t *zs_pp, uint64_t ps_z, uint64_t ps_zs, hipStream_t st);
hipError_t pk_quotient(const QuotientArgs &a, const GateDev *host_gates, hipStream_t st);
hipError_t pk_gate_sums(const QuotientArgs &a, const GateDev *host_gates, hipStream_t st);
// result: [batch][2]
hipError_t pk_witness_check(const uint64_t *acc, uint64_t n, uint32_t nch, const uint64_t *z, const uint64_t *rowprod, uint64_t *result, uint32_t batch, hipStream_t st);
hipError_t pk_scale_powers(uint64_t *data, uint64_t n, uint64_t ncols, const uint64_t *pw_lo, const uint64_t *pw_hi, uint32_t lo_bits, hipStream_t st);
// out[b][point][poly]; coeffs of proof b at coeffs + b * ps_coeffs (0: shared), its points at points + b * ps_points
hipError_t pk_poly_eval(const uint64_t *coeffs, uint64_t n, uint32_t npolys, const gl::e2 *points, uint32_t npoints, gl::e2 *out,
                        uint32_t batch, uint64_t ps_coeffs, uint64_t ps_points, uint64_t ps_out, hipStream_t st);
hipError_t pk_reduce_polys(const ReduceArgs &a, hipStream_t st);
// zs / shifts: [batch] device tables; comp and fin are per-proof with strides ps_comp / ps_fin
hipError_t pk_divide_linear(const uint64_t *comp_a, const uint64_t *comp_b, uint64_t n, const gl::e2 *zs, const gl::e2 *shifts, int mode,
                            uint64_t *fin_a, uint64_t *fin_b, uint32_t batch, uint64_t ps_comp, uint64_t ps_fin, hipStream_t st);
// device-side copy of a small block; dst or src may be pinned host memory (used in place of the runtime's small-copy path)
hipError_t pk_copy(void *dst, const void *src, size_t bytes, hipStream_t st);
// dense rows -> dst + r * pitch_words (the upload-side counterpart)
hipError_t pk_unpack_rows(const uint64_t *src, uint64_t pitch_words, uint64_t width_words, uint64_t rows, uint64_t *dst, hipStream_t st);
hipError_t pk_pack_rows(const uint64_t *src, uint64_t pitch_words, uint64_t width_words, uint64_t rows, uint64_t *dst, hipStream_t st);
hipError_t pk_interleave_ext(const uint64_t *va, const uint64_t *vb, uint64_t n, uint64_t *rows, uint32_t batch, uint64_t ps_vals, uint64_t ps_rows, hipStream_t st);
hipError_t pk_fri_fold(const uint64_t *ca, const uint64_t *cb, uint64_t new_n, uint32_t arity, const gl::e2 *betas, uint64_t *oa, uint64_t *ob,
                       uint32_t batch, uint64_t ps_in, uint64_t ps_out, hipStream_t st);
struct HasherDev;
hipError_t pk_pow(const PowArgs &a, const HasherDev &h, hipStream_t st);   // defined next to the Poseidon constants (merkle_kernels.hip)
// idx: [batch][nq]; out: per proof at out + b * ps_out
hipError_t pk_gather_rows(const uint64_t *cols, uint64_t stride, uint32_t ncols, const uint64_t *idx, uint32_t nq, uint64_t *out,
                          uint32_t batch, uint64_t ps_cols, uint64_t ps_out, hipStream_t st);
hipError_t pk_gather_paths(const uint64_t *digests, uint64_t n_leaves, uint32_t path_len, const uint64_t *idx, uint32_t shift, uint32_t nq, uint64_t *out,
                           uint32_t batch, uint64_t ps_digests, uint64_t ps_out, hipStream_t st);
hipError_t pk_gather_leaf_rows(const uint64_t *rows, uint32_t width, const uint64_t *idx, uint32_t shift, uint32_t nq, uint64_t *out,
                               uint32_t batch, uint64_t ps_rows, uint64_t ps_out, hipStream_t st);
// salt columns [batch][4][lde_n]: ChaCha20 keyed per proof (keys: [batch][8] 32-bit words on the device), stream = oracle_index
hipError_t pk_salt(const uint32_t *keys, uint32_t oracle_index, uint64_t lde_n, uint64_t *out, uint32_t batch, hipStream_t st);
// `count` uniform field elements per witness from ChaCha20 under keys[b] ([batch][8] 32-bit words on the device), out: [batch][pitch]
hipError_t pk_random_felts(const uint32_t *keys, uint64_t count, uint64_t *out, uint64_t pitch, uint32_t batch, hipStream_t st);
hipError_t pk_coset_tables(uint64_t lde_n, uint32_t log_lde, const uint64_t *pw_lo, const uint64_t *pw_hi, uint32_t lo_bits, const uint64_t *zh,
                           uint32_t rate, uint64_t n_field, uint64_t *x_coset, uint64_t *l0_coset, hipStream_t st);
