// ntt_pass.hpp — argument block of one on-chip NTT pass (see ntt_kernels.hip).
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <hip/hip_runtime_api.h>

struct NttPassArgs {
    const uint64_t *in;
    uint64_t *out;
    // geometry: a lane is (row, mm) with lane = row << log_m | mm; element (p, lane) of the input sits at
    //   in[col*in_col_stride + row*in_row_stride + mm*in_l_stride + p*in_p_stride]
    uint64_t in_col_stride, out_col_stride;
    uint64_t in_proof_stride, out_proof_stride;   // grid.z = proof of a lockstep batch (0 when there is one proof)
    uint64_t in_row_stride, in_l_stride, in_p_stride;
    uint64_t out_row_stride, out_l_stride, out_p_stride;
    uint64_t lanes_total;      // lanes per column (rows << log_m)
    uint32_t log_m;            // lanes per row = 2^log_m (row length = 2^(ka+kb+log_m))
    uint32_t log_t;            // lanes per tile = 2^log_t
    uint32_t ka, kb;           // round radices (kb == 0: single round)
    uint32_t p_valid;          // input points p >= p_valid are read as zero (LDE padding)
    uint32_t inverse;          // use inverse roots
    uint32_t load_lane_fast;   // adjacent threads -> adjacent lanes (strided pass) or adjacent points
    uint32_t store_lane_fast;
    uint32_t out_bitrev;       // 1: store at slot position (bit-reversed k), 0: store at k
    uint32_t has_out_scale;
    uint64_t out_scale;        // e.g. 1/N on the last inverse pass
    const uint64_t *tw_inner;  // 2^(ka+kb) entries: w_R^e (inverse roots if inverse)
    const uint64_t *tw_hi;     // inter-pass twiddle w_rowlen^(mm*k) = tw_hi[e >> bits] * tw_lo[e & mask]
    const uint64_t *tw_lo;     // null: no inter-pass twiddle
    uint32_t tw_lo_bits;
    uint32_t out_loose;        // 1: the output is an intermediate of the transform and need not be canonical
    uint64_t tw_scale;         // non-zero: folded into the running twiddle product (the 1/N of an inverse transform)
    uint32_t tw_mode;          // 0: two table reads per element; 1: per-thread running product; 2: skipped (timing experiments only);
                               // 3: one read per element from the full table tw_full[k << log_m | mm] = w^(mm k) (x tw_scale), one product
    const uint64_t *tw_full;
    int32_t sparse_lv;         // set by ntt_pass_launch: >= 0 when only the first 2^sparse_lv inputs of every round-A thread can be non-zero
    uint32_t row_pitch;        // set by ntt_pass_launch: exchange words between two lanes' LDS rows (ntt_pass_row_pitch)
    uint32_t split_lds;        // set by ntt_pass_launch: exchange the 32-bit halves one after the other (half the LDS per workgroup)
    const uint64_t *in_scale_a;  // coset input scale: x[p, mm] *= a[p] * b[mm]; null: none
    const uint64_t *in_scale_b;
};

bool ntt_pass_uses_split(int ka, int kb);   // split 32-bit LDS exchange (half the LDS per workgroup) for this pass shape
unsigned ntt_pass_row_pitch(int ka, int kb, int log_t);
size_t ntt_pass_lds_bytes(int ka, int kb, int log_t);
hipError_t ntt_pass_init();
hipError_t ntt_pass_launch(const NttPassArgs &a, uint64_t n_tiles, uint64_t n_cols, hipStream_t st, uint32_t n_proofs = 1);
