// ntt_kernels.hip — dispatcher for the NTT pass kernels (template in ntt_kernel_impl.hpp, instances in ntt_inst_*.hip).
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "ntt_pass.hpp"

hipError_t ntt_launch_1_0(const NttPassArgs &a, dim3 grid, dim3 block, size_t lds, hipStream_t st);
hipError_t ntt_attr_1_0();
hipError_t ntt_launch_2_0(const NttPassArgs &a, dim3 grid, dim3 block, size_t lds, hipStream_t st);
hipError_t ntt_attr_2_0();
hipError_t ntt_launch_3_0(const NttPassArgs &a, dim3 grid, dim3 block, size_t lds, hipStream_t st);
hipError_t ntt_attr_3_0();
hipError_t ntt_launch_4_0(const NttPassArgs &a, dim3 grid, dim3 block, size_t lds, hipStream_t st);
hipError_t ntt_attr_4_0();
hipError_t ntt_launch_5_0(const NttPassArgs &a, dim3 grid, dim3 block, size_t lds, hipStream_t st);
hipError_t ntt_attr_5_0();
hipError_t ntt_launch_1_1(const NttPassArgs &a, dim3 grid, dim3 block, size_t lds, hipStream_t st);
hipError_t ntt_attr_1_1();
hipError_t ntt_launch_2_1(const NttPassArgs &a, dim3 grid, dim3 block, size_t lds, hipStream_t st);
hipError_t ntt_attr_2_1();
hipError_t ntt_launch_2_2(const NttPassArgs &a, dim3 grid, dim3 block, size_t lds, hipStream_t st);
hipError_t ntt_attr_2_2();
hipError_t ntt_launch_3_2(const NttPassArgs &a, dim3 grid, dim3 block, size_t lds, hipStream_t st);
hipError_t ntt_attr_3_2();
hipError_t ntt_launch_3_3(const NttPassArgs &a, dim3 grid, dim3 block, size_t lds, hipStream_t st);
hipError_t ntt_attr_3_3();
hipError_t ntt_launch_4_3(const NttPassArgs &a, dim3 grid, dim3 block, size_t lds, hipStream_t st);
hipError_t ntt_attr_4_3();
hipError_t ntt_launch_4_4(const NttPassArgs &a, dim3 grid, dim3 block, size_t lds, hipStream_t st);
hipError_t ntt_attr_4_4();
hipError_t ntt_launch_5_4(const NttPassArgs &a, dim3 grid, dim3 block, size_t lds, hipStream_t st);
hipError_t ntt_attr_5_4();
hipError_t ntt_launch_5_5(const NttPassArgs &a, dim3 grid, dim3 block, size_t lds, hipStream_t st);
hipError_t ntt_attr_5_5();

// QPGPU_NTT_SPLIT: 0 whole-word LDS exchange everywhere; 1 (default) split exchange for the 2^9- and 2^10-point passes
bool ntt_pass_uses_split(int ka, int kb) {
    static const int split = [] { const char *e = getenv("QPGPU_NTT_SPLIT"); return e && *e ? atoi(e) : 1; }();
    return split && kb > 0 && ka + kb >= 9;
}

// Row pitch of a tile's LDS image in exchange words (ntt_kernel_impl.hpp: element s of lane l at l * RP + s + (s >> KB)).
// Whole 8-byte words: odd, as before. Split exchange (4-byte ds_write_b32 / ds_read_b32: 32 banks, conflicts counted inside a
// 32-lane half of the wave): a half-wave of the lane-fast thread mappings is T lanes x 32/T consecutive rows of the other
// index, which land on distinct banks exactly when RP = 32/T (mod 32) (T = 16: banks 2 l + m); the row-fast mappings are
// conflict-free for any pitch. With the odd pitch of round 2 the lane-fast mappings were two-way conflicted (counters:
// SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 50 % on the strided pass, 33 % on the natural-order rows pass).
unsigned ntt_pass_row_pitch(int ka, int kb, int log_t) {
    const unsigned base = (1u << (ka + kb)) + (1u << ka);       // a multiple of 32 for the split sizes (ka + kb >= 9)
    static const int tuned = [] { const char *e = getenv("QPGPU_NTT_PITCH"); return e && *e ? atoi(e) : 1; }();   // 0: the odd pitch of round 2
    if (!ntt_pass_uses_split(ka, kb) || !tuned || log_t >= 5 || base % 32) return base + 1;
    return base + (32u >> log_t);
}
size_t ntt_pass_lds_bytes(int ka, int kb, int log_t) {
    if (kb == 0) return 0;
    const size_t rp = ntt_pass_row_pitch(ka, kb, log_t);
    return (rp << log_t) * (ntt_pass_uses_split(ka, kb) ? sizeof(uint32_t) : sizeof(uint64_t));
}

hipError_t ntt_pass_init() {
    hipError_t e;
    if ((e = ntt_attr_1_1()) != hipSuccess) return e;
    if ((e = ntt_attr_2_1()) != hipSuccess) return e;
    if ((e = ntt_attr_2_2()) != hipSuccess) return e;
    if ((e = ntt_attr_3_2()) != hipSuccess) return e;
    if ((e = ntt_attr_3_3()) != hipSuccess) return e;
    if ((e = ntt_attr_4_3()) != hipSuccess) return e;
    if ((e = ntt_attr_4_4()) != hipSuccess) return e;
    if ((e = ntt_attr_5_4()) != hipSuccess) return e;
    if ((e = ntt_attr_5_5()) != hipSuccess) return e;
    return hipSuccess;
}

hipError_t ntt_pass_launch(const NttPassArgs &a_in, uint64_t n_tiles, uint64_t n_cols, hipStream_t st, uint32_t n_proofs) {
    NttPassArgs a = a_in;
    const int ka = a.ka, kb = a.kb;
    a.split_lds = ntt_pass_uses_split(ka, kb) ? 1 : 0;
    // zero-padded input (first pass of an LDE): thread inputs i * 2^kb + m are zero from p_valid on. QPGPU_NTT_SPARSE=0 disables.
    static const int sparse = [] { const char *e = getenv("QPGPU_NTT_SPARSE"); return e && *e ? atoi(e) : 1; }();
    a.sparse_lv = -1;
    if (sparse && kb > 0 && !a.inverse) {       // the LDE kernels exist for ka = 4 (two live inputs) and ka = 5 (four)
        const uint64_t nb = 1ull << kb;
        if (ka == 4 && a.p_valid <= 2 * nb) a.sparse_lv = 1;
        else if (ka == 5 && a.p_valid <= 4 * nb) a.sparse_lv = 2;
    }
    if (n_cols > 65535 || n_proofs > 65535 || n_proofs == 0) return hipErrorInvalidValue;
    dim3 grid((unsigned)n_tiles, (unsigned)n_cols, n_proofs);
    dim3 block((unsigned)(1u << (ka + a.log_t)), 1, 1);
    if (kb == 0) block.x = 1u << a.log_t;
    if (block.x < 64) block.x = 64;
    size_t lds = ntt_pass_lds_bytes(ka, kb, a.log_t);
    a.row_pitch = ntt_pass_row_pitch(ka, kb, a.log_t);
    if (ka == 1 && kb == 0) return ntt_launch_1_0(a, grid, block, lds, st);
    if (ka == 2 && kb == 0) return ntt_launch_2_0(a, grid, block, lds, st);
    if (ka == 3 && kb == 0) return ntt_launch_3_0(a, grid, block, lds, st);
    if (ka == 4 && kb == 0) return ntt_launch_4_0(a, grid, block, lds, st);
    if (ka == 5 && kb == 0) return ntt_launch_5_0(a, grid, block, lds, st);
    if (ka == 1 && kb == 1) return ntt_launch_1_1(a, grid, block, lds, st);
    if (ka == 2 && kb == 1) return ntt_launch_2_1(a, grid, block, lds, st);
    if (ka == 2 && kb == 2) return ntt_launch_2_2(a, grid, block, lds, st);
    if (ka == 3 && kb == 2) return ntt_launch_3_2(a, grid, block, lds, st);
    if (ka == 3 && kb == 3) return ntt_launch_3_3(a, grid, block, lds, st);
    if (ka == 4 && kb == 3) return ntt_launch_4_3(a, grid, block, lds, st);
    if (ka == 4 && kb == 4) return ntt_launch_4_4(a, grid, block, lds, st);
    if (ka == 5 && kb == 4) return ntt_launch_5_4(a, grid, block, lds, st);
    if (ka == 5 && kb == 5) return ntt_launch_5_5(a, grid, block, lds, st);
    return hipErrorInvalidValue;
}
