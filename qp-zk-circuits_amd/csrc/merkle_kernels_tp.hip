// merkle_kernels_tp.hip — throughput build of the thread-per-hash kernels (merkle_hash_impl.hpp): the S-box products of the
// Poseidon permutation are compiled as rare-fold groups (gl64.hpp: lazy forms; a product is 19 instead of 22 vector instructions
// and a stage of a layer has one wave-uniform branch), which pays when enough waves are resident to hide a scalar branch's
// latency. merkle_kernels.hip routes launches of qpgpu_tp_min_threads() threads or more here.
#define POSEIDON_GROUPED_SBOX 1
#include <hip/hip_runtime.h>
#include "merkle.hpp"
#include "poseidon.hpp"
#include "prover_kernels.hpp"

using gl::u32;
using gl::u64;

namespace tp {
__constant__ u64 c_poseidon_rc[poseidon::ROUNDS * poseidon::WIDTH];
#include "merkle_hash_impl.hpp"
}  // namespace tp

hipError_t merkle_tp_upload_constants(const u64 *rc360) {
    return hipMemcpyToSymbol(HIP_SYMBOL(tp::c_poseidon_rc), rc360, sizeof(u64) * poseidon::ROUNDS * poseidon::WIDTH);
}
hipError_t merkle_tp_leaves(const MerkleLeafArgs &a, u64 total, const HasherDev &h, hipStream_t st) { return tp::hash_launch_leaves(a, total, h, st); }
hipError_t merkle_tp_rows(const u64 *rows, u64 n_leaves, u32 width, u64 *digests, u32 batch, u64 ps_rows, u64 ps_digests, const HasherDev &h, hipStream_t st) {
    return tp::hash_launch_rows(rows, n_leaves, width, digests, batch, ps_rows, ps_digests, h, st);
}
hipError_t merkle_tp_nodes(const u64 *in, u64 *out, u64 n_out, u32 batch, u64 ps, const HasherDev &h, hipStream_t st) { return tp::hash_launch_nodes(in, out, n_out, batch, ps, h, st); }
hipError_t merkle_tp_pow(const PowArgs &a, dim3 g, const HasherDev &h, hipStream_t st) { return tp::hash_launch_pow(a, g, h, st); }
