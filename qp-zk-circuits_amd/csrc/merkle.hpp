// merkle.hpp — launch interface of the Poseidon / Merkle kernels (merkle_kernels.hip).
//
// Every launcher works on a lockstep batch of `batch` trees of the same shape: tree b reads its leaves at
// src + b * ps_src and writes its digest array at digests + b * ps_digests (strides in words). Leaf counts are powers
// of two. The hashing permutation is a property of the caller's context (HasherDev), not of the process.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

namespace poseidon2 { struct Params; }
// which permutation the hashing kernels run: plonky2's Poseidon (constants in __constant__ memory, fixed) or Poseidon2 with
// the context's parameter block (device pointer)
// qp: the parameter block is qp-poseidon-core's set (poseidon2::qp_params): its external block has a multiplication-free form and
// its internal rounds a matrix-pipe form (merkle_kernels_mx.hip); any other block runs the general plug
struct HasherDev { int kind = 0; const poseidon2::Params *p2 = nullptr; bool qp = false; };

struct MerkleLeafArgs {
    const uint64_t *src0;   // column-major: column c at src0 + c*stride0, leaf j at slot j
    const uint64_t *src1;   // optional second column group (e.g. salt columns); may be null when ncols1 == 0
    uint64_t stride0, stride1;
    uint32_t ncols0, ncols1;
    uint64_t n_leaves;      // per tree, a power of two
    uint64_t *digests;      // n_leaves x 4 per tree
    uint32_t batch;         // trees (0 is read as 1)
    uint64_t ps_src0, ps_src1, ps_digests;
};

uint64_t qpgpu_tp_min_threads();   // launches at least this large take the throughput build of a hashing kernel (merkle_kernels.hip)
hipError_t merkle_upload_constants(const uint64_t *rc360);   // plonky2 Poseidon round constants, once per device
bool merkle_mx_in_use();                                      // false under QPGPU_MX=0
hipError_t merkle_upload_p2_tables(const poseidon2::Params &qp);   // matrix-form table of qp-poseidon-core's Poseidon2, once per device
hipError_t merkle_leaf_hash(const MerkleLeafArgs &a, const HasherDev &h, hipStream_t st);
hipError_t merkle_leaf_hash_rows(const uint64_t *rows, uint64_t n_leaves, uint32_t width, uint64_t *digests, uint32_t batch, uint64_t ps_rows, uint64_t ps_digests,
                                 const HasherDev &h, hipStream_t st);
// all levels above `levels` (cnt digests per tree, the following levels stored behind them) down to the cap
hipError_t merkle_reduce_to_cap(uint64_t *levels, uint64_t cnt, uint64_t cap_n, uint32_t batch, uint64_t ps_digests, const HasherDev &h, hipStream_t st);
hipError_t poseidon_permute_batch(uint64_t *states, uint64_t n, const HasherDev &h, hipStream_t st);
// `count` preimages of `len` elements (row-major) through the qp fork's Poseidon2 sponge (pad `|| 1 || 0*`, additive absorption)
// qp_set: p2 is qp-poseidon-core's parameter set (its external block has a multiplication-free form)
hipError_t poseidon2_hash_pad10_batch(const uint64_t *in, uint64_t len, uint64_t count, uint64_t *out, const poseidon2::Params *p2, bool qp_set, hipStream_t st);
