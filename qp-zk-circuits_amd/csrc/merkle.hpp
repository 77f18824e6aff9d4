// merkle.hpp — launch interface of the Poseidon / Merkle kernels (merkle_kernels.hip).
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

struct MerkleLeafArgs {
    const uint64_t *src0;   // column-major: column c at src0 + c*stride0, leaf j at slot j
    const uint64_t *src1;   // optional second column group (e.g. salt columns); may be null when ncols1 == 0
    uint64_t stride0, stride1;
    uint32_t ncols0, ncols1;
    uint64_t n_leaves;
    uint64_t *digests;      // n_leaves x 4
};

hipError_t merkle_upload_constants(const uint64_t *rc360);
namespace poseidon2 { struct Params; }
hipError_t merkle_select_hasher(int kind, const poseidon2::Params *p2);   // which permutation the hashing kernels run
hipError_t merkle_leaf_hash(const MerkleLeafArgs &a, hipStream_t st);
hipError_t merkle_leaf_hash_rows(const uint64_t *rows, uint64_t n_leaves, uint32_t width, uint64_t *digests, hipStream_t st);
hipError_t merkle_reduce_level(const uint64_t *in, uint64_t *out, uint64_t n_out, hipStream_t st);
hipError_t merkle_reduce_to_cap(uint64_t *levels, uint64_t cnt, uint64_t cap_n, hipStream_t st);   // all levels above `levels`, the top fused
hipError_t poseidon_permute_batch(uint64_t *states, uint64_t n, hipStream_t st);
