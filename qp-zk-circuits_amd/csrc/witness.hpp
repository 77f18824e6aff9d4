// witness.hpp — stage s1 on the device: plonky2's `generate_partial_witness` for the gate generators this backend
// knows. The plan (built once per circuit on the host) is the dependency-levelled list of generator instances;
// each level is one kernel launch. See witness_plan.cpp.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include "prover_kernels.hpp"

// one generator instance = one gate generator of plonky2 (a whole row for most gates, one operation of the row for
// Arithmetic / ArithmeticExtension / MulExtension / RandomAccess copies / Constant wires)
struct WitnessInst { uint32_t row, gate, op; };   // gate == WITNESS_HINT: `row` indexes the hint list, `op` is its opcode
constexpr uint32_t WITNESS_HINT = 0xFFFFFFFFu;

struct WitnessArgs {
    uint64_t *wires;            // [num_wires][n] in place
    const uint32_t *src_of;     // [n][num_routed] flat index (col*n + row) of the cell every member of the copy class reads
    const WitnessInst *insts;   // sorted by level
    const GateDev *gates;
    const uint64_t *cs;         // constants_sigmas VALUES [ncs][n] (constants of the row)
    const uint64_t *poseidon_rc, *poseidon_fast;
    const poseidon2::Params *p2_gate;   // Poseidon2 gate: qp-poseidon-core's constants (device block) and the gate's wire layout
    P2GateLayout p2_layout;
    const uint64_t *hints;      // [n_hints][8]: the pack's free-standing generators (circuit.hpp)
    uint32_t num_wires;
    const uint64_t *pi_hash;    // [batch][4]: PublicInputGate wires
    uint64_t n;
    uint64_t batch_stride;      // words between the wire matrices of a batch (blockIdx.y = witness index)
    uint32_t num_routed, num_selectors;
};

hipError_t wk_run_level(const WitnessArgs &a, uint32_t first, uint32_t count, uint32_t batch, hipStream_t st);
hipError_t wk_run_combined(const WitnessArgs &a, uint32_t first, uint32_t n_generic, uint32_t n_poseidon, uint32_t batch, hipStream_t st);
// A run of consecutive narrow levels [l0, l1) in one launch: one workgroup per witness walks the levels with a workgroup barrier
// between them (level l = insts[level_start[l] .. level_start[l+1]), its PoseidonGate rows from level_poseidon[l] on).
constexpr uint32_t WITNESS_RUN_GENERIC_CAP = 1024, WITNESS_RUN_POSEIDON_CAP = 32;   // widest level a run takes
hipError_t wk_run_levels(const WitnessArgs &a, const uint32_t *d_level_start, const uint32_t *d_level_poseidon, uint32_t l0, uint32_t l1, uint32_t batch, hipStream_t st);
hipError_t wk_run_poseidon(const WitnessArgs &a, uint32_t first, uint32_t count, uint32_t batch, hipStream_t st);   // PoseidonGate instances, 16 lanes each
hipError_t wk_fill_copies(const WitnessArgs &a, uint32_t batch, hipStream_t st);
// wires[b * batch_stride + idx[i]] = vals[b * val_stride + i] (flat cell index = column * n + row); out[i] = wires[idx[i]]
hipError_t wk_scatter(uint64_t *wires, const uint32_t *idx, const uint64_t *vals, uint32_t count, uint32_t batch, uint64_t batch_stride, uint32_t val_stride, hipStream_t st);
hipError_t wk_gather(const uint64_t *wires, const uint32_t *idx, uint64_t *out, uint32_t count, hipStream_t st);
// a partition set twice: err[2 * b + slot] = min over failing comparisons of (index), 0xFFFFFFFF = none (the caller presets it).
// pairs: wires[own[i]] against wires[src[i]]; vals: wires[idx[i]] against vals[b * val_stride + i] (flat cells, canonical compare)
hipError_t wk_check_nonzero(const uint64_t *wires, const uint32_t *a, const uint32_t *b, uint32_t count, uint32_t batch, uint64_t batch_stride, uint32_t *err, uint32_t slot, hipStream_t st);
hipError_t wk_check_pairs(const uint64_t *wires, const uint32_t *own, const uint32_t *src, uint32_t count, uint32_t batch, uint64_t batch_stride, uint32_t *err, uint32_t slot, hipStream_t st);
hipError_t wk_check_vals(const uint64_t *wires, const uint32_t *idx, const uint64_t *vals, uint32_t count, uint32_t batch, uint64_t batch_stride, uint32_t val_stride, uint32_t *err, uint32_t slot, hipStream_t st);
