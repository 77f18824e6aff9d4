// circuit_state.hpp — a loaded circuit: setup-time residents and the per-proof workspace (one per in-flight proof).
#pragma once
#include <hip/hip_runtime.h>
#include <algorithm>
#include <vector>
#include "circuit.hpp"
#include "ctx.hpp"
#include "prover_host.hpp"
#include "prover_kernels.hpp"

struct WitnessPlan;
void witness_plan_free(WitnessPlan *p);

struct qpgpu_circuit {
    qpgpu_ctx *ctx = nullptr;
    CircuitPack pack;
    std::vector<void *> allocs;
    // setup-time residents
    gl::u64 *d_cs_values = nullptr;
    PolyOracle cs;
    GateDev *d_gates = nullptr;
    std::vector<GateDev> h_gates;
    gl::u64 *d_qacc = nullptr;
    gl::u64 *d_poseidon_rc = nullptr, *d_poseidon_fast = nullptr;
    gl::u64 *d_omega = nullptr, *d_x_coset = nullptr, *d_l0_coset = nullptr, *d_zh_inv = nullptr;
    gl::u64 *d_ginv_lo = nullptr, *d_ginv_hi = nullptr; uint32_t ginv_lo_bits = 0;
    // per-proof workspace
    gl::u64 *d_wires_vals = nullptr;
    PolyOracle wires, zs, quot;
    gl::u64 *d_qcp = nullptr, *d_rowprod = nullptr, *d_z = nullptr, *d_zs_vals = nullptr;
    gl::u64 *d_small = nullptr;          // betas, gammas, beta_k_is, alpha pows, pi hash
    gl::e2 *d_points = nullptr, *d_open = nullptr;
    FriParams fri;
    FriWork fri_work;                // s8..s11 workspace, one allocation
    Stager stage;                    // pinned host staging for the small per-proof tables (no sync on upload)
    bool seed_set = false;
    bool check_witness = false;
    gl::u64 *d_check = nullptr;          // [2]: first bad row, permutation flag
    gl::u64 blinding_seed = 0;
    WitnessPlan *wplan = nullptr;    // stage s1, built on first use (witness_plan.cpp)

    template <class T> int alloc(T **p, size_t count) {
        void *v = nullptr;
        hipError_t e = hipMalloc(&v, std::max<size_t>(count * sizeof(T), 8));
        if (e != hipSuccess) return ctx->hip_fail(e, "hipMalloc(circuit)");
        allocs.push_back(v);
        *p = (T *)v;
        return QPGPU_OK;
    }
};
