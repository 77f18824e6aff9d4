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
    // witness-derived regions: overwritten by qpgpu_circuit_scrub, by qpgpu_circuit_free, and after every proof of the
    // host-buffer entry qpgpu_prove. The _dev / batch / pool entries leave them resident between proofs (the next proof
    // overwrites them); include/qpgpu.h says so at qpgpu_circuit_scrub.
    std::vector<std::pair<void *, size_t>> secret_allocs;
    uint32_t max_batch = 1;          // proofs the per-proof workspace below has room for (lockstep batch)
    // setup-time residents
    gl::u64 *d_cs_values = nullptr;
    PolyOracle cs;
    GateDev *d_gates = nullptr;
    std::vector<GateDev> h_gates;
    gl::u64 *d_qacc = nullptr;
    gl::u64 *d_poseidon_rc = nullptr, *d_poseidon_fast = nullptr;
    bool has_p2_gate = false;        // the gate list holds a Poseidon2 gate: its constants are the context's d_p2_app block
    gl::u64 *d_omega = nullptr, *d_x_coset = nullptr, *d_l0_coset = nullptr, *d_zh_inv = nullptr;
    gl::u64 *d_ginv_lo = nullptr, *d_ginv_hi = nullptr; uint32_t ginv_lo_bits = 0;
    // per-proof workspace: every buffer is [max_batch][one proof's size]
    gl::u64 *d_wires_vals = nullptr;
    uint32_t *d_salt_keys = nullptr;     // [max_batch][8] ChaCha20 key words of the zero-knowledge salts
    PolyOracle wires, zs, quot;
    gl::u64 *d_qcp = nullptr, *d_rowprod = nullptr, *d_z = nullptr, *d_zs_vals = nullptr;
    gl::u64 *d_small = nullptr;          // per proof: betas, gammas, beta_k_is, alpha pows, pi hash
    size_t small_words = 0;              // words of that table per proof
    gl::e2 *d_points = nullptr, *d_open = nullptr;
    FriParams fri;
    FriWork fri_work;                // s8..s11 workspace, one allocation
    Stager stage;                    // pinned host staging for the small per-proof tables (no sync on upload)
    bool seed_set = false;
    bool check_witness = false;
    gl::u64 *d_check = nullptr;          // [max_batch][2]: first bad row, permutation flag
    gl::u64 blinding_seed = 0;           // test hook: proof b of the next batch uses the key derived from blinding_seed + b
    WitnessPlan *wplan = nullptr;    // stage s1, built on first use (witness_plan.cpp)

    template <class T> int alloc(T **p, size_t count, bool secret = false) {
        void *v = nullptr;
        const size_t bytes = std::max<size_t>(count * sizeof(T), 8);
        hipError_t e = hipMalloc(&v, bytes);
        if (e == hipErrorOutOfMemory) return ctx->fail(QPGPU_ENOMEM, "hipMalloc(circuit): out of device memory");
        if (e != hipSuccess) return ctx->hip_fail(e, "hipMalloc(circuit)");
        allocs.push_back(v);
        if (secret) secret_allocs.push_back({v, bytes});
        *p = (T *)v;
        return QPGPU_OK;
    }
};
