// prover_host.hpp — host-side pieces shared by the all-in-one prover (prover.cpp) and the stage-level
// C ABI (oracle_api.cpp): the duplex challenger, committed polynomial batches ("oracles", plonky2's
// PolynomialBatch), the FRI opening prover and the proof byte writer.
#pragma once
#include <hip/hip_runtime_api.h>
#include <cstring>
#include <vector>
#include "ctx.hpp"
#include "gl64.hpp"
#include "poseidon.hpp"

// ---- host duplex challenger (plonky2::iop::challenger::Challenger) ----
struct Challenger {
    gl::u64 state[12] = {0};
    gl::u64 in[8]; int n_in = 0;
    gl::u64 out[8]; int n_out = 0;
    void duplex() {
        for (int i = 0; i < n_in; i++) state[i] = in[i];
        n_in = 0;
        hasher::host_permute(state);
        std::memcpy(out, state, sizeof out);
        n_out = 8;
    }
    void observe(const gl::u64 *x, size_t n) {
        for (size_t i = 0; i < n; i++) { n_out = 0; in[n_in++] = gl::canon(x[i]); if (n_in == 8) duplex(); }
    }
    gl::u64 get() { if (n_in > 0 || n_out == 0) duplex(); return out[--n_out]; }
    gl::e2 get_ext() { gl::u64 a = get(), b = get(); return gl::e2_make(a, b); }
};

struct ByteWriter {
    uint8_t *p; size_t cap, len = 0; bool overflow = false;
    void u64le(gl::u64 v) { if (len + 8 > cap) { overflow = true; len += 8; return; } std::memcpy(p + len, &v, 8); len += 8; }
    void u8(uint8_t v) { if (len + 1 > cap) { overflow = true; len += 1; return; } p[len++] = v; }
    void vec(const gl::u64 *v, size_t n) { for (size_t i = 0; i < n; i++) u64le(v[i]); }
    void ext(gl::e2 v) { u64le(v.a); u64le(v.b); }
};

// One committed polynomial batch, resident on the device. Column-major everywhere: coefficient c of polynomial j at
// coeffs[j*n + c]; LDE value of polynomial j at leaf slot s (= point g*w^bitrev(s)) at lde[j*lde_n + s].
struct PolyOracle {
    uint32_t ncols = 0;
    unsigned log_n = 0, rate_bits = 0, cap_h = 0;
    gl::u64 *coeffs = nullptr, *lde = nullptr, *digests = nullptr;
    gl::u64 *salt = nullptr;            // [4][lde_n] when the oracle is blinded
    uint32_t oracle_index = 0;          // selects the salt stream of a blinded oracle
    std::vector<gl::u64> cap;
    gl::u64 lde_n() const { return 1ull << (log_n + rate_bits); }
    unsigned log_lde() const { return log_n + rate_bits; }
};

inline size_t digest_words(unsigned log_leaves, unsigned cap_h) { return ((2ull << log_leaves) - (1ull << cap_h)) * 4; }

// Pinned host staging for small per-proof tables: asynchronous uploads without a stream sync. A region is used once
// per proof; the owner resets `pos` when the previous proof has completed.
struct Stager {
    gl::u64 *h = nullptr;
    size_t words = 0, pos = 0;
    int put(qpgpu_ctx *ctx, void *dst, const void *src, size_t bytes);
};

struct FriParams {
    unsigned degree_bits = 0, rate_bits = 0, cap_h = 0, pow_bits = 0;
    uint32_t num_queries = 0;
    std::vector<unsigned> arity_bits;
};
struct FriRange { uint32_t oracle, first, count; };
struct FriBatch { gl::e2 point; std::vector<FriRange> ranges; };

// Device workspace of one FRI opening proof; carved out of a single allocation.
struct FriWork {
    gl::u64 *comp = nullptr, *fin = nullptr, *vals = nullptr, *coeffs[2] = {nullptr, nullptr};
    std::vector<gl::u64 *> digests, leafrows;
    gl::u64 *pow = nullptr, *qidx = nullptr, *gather = nullptr;
    gl::e2 *alpha_ext = nullptr;
    size_t gather_words = 0, max_batch_polys = 0;
    // leaf widths (felts, salts included) of the initial oracles, in order
    static size_t words(const FriParams &p, const std::vector<size_t> &leaf_widths, size_t max_batch_polys);
    void bind(gl::u64 *base, const FriParams &p, const std::vector<size_t> &leaf_widths, size_t max_batch_polys);
    static size_t stage_words(const FriParams &p, size_t max_batch_polys) { return 2 * max_batch_polys * 2 + p.num_queries + 16; }
};

// PolynomialBatch::from_coeffs / from_values on device-resident columns (fri/oracle.rs). The cap lands in o.cap
// (stream is synchronised). blinding_seed is used when o.salt is set.
int oracle_commit_coeffs(qpgpu_ctx *ctx, PolyOracle &o, gl::u64 blinding_seed);
int oracle_commit_values(qpgpu_ctx *ctx, const gl::u64 *d_values, PolyOracle &o, gl::u64 blinding_seed);

// PolynomialBatch::prove_openings + fri_proof: squeezes the FRI alpha, builds the batched opening polynomial, runs the
// commit phase, the proof of work and the query phase against `ch`, and appends FriProof bytes (write_fri_proof order).
int fri_prove(qpgpu_ctx *ctx, const FriParams &p, const PolyOracle *const *oracles, size_t n_oracles,
              const std::vector<FriBatch> &batches, Challenger &ch, FriWork &w, Stager &stage, ByteWriter &out);
size_t fri_proof_bytes(const FriParams &p, const std::vector<size_t> &leaf_widths);

#define QP_TRY(expr) do { int _rc = (expr); if (_rc) return _rc; } while (0)
