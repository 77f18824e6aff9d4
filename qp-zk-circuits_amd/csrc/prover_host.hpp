// prover_host.hpp — host-side pieces shared by the all-in-one prover (prover.cpp) and the stage-level
// C ABI (oracle_api.cpp): the duplex challenger, committed polynomial batches ("oracles", plonky2's
// PolynomialBatch), the FRI opening prover and the proof byte writer.
//
// Everything here works on a lockstep batch of `nb` proofs of one circuit: each stage is launched once for all of
// them (grid.z / folded leading dimension = proof), the nb Fiat-Shamir transcripts advance together on the host, and
// there is one stream synchronisation per stage for the whole batch. A single proof is the batch of one.
#pragma once
#include <hip/hip_runtime_api.h>
#include <cstring>
#include <vector>
#include "ctx.hpp"
#include "gl64.hpp"
#include "poseidon.hpp"

// ---- host duplex challenger (plonky2::iop::challenger::Challenger) ----
struct Challenger {
    const hasher::Config *h;
    gl::u64 state[12] = {0};
    gl::u64 in[8]; int n_in = 0;
    gl::u64 out[8]; int n_out = 0;
    explicit Challenger(const hasher::Config &cfg) : h(&cfg) {}
    void duplex() {
        for (int i = 0; i < n_in; i++) state[i] = in[i];
        n_in = 0;
        h->permute(state);
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

// One committed polynomial batch per proof, resident on the device. Column-major everywhere: coefficient c of polynomial j
// at coeffs[j*n + c]; LDE value of polynomial j at leaf slot s (= point g*w^bitrev(s)) at lde[j*lde_n + s]. With nb > 1
// the buffers are [nb][...] arrays (proof b at + b * ps_*); an oracle shared by all proofs of a batch (constants/sigmas)
// has nb == 1 and strides 0.
struct PolyOracle {
    uint32_t ncols = 0;
    unsigned log_n = 0, rate_bits = 0, cap_h = 0;
    gl::u64 *coeffs = nullptr, *lde = nullptr, *digests = nullptr;
    gl::u64 *salt = nullptr;            // [4][lde_n] per proof when the oracle is blinded
    uint32_t nb = 1;
    gl::u64 ps_coeffs = 0, ps_lde = 0, ps_digests = 0, ps_salt = 0;
    uint32_t oracle_index = 0;          // selects the salt stream of a blinded oracle
    std::vector<gl::u64> cap;           // [nb][4 << cap_h]
    gl::u64 lde_n() const { return 1ull << (log_n + rate_bits); }
    unsigned log_lde() const { return log_n + rate_bits; }
    size_t cap_words() const { return (size_t)4 << cap_h; }
    const gl::u64 *cap_of(uint32_t b) const { return cap.data() + (size_t)(nb == 1 ? 0 : b) * cap_words(); }
    // dense per-proof strides for a batch of nb
    void set_batch(uint32_t n_proofs, bool shared = false);
};

inline size_t digest_words(unsigned log_leaves, unsigned cap_h) { return ((2ull << log_leaves) - (1ull << cap_h)) * 4; }

inline void PolyOracle::set_batch(uint32_t n_proofs, bool shared) {
    nb = shared ? 1 : n_proofs;
    if (shared) { ps_coeffs = ps_lde = ps_digests = ps_salt = 0; return; }
    ps_coeffs = (gl::u64)ncols << log_n; ps_lde = (gl::u64)ncols << log_lde();
    ps_digests = digest_words(log_lde(), cap_h); ps_salt = (gl::u64)4 << log_lde();
}

// Pinned host staging for small per-batch tables: asynchronous uploads without a stream sync. A region is used once
// per batch; the owner resets `pos` when the previous batch has completed.
struct Stager {
    gl::u64 *h = nullptr;
    size_t words = 0, pos = 0;
    int put(qpgpu_ctx *ctx, void *dst, const void *src, size_t bytes);
    // `rows` tables of row_words each (src rows src_pitch words apart on the host) to dst + r * dst_pitch: ONE launch for the
    // per-proof tables of a lockstep batch
    int put_rows(qpgpu_ctx *ctx, gl::u64 *dst, size_t dst_pitch, const gl::u64 *src, size_t src_pitch, size_t row_words, size_t rows);
};

struct FriParams {
    unsigned degree_bits = 0, rate_bits = 0, cap_h = 0, pow_bits = 0;
    uint32_t num_queries = 0;
    std::vector<unsigned> arity_bits;
};
struct FriRange { uint32_t oracle, first, count; };
struct FriBatch { std::vector<gl::e2> points; std::vector<FriRange> ranges; };   // points: one per proof (FriBatchInfo per proof)

// Device workspace of the FRI opening proofs of a lockstep batch; carved out of a single allocation: a per-proof block
// of `ws` words (pointers below are proof 0's) followed by the batch-level tables.
struct FriWork {
    gl::u64 ws = 0;                  // words per proof
    uint32_t nb_cap = 0;             // proofs the allocation has room for
    gl::u64 *comp = nullptr, *fin = nullptr, *vals = nullptr, *coeffs[2] = {nullptr, nullptr};
    std::vector<gl::u64 *> digests, leafrows;
    gl::u64 *gather = nullptr;
    gl::e2 *alpha_ext = nullptr;
    size_t gather_words = 0, max_batch_polys = 0;
    // batch-level tables: [nb] each (qidx: [nb][num_queries], pow_states: [nb][12])
    gl::e2 *t_points = nullptr, *t_shifts = nullptr, *t_betas = nullptr;
    gl::u64 *pow_states = nullptr, *pow_bases = nullptr, *pow_results = nullptr, *qidx = nullptr;
    // leaf widths (felts, salts included) of the initial oracles, in order
    static size_t words(const FriParams &p, const std::vector<size_t> &leaf_widths, size_t max_batch_polys, uint32_t nb);
    void bind(gl::u64 *base, const FriParams &p, const std::vector<size_t> &leaf_widths, size_t max_batch_polys, uint32_t nb);
    static size_t stage_words(const FriParams &p, size_t max_batch_polys, uint32_t nb) {
        return (size_t)nb * (2 * max_batch_polys + p.num_queries + 2 * 2 * 4 + 2 * (p.arity_bits.size() + 1) + 14 + 8) + 64;
    }
};

// PolynomialBatch::from_coeffs / from_values on device-resident columns (fri/oracle.rs) for the o.nb proofs of a batch.
// The caps land in o.cap (stream is synchronised). d_keys: [nb][8] 32-bit ChaCha20 key words on the device, used when
// o.salt is set. d_values: [nb][ncols][n], dense.
int oracle_commit_coeffs(qpgpu_ctx *ctx, PolyOracle &o, const uint32_t *d_keys);
int oracle_commit_values(qpgpu_ctx *ctx, const gl::u64 *d_values, PolyOracle &o, const uint32_t *d_keys);

// PolynomialBatch::prove_openings + fri_proof for nb proofs: squeezes the FRI alphas, builds the batched opening
// polynomials, runs the commit phase, the proof of work and the query phase against chs[b], and appends FriProof bytes
// (write_fri_proof order) to outs[b].
int fri_prove(qpgpu_ctx *ctx, const FriParams &p, const PolyOracle *const *oracles, size_t n_oracles,
              const std::vector<FriBatch> &batches, uint32_t nb, Challenger *chs, FriWork &w, Stager &stage, ByteWriter *outs);
size_t fri_proof_bytes(const FriParams &p, const std::vector<size_t> &leaf_widths);

// 256-bit salt keys: from the OS entropy source, or derived from a 64-bit seed (tests: reproducible proofs)
int salt_key_random(uint32_t key[8]);
void salt_key_from_seed(uint64_t seed, uint32_t key[8]);

#define QP_TRY(expr) do { int _rc = (expr); if (_rc) return _rc; } while (0)
