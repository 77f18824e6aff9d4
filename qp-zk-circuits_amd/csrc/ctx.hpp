// ctx.hpp — library context: one GPU, one stream, cached plans and scratch.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <map>
#include <memory>
#include <string>
#include <vector>
#include "../../include/qpgpu.h"
#include "merkle.hpp"
#include "poseidon.hpp"

struct NttTables;  // ntt_plan.cpp

struct qpgpu_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;
    // scratch buffer reused by multi-pass transforms
    uint64_t *scratch = nullptr;
    size_t scratch_bytes = 0;
    std::map<std::string, std::shared_ptr<NttTables>> ntt_tables;
    std::vector<void *> owned;  // table allocations freed at destroy
    // plan_only: ntt_run goes through its planning (twiddle / coset tables uploaded and cached, scratch sized) without
    // launching anything. A circuit load replays the transforms a proof will issue this way, so that the proving threads
    // never allocate, free or upload.
    bool plan_only = false;
    // the proof-system hasher of everything created on this context (copied from the process default at creation,
    // changed by qpgpu_ctx_set_hasher until the first circuit or oracle is created here)
    hasher::Config hasher;
    poseidon2::Params *d_p2 = nullptr;       // device copy of hasher.p2 (Poseidon2 only)
    bool hasher_in_use = false;
    // the APPLICATION hash of the Wormhole circuits (Poseidon2 with qp-poseidon-core's parameters: the Poseidon2 gate's
    // constants and the pad-10 sponge), independent of which permutation is the proof-system hasher above
    poseidon2::Params *d_p2_app = nullptr;
    int ensure_p2_app();
    HasherDev hasher_dev() const {
        HasherDev h; h.kind = hasher.kind; h.p2 = d_p2;
        h.qp = hasher.kind == hasher::POSEIDON2 && hasher_is_qp();
        return h;
    }
    bool hasher_is_qp() const;   // the context's Poseidon2 block equals poseidon2::qp_params() (merkle_api.cpp)

    // optional per-kernel timing with HIP events on `stream` (bench.py's roofline leg)
    struct KStat { double ms = 0; uint64_t launches = 0; };
    struct Pending { std::string name; hipEvent_t e0, e1; };
    bool profiling = false;
    std::map<std::string, KStat> kstats;
    std::vector<Pending> pending;
    std::vector<size_t> prof_stack;   // indices into pending: begin/end pairs may nest
    void prof_begin(const char *name);
    void prof_end();
    int prof_collect();

    int fail(int code, const std::string &msg) { err = msg; return code; }
    int hip_fail(hipError_t e, const char *what) {
        err = std::string(what) + ": " + hipGetErrorString(e);
        return QPGPU_EDEVICE;
    }
    int ensure_scratch(size_t bytes);
    // small device-to-host reads on the proving path (caps, openings, query data): a kernel writes them into a pinned host
    // buffer in place (read_back_2d packs the per-proof rows on the way), then one stream sync. The runtime's own small-copy
    // path is not used on the proving path: with several proving threads it faulted inside hipMemcpyAsync now and then under
    // rocprofv3 (a host memcpy to or from device memory through the BAR mapping; the same fault has since been seen once inside a
    // plain kernel launch, so this lowers the exposure, it does not remove it), and a launch costs no more than a copy command.
    void *h_pin = nullptr;
    size_t h_pin_bytes = 0;
    int read_back(void *host_dst, const void *dev_src, size_t bytes);
    // size the pinned buffer up front (circuit load: every read-back of a batch is a piece of its proofs), so that proving
    // neither allocates nor frees
    int reserve_read_back(size_t bytes);
    // `rows` pieces of `width` bytes, `src_pitch` bytes apart on the device, packed back to back on the host
    int read_back_2d(void *host_dst, const void *dev_src, size_t src_pitch, size_t width, size_t rows);
    int upload(const std::vector<uint64_t> &host, uint64_t **dptr);
};

// HIP's current device is per host thread: every entry point selects the ctx's GPU first
#define QP_DEV(ctx) do { hipError_t _e = hipSetDevice((ctx)->device); if (_e != hipSuccess) return (ctx)->hip_fail(_e, "hipSetDevice"); } while (0)
#define QP_HIP(ctx, call) do { hipError_t _e = (call); if (_e != hipSuccess) return (ctx)->hip_fail(_e, #call); } while (0)

int merkle_ensure_constants(qpgpu_ctx *ctx);   // Poseidon round constants on the device, the context's Poseidon2 block uploaded
// leaf hashing + all levels down to the cap for `leaf.batch` trees (digest arrays at d_digests + b * leaf.ps_digests)
int merkle_build(qpgpu_ctx *ctx, const MerkleLeafArgs &leaf, unsigned log_leaves, unsigned cap_height, uint64_t *d_digests);
// `batch` columns per proof; with nproofs > 1 proof b reads at d_in + b * in_ps and writes at d_out + b * out_ps (words)
struct NttProofs { uint32_t nproofs = 1; uint64_t in_ps = 0, out_ps = 0; };
int ntt_run(qpgpu_ctx *ctx, const uint64_t *d_in, uint64_t *d_out, unsigned log_n_in, unsigned log_n_out,
            size_t batch, bool inverse, bool out_bitrev, uint64_t coset_shift, NttProofs np = NttProofs());
