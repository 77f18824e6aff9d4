// ctx.hpp — library context: one GPU, one stream, cached plans and scratch.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <map>
#include <memory>
#include <string>
#include <vector>
#include "../../include/qpgpu.h"

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
    unsigned hasher_generation = 0;  // hasher::generation() whose constants this ctx last uploaded to __constant__ memory

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
    // small device-to-host reads on the proving path (caps, openings, query data): through a pinned bounce buffer, then a
    // stream sync — a pageable destination makes the runtime stage the copy itself, tens of microseconds per read
    void *h_pin = nullptr;
    size_t h_pin_bytes = 0;
    int read_back(void *host_dst, const void *dev_src, size_t bytes);
    int upload(const std::vector<uint64_t> &host, uint64_t **dptr);
};

// HIP's current device is per host thread: every entry point selects the ctx's GPU first
#define QP_DEV(ctx) do { hipError_t _e = hipSetDevice((ctx)->device); if (_e != hipSuccess) return (ctx)->hip_fail(_e, "hipSetDevice"); } while (0)
#define QP_HIP(ctx, call) do { hipError_t _e = (call); if (_e != hipSuccess) return (ctx)->hip_fail(_e, #call); } while (0)

struct MerkleLeafArgs;
int merkle_ensure_constants(qpgpu_ctx *ctx);
int merkle_build(qpgpu_ctx *ctx, const MerkleLeafArgs &leaf, unsigned log_leaves, unsigned cap_height, uint64_t *d_digests);
int ntt_run(qpgpu_ctx *ctx, const uint64_t *d_in, uint64_t *d_out, unsigned log_n_in, unsigned log_n_out,
            size_t batch, bool inverse, bool out_bitrev, uint64_t coset_shift);
