// pool.cpp — a proving pool: several proofs of one circuit in flight on one GPU.
//
// The reference proves on a Rayon pool and tells callers to keep proving on a dedicated worker
// (wormhole/aggregator/src/aggregator.rs:14-43); independent leaf proofs are the unit of parallelism.
// One proof leaves most of an MI355X idle between its latency-bound stages, so the native counterpart of that worker is
// a small pool: each worker owns a context (HIP stream), a loaded copy of the circuit (constants/sigmas commitment,
// workspace) and a host thread for the Fiat-Shamir transcript; jobs are taken from one queue.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include "ctx.hpp"

struct qpgpu_circuit;
extern "C" size_t qpgpu_circuit_num_public_inputs(const qpgpu_circuit *c);
extern "C" int qpgpu_circuit_set_witness_check(qpgpu_circuit *c, int on);

namespace {
struct Job {
    uint64_t ticket;
    const uint64_t *d_wires, *public_inputs;
    uint8_t *out; size_t out_cap;
};
struct Done { int rc = 1; size_t len = 0; std::string err; };   // rc 1 = pending
}  // namespace

struct qpgpu_pool {
    int device = 0;
    std::vector<qpgpu_ctx *> ctxs;
    std::vector<qpgpu_circuit *> circuits;
    std::vector<std::thread> threads;
    std::mutex mu;
    std::condition_variable cv_job, cv_done;
    std::deque<Job> queue;
    std::vector<Done> done;          // ring indexed by ticket % size
    uint64_t next_ticket = 0, oldest_live = 0;
    bool stopping = false;
    unsigned max_batch = 1;          // proofs a worker takes from the queue at once and proves in lockstep
    std::string err;
};

namespace {
void worker(qpgpu_pool *p, size_t w) {
    (void)hipSetDevice(p->device);
    for (;;) {
        std::vector<Job> js;
        {
            std::unique_lock<std::mutex> lk(p->mu);
            p->cv_job.wait(lk, [&] { return p->stopping || !p->queue.empty(); });
            if (p->queue.empty()) return;   // stopping and drained
            // whatever is queued, up to the lockstep width, split evenly when several workers wait for little work
            size_t take = std::min<size_t>(p->max_batch, p->queue.size());
            if (p->queue.size() < (size_t)p->max_batch * p->circuits.size()) take = std::min<size_t>(take, (p->queue.size() + p->circuits.size() - 1) / p->circuits.size());
            for (size_t i = 0; i < take; i++) { js.push_back(p->queue.front()); p->queue.pop_front(); }
        }
        const uint32_t nb = (uint32_t)js.size();
        std::vector<const uint64_t *> wires(nb), pis(nb);
        std::vector<uint8_t *> outs(nb);
        std::vector<size_t> lens(nb, 0);
        size_t cap = ~(size_t)0;
        for (uint32_t i = 0; i < nb; i++) { wires[i] = js[i].d_wires; pis[i] = js[i].public_inputs; outs[i] = js[i].out; cap = std::min(cap, js[i].out_cap); }
        std::vector<int> rcs(nb, QPGPU_OK);
        std::vector<std::string> errs(nb);
        const int rc = qpgpu_prove_batch_dev(p->circuits[w], wires.data(), nb, pis.data(), outs.data(), cap, lens.data());
        if (rc != QPGPU_OK && nb > 1) {
            // the jobs of a lockstep batch are unrelated callers' proofs, and in the reference a failing prove concerns its
            // caller only: prove them again one at a time, so that the offender alone gets the error (and its own text)
            for (uint32_t i = 0; i < nb; i++) {
                lens[i] = 0;
                rcs[i] = qpgpu_prove_batch_dev(p->circuits[w], &wires[i], 1, &pis[i], &outs[i], js[i].out_cap, &lens[i]);
                if (rcs[i] != QPGPU_OK) errs[i] = qpgpu_last_error(p->ctxs[w]);
            }
        } else if (rc != QPGPU_OK) {
            rcs[0] = rc; errs[0] = qpgpu_last_error(p->ctxs[w]);
        }
        {
            std::lock_guard<std::mutex> lk(p->mu);
            for (uint32_t i = 0; i < nb; i++) {
                Done &d = p->done[js[i].ticket % p->done.size()];
                d.len = lens[i]; d.err = errs[i];
                d.rc = rcs[i];          // last: wait() watches rc
            }
        }
        p->cv_done.notify_all();
    }
}
}  // namespace

extern "C" {

void qpgpu_pool_destroy(qpgpu_pool *p) {
    if (!p) return;
    { std::lock_guard<std::mutex> lk(p->mu); p->stopping = true; }
    p->cv_job.notify_all();
    for (auto &t : p->threads) if (t.joinable()) t.join();
    for (auto *c : p->circuits) qpgpu_circuit_free(c);
    for (auto *c : p->ctxs) qpgpu_ctx_destroy(c);
    delete p;
}

int qpgpu_pool_create(int device, const uint64_t *pack_words, size_t n_words, unsigned workers, qpgpu_pool **out) {
    return qpgpu_pool_create_batched(device, pack_words, n_words, workers, 1, out);
}

int qpgpu_pool_create_batched(int device, const uint64_t *pack_words, size_t n_words, unsigned workers, unsigned max_batch, qpgpu_pool **out) {
    if (!out || !pack_words || workers == 0 || workers > 64 || max_batch == 0 || max_batch > 1024) return QPGPU_EINVAL;
    *out = nullptr;
    qpgpu_pool *p = new qpgpu_pool();
    p->device = device;
    p->max_batch = max_batch;
    p->done.resize(4096);
    for (unsigned w = 0; w < workers; w++) {
        qpgpu_ctx *ctx = nullptr;
        int rc = qpgpu_ctx_create(device, &ctx);
        if (rc != QPGPU_OK) { qpgpu_pool_destroy(p); return rc; }
        p->ctxs.push_back(ctx);
        qpgpu_circuit *c = nullptr;
        rc = qpgpu_circuit_load_batch(ctx, pack_words, n_words, max_batch, &c);
        if (rc != QPGPU_OK) { qpgpu_pool_destroy(p); return rc; }
        p->circuits.push_back(c);
    }
    for (unsigned w = 0; w < workers; w++) p->threads.emplace_back(worker, p, (size_t)w);
    *out = p;
    return QPGPU_OK;
}

size_t qpgpu_pool_proof_size(const qpgpu_pool *p) { return p && !p->circuits.empty() ? qpgpu_proof_size(p->circuits[0]) : 0; }
unsigned qpgpu_pool_workers(const qpgpu_pool *p) { return p ? (unsigned)p->circuits.size() : 0; }
const char *qpgpu_pool_last_error(const qpgpu_pool *p) { return p ? p->err.c_str() : "null pool"; }

// qpgpu_circuit_set_witness_check for every worker; call while no job is queued or running
int qpgpu_pool_set_witness_check(qpgpu_pool *p, int on) {
    if (!p) return QPGPU_EINVAL;
    std::lock_guard<std::mutex> lk(p->mu);
    if (!p->queue.empty() || p->next_ticket != p->oldest_live) { p->err = "pool_set_witness_check: jobs in flight"; return QPGPU_EINVAL; }
    for (auto *c : p->circuits) qpgpu_circuit_set_witness_check(c, on);
    return QPGPU_OK;
}

int qpgpu_pool_submit(qpgpu_pool *p, const uint64_t *d_wires, const uint64_t *public_inputs, uint8_t *out, size_t out_cap, uint64_t *ticket) {
    if (!p || !ticket) return QPGPU_EINVAL;
    // a job is checked on its own here, so that a bad one cannot reach a lockstep batch of other callers' proofs
    const size_t need = qpgpu_pool_proof_size(p);
    const bool has_pis = !p->circuits.empty() && qpgpu_circuit_num_public_inputs(p->circuits[0]) > 0;
    if (!d_wires || !out || (has_pis && !public_inputs) || out_cap < need) {
        std::lock_guard<std::mutex> lk(p->mu);
        p->err = out_cap < need && d_wires && out ? "pool_submit: output buffer smaller than the proof (" + std::to_string(out_cap) + " < " + std::to_string(need) + " bytes)"
                                                  : "pool_submit: null argument";
        return out_cap < need && d_wires && out ? QPGPU_EBUFSIZE : QPGPU_EINVAL;
    }
    {
        std::lock_guard<std::mutex> lk(p->mu);
        if (p->next_ticket - p->oldest_live >= p->done.size()) { p->err = "pool_submit: too many unwaited jobs"; return QPGPU_EBUFSIZE; }
        const uint64_t t = p->next_ticket++;
        p->done[t % p->done.size()] = Done();
        p->queue.push_back({t, d_wires, public_inputs, out, out_cap});
        *ticket = t;
    }
    p->cv_job.notify_one();
    return QPGPU_OK;
}

int qpgpu_pool_wait(qpgpu_pool *p, uint64_t ticket, size_t *out_len) {
    if (!p) return QPGPU_EINVAL;
    std::unique_lock<std::mutex> lk(p->mu);
    if (ticket >= p->next_ticket || ticket < p->oldest_live) { p->err = "pool_wait: unknown ticket"; return QPGPU_EINVAL; }
    Done &d = p->done[ticket % p->done.size()];
    if (d.rc == 2) { p->err = "pool_wait: ticket already waited for"; return QPGPU_EINVAL; }
    p->cv_done.wait(lk, [&] { return d.rc != 1; });
    if (out_len) *out_len = d.len;
    if (d.rc != QPGPU_OK) p->err = d.err;
    const int rc = d.rc;
    // tickets are waited in any order; the window of live tickets advances over the finished prefix
    d.rc = 2;   // consumed
    while (p->oldest_live < p->next_ticket && p->done[p->oldest_live % p->done.size()].rc == 2) p->oldest_live++;
    return rc;
}

}  // extern "C"
