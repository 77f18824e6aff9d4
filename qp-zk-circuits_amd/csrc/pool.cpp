// pool.cpp — a proving pool: many proofs of one circuit in flight on one GPU or on several.
//
// The reference proves on a Rayon pool and tells callers to keep proving on a dedicated worker
// (wormhole/aggregator/src/aggregator.rs:14-43); independent leaf proofs are the unit of parallelism, and
// north_star shards them one per GPU. One proof leaves most of an MI355X idle between its latency-bound stages, so the native
// counterpart of that worker is a pool: each worker owns a context (HIP stream) on its device, a loaded copy of the circuit
// (constants/sigmas commitment, workspace) and a host thread for the Fiat-Shamir transcript; jobs are taken from one queue. A
// pool over several devices is the same thing with the workers spread over them: the "gather of proof bytes" of a
// single-process deployment is the workers writing into the caller's host buffers.
//
// A job is a witness in one of three forms: a full wire matrix resident on a device (pinned to that device's workers), a full
// wire matrix in host memory (what `generate_partial_witness(..).full_witness()` holds in a patched plonky2 prove()), or the
// PartialWitness values of a prepared cell list (WormholeProver::commit's output, wormhole/prover/src/lib.rs:156-163): the
// worker then runs stage s1 for its whole lockstep batch before stages s2..s12.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <algorithm>
#include <condition_variable>
#include <cstring>
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
enum JobKind : int { JOB_DEVICE = 0, JOB_HOST = 1, JOB_PARTIAL = 2 };
struct Job {
    uint64_t ticket;
    int kind, slot;                      // slot: index of the device whose workers may take it, -1 = any
    const uint64_t *wires;               // JOB_DEVICE: device pointer; JOB_HOST: host pointer
    const uint64_t *public_inputs;
    uint8_t *out; size_t out_cap;
    std::vector<uint64_t> values;        // JOB_PARTIAL: copied at submit, wiped after use (they carry the spend secret)
};
struct Done { int rc = 1; size_t len = 0; std::string err; };   // rc 1 = pending
struct Worker { int slot = 0, device = 0; qpgpu_ctx *ctx = nullptr; qpgpu_circuit *circ = nullptr; uint64_t *d_wit = nullptr; };
void wipe(std::vector<uint64_t> &v) { volatile uint64_t *q = v.data(); for (size_t i = 0; i < v.size(); i++) q[i] = 0; }
}  // namespace

struct qpgpu_pool {
    std::vector<int> devices;        // slot -> HIP device
    std::vector<Worker> workers;
    std::vector<std::thread> threads;
    std::mutex mu;
    std::condition_variable cv_job, cv_done;
    std::deque<Job> queue;
    std::vector<Done> done;          // ring indexed by ticket % size
    uint64_t next_ticket = 0, oldest_live = 0;
    bool stopping = false;
    unsigned max_batch = 1;          // proofs a worker takes from the queue at once and proves in lockstep
    size_t wit_words = 0, num_pis = 0;    // words of one wire matrix; public inputs per proof
    std::vector<uint64_t> cells;     // the prepared PartialWitness cell list (qpgpu_pool_set_partial_cells)
    size_t n_blinding = 0;           // its last n_blinding cells are drawn on the device per proof (qpgpu_pool_set_partial_cells_blinded)
    bool host_witness = false;       // workers own a witness workspace of max_batch matrices
    // Under a profiler that intercepts the HSA queues (rocprofv3) the workers take turns on the device: one thread submits at a
    // time. rocprofiler-sdk's queue-write interceptor has faulted (a read one AQL packet slot past a mapping) in multi-worker
    // runs of rounds 2-4, also with a single consistent ROCm 7.2 stack and no torch in the process (profiles/r04_crash_trace_*.txt,
    // DESIGN.md section 8); it has never faulted with one submitting thread. Per-kernel statistics are unaffected; what a profiled
    // run then does not show is the overlap between workers. QPGPU_POOL_SERIALIZE=0 / 1 overrides the detection.
    bool serialize = false;
    std::mutex device_turn;
    std::string err;
};

namespace {
void finish(qpgpu_pool *p, const std::vector<Job> &js, const std::vector<int> &rcs, const std::vector<size_t> &lens, const std::vector<std::string> &errs) {
    {
        std::lock_guard<std::mutex> lk(p->mu);
        for (size_t i = 0; i < js.size(); i++) {
            Done &d = p->done[js[i].ticket % p->done.size()];
            d.len = lens[i]; d.err = errs[i];
            d.rc = rcs[i];          // last: wait() watches rc
        }
    }
    p->cv_done.notify_all();
}

void worker(qpgpu_pool *p, size_t wi) {
    Worker &w = p->workers[wi];
    (void)hipSetDevice(w.device);
    const size_t per_slot = p->workers.size() / p->devices.size();
    for (;;) {
        std::vector<Job> js;
        {
            std::unique_lock<std::mutex> lk(p->mu);
            auto mine = [&](const Job &j) { return j.slot < 0 || j.slot == w.slot; };
            auto first = [&]() { return std::find_if(p->queue.begin(), p->queue.end(), mine); };
            auto it = first();
            while (it == p->queue.end()) {
                if (p->stopping) return;        // drained as far as this worker is concerned (jobs pinned elsewhere are their workers')
                p->cv_job.wait(lk);
                it = first();
            }
            // whatever is queued for this worker, up to the lockstep width, split evenly when several workers wait for little
            // work; a batch holds jobs of one kind
            size_t avail = 0;
            for (const Job &j : p->queue) if (mine(j) && j.kind == it->kind) avail++;
            const size_t peers = it->slot < 0 ? p->workers.size() : per_slot;
            size_t take = std::min<size_t>(p->max_batch, avail);
            if (avail < (size_t)p->max_batch * peers) take = std::min<size_t>(take, (avail + peers - 1) / peers);
            const int kind = it->kind;
            for (auto q = it; q != p->queue.end() && js.size() < take;) {
                if (mine(*q) && q->kind == kind) { js.push_back(std::move(*q)); q = p->queue.erase(q); }
                else ++q;
            }
        }
        const uint32_t nb = (uint32_t)js.size();
        std::unique_lock<std::mutex> turn(p->device_turn, std::defer_lock);
        if (p->serialize) turn.lock();
        std::vector<int> rcs(nb, QPGPU_OK);
        std::vector<std::string> errs(nb);
        std::vector<size_t> lens(nb, 0);
        std::vector<const uint64_t *> wires(nb), pis(nb);
        std::vector<uint64_t> derived_pis;           // JOB_PARTIAL: the public inputs the batch is proven with (supplied or read out of the witnesses)
        std::vector<uint8_t *> outs(nb);
        for (uint32_t i = 0; i < nb; i++) { wires[i] = js[i].wires; pis[i] = js[i].public_inputs; outs[i] = js[i].out; }
        if (js[0].kind == JOB_HOST) {
            // the wire matrices go up into this worker's workspace on its own stream (the caller's memory is read here, once)
            hipError_t e = hipSuccess;
            for (uint32_t i = 0; i < nb && e == hipSuccess; i++) {
                e = hipMemcpyAsync(w.d_wit + (size_t)i * p->wit_words, js[i].wires, p->wit_words * 8, hipMemcpyHostToDevice, w.ctx->stream);
                wires[i] = w.d_wit + (size_t)i * p->wit_words;
            }
            if (e == hipSuccess) e = hipStreamSynchronize(w.ctx->stream);
            if (e != hipSuccess) { for (uint32_t i = 0; i < nb; i++) { rcs[i] = QPGPU_EDEVICE; errs[i] = std::string("pool: witness upload: ") + hipGetErrorString(e); } finish(p, js, rcs, lens, errs); continue; }
        } else if (js[0].kind == JOB_PARTIAL) {
            // stage s1 for the whole batch: WormholeProver::commit's assignments -> generate_partial_witness on the device. The
            // last n_blinding cells of the list are drawn on the device; a job submitted without public inputs takes them out
            // of its witness afterwards (then the whole batch is generated that way and supplied ones are compared)
            const size_t all_cells = p->cells.size(), count = all_cells - p->n_blinding;
            bool derive = false;
            for (uint32_t i = 0; i < nb; i++) derive = derive || (p->num_pis && !js[i].public_inputs);
            std::vector<uint64_t> vals((size_t)nb * count), pv((size_t)nb * p->num_pis);
            for (uint32_t i = 0; i < nb; i++) {
                std::memcpy(vals.data() + (size_t)i * count, js[i].values.data(), count * 8);
                if (p->num_pis && js[i].public_inputs) std::memcpy(pv.data() + (size_t)i * p->num_pis, js[i].public_inputs, p->num_pis * 8);
                wipe(js[i].values);
                wires[i] = w.d_wit + (size_t)i * p->wit_words;
            }
            std::vector<int> st(nb, QPGPU_OK);
            auto generate = [&](const uint64_t *v, const uint64_t *pi, uint32_t batch, uint64_t *dst, int *status) {
                return p->n_blinding ? qpgpu_generate_witness_partial_batch_blinded_dev(w.circ, p->cells.data(), all_cells, p->n_blinding, v, nullptr, pi, batch, dst, status)
                                     : qpgpu_generate_witness_partial_batch_dev(w.circ, p->cells.data(), all_cells, v, pi, batch, dst, status);
            };
            int rc = generate(vals.data(), derive ? nullptr : pv.data(), nb, w.d_wit, st.data());
            if ((rc == QPGPU_OK || rc == QPGPU_EUNSAT) && derive) {
                std::vector<uint64_t> got((size_t)nb * p->num_pis);
                const int rr = qpgpu_witness_public_inputs_dev(w.circ, w.d_wit, nb, got.data());
                if (rr != QPGPU_OK) rc = rr;
                else
                    for (uint32_t i = 0; i < nb; i++) {
                        if (js[i].public_inputs && st[i] == QPGPU_OK && std::memcmp(got.data() + (size_t)i * p->num_pis, pv.data() + (size_t)i * p->num_pis, p->num_pis * 8) != 0) {
                            st[i] = QPGPU_EUNSAT; rc = QPGPU_EUNSAT;
                        }
                        std::memcpy(pv.data() + (size_t)i * p->num_pis, got.data() + (size_t)i * p->num_pis, p->num_pis * 8);
                    }
            }
            derived_pis.swap(pv);
            for (uint32_t i = 0; i < nb; i++) pis[i] = derived_pis.data() + (size_t)i * p->num_pis;
            if (rc != QPGPU_OK && rc != QPGPU_EUNSAT) {
                for (uint32_t i = 0; i < nb; i++) { rcs[i] = rc; errs[i] = qpgpu_last_error(w.ctx); }
                wipe(vals);
                finish(p, js, rcs, lens, errs);
                continue;
            }
            if (rc == QPGPU_EUNSAT) {
                // an unsatisfiable witness concerns its caller only: it gets its own message (one witness regenerated alone, in the
                // last slot of the workspace so that the others stay), the rest of the batch is proven
                for (uint32_t i = 0; i < nb; i++) {
                    if (st[i] == QPGPU_OK) continue;
                    rcs[i] = QPGPU_EUNSAT;
                    uint64_t *scratch = w.d_wit + (size_t)(p->max_batch - 1) * p->wit_words;
                    bool slot_free = true;
                    for (uint32_t k = 0; k < nb; k++) if (st[k] == QPGPU_OK && wires[k] == scratch) slot_free = false;
                    if (slot_free && nb > 1) {
                        const int r1 = generate(vals.data() + (size_t)i * count, js[i].public_inputs ? js[i].public_inputs : nullptr, 1, scratch, nullptr);
                        errs[i] = r1 == QPGPU_OK ? "witness generation: the public inputs handed in are not the ones the circuit computes" : std::string(qpgpu_last_error(w.ctx));
                    } else errs[i] = nb > 1 ? "witness generation: a target was set twice with different values" : std::string(qpgpu_last_error(w.ctx));
                }
            }
            wipe(vals);
        }
        // stages s2..s12 for the jobs that have a witness
        std::vector<uint32_t> live;
        for (uint32_t i = 0; i < nb; i++) if (rcs[i] == QPGPU_OK) live.push_back(i);
        if (!live.empty()) {
            const uint32_t nl = (uint32_t)live.size();
            std::vector<const uint64_t *> lw(nl), lp(nl);
            std::vector<uint8_t *> lo(nl);
            std::vector<size_t> ll(nl, 0);
            size_t cap = ~(size_t)0;
            for (uint32_t k = 0; k < nl; k++) { lw[k] = wires[live[k]]; lp[k] = pis[live[k]]; lo[k] = outs[live[k]]; cap = std::min(cap, js[live[k]].out_cap); }
            const int rc = qpgpu_prove_batch_dev(w.circ, lw.data(), nl, lp.data(), lo.data(), cap, ll.data());
            if (rc != QPGPU_OK && nl > 1) {
                // the jobs of a lockstep batch are unrelated callers' proofs, and in the reference a failing prove concerns its
                // caller only: prove them again one at a time, so that the offender alone gets the error (and its own text)
                for (uint32_t k = 0; k < nl; k++) {
                    ll[k] = 0;
                    rcs[live[k]] = qpgpu_prove_batch_dev(w.circ, &lw[k], 1, &lp[k], &lo[k], js[live[k]].out_cap, &ll[k]);
                    if (rcs[live[k]] != QPGPU_OK) errs[live[k]] = qpgpu_last_error(w.ctx);
                }
            } else if (rc != QPGPU_OK) { rcs[live[0]] = rc; errs[live[0]] = qpgpu_last_error(w.ctx); }
            for (uint32_t k = 0; k < nl; k++) lens[live[k]] = rcs[live[k]] == QPGPU_OK ? ll[k] : 0;
        }
        finish(p, js, rcs, lens, errs);
    }
}

int ensure_workspace(qpgpu_pool *p) {
    if (p->host_witness) return QPGPU_OK;
    for (Worker &w : p->workers) {
        if (hipSetDevice(w.device) != hipSuccess) return QPGPU_EDEVICE;
        if (hipMalloc((void **)&w.d_wit, p->wit_words * 8 * p->max_batch) != hipSuccess) { p->err = "pool: witness workspace allocation failed"; return QPGPU_EDEVICE; }
    }
    p->host_witness = true;
    return QPGPU_OK;
}
int submit(qpgpu_pool *p, Job &&j, uint64_t *ticket) {
    {
        std::lock_guard<std::mutex> lk(p->mu);
        if (p->next_ticket - p->oldest_live >= p->done.size()) { p->err = "pool_submit: too many unwaited jobs"; return QPGPU_EBUFSIZE; }
        const uint64_t t = p->next_ticket++;
        p->done[t % p->done.size()] = Done();
        j.ticket = t;
        p->queue.push_back(std::move(j));
        *ticket = t;
    }
    p->cv_job.notify_all();     // workers of other devices may not take a pinned job: wake them all, the right one picks it up
    return QPGPU_OK;
}
// a job is checked on its own at submit, so that a bad one cannot reach a lockstep batch of other callers' proofs
int check_job(qpgpu_pool *p, const void *witness, const uint64_t *public_inputs, const uint8_t *out, size_t out_cap, uint64_t *ticket, bool pis_optional = false) {
    if (!p || !ticket) return QPGPU_EINVAL;
    const size_t need = qpgpu_pool_proof_size(p);
    if (!witness || !out || (p->num_pis && !public_inputs && !pis_optional) || out_cap < need) {
        std::lock_guard<std::mutex> lk(p->mu);
        const bool small = out_cap < need && witness && out;
        p->err = small ? "pool_submit: output buffer smaller than the proof (" + std::to_string(out_cap) + " < " + std::to_string(need) + " bytes)" : "pool_submit: null argument";
        return small ? QPGPU_EBUFSIZE : QPGPU_EINVAL;
    }
    return QPGPU_OK;
}
}  // namespace

extern "C" {

void qpgpu_pool_destroy(qpgpu_pool *p) {
    if (!p) return;
    { std::lock_guard<std::mutex> lk(p->mu); p->stopping = true; }
    p->cv_job.notify_all();
    for (auto &t : p->threads) if (t.joinable()) t.join();
    for (Worker &w : p->workers) {
        (void)hipSetDevice(w.device);
        if (w.d_wit) {     // the workspace held witnesses: overwritten before release
            (void)hipMemsetAsync(w.d_wit, 0, p->wit_words * 8 * p->max_batch, w.ctx ? w.ctx->stream : nullptr);
            (void)hipStreamSynchronize(w.ctx ? w.ctx->stream : nullptr);
            (void)hipFree(w.d_wit);
        }
        if (w.circ) qpgpu_circuit_free(w.circ);
        if (w.ctx) qpgpu_ctx_destroy(w.ctx);
    }
    wipe(p->cells);
    delete p;
}

int qpgpu_pool_create(int device, const uint64_t *pack_words, size_t n_words, unsigned workers, qpgpu_pool **out) {
    return qpgpu_pool_create_batched(device, pack_words, n_words, workers, 1, out);
}
int qpgpu_pool_create_batched(int device, const uint64_t *pack_words, size_t n_words, unsigned workers, unsigned max_batch, qpgpu_pool **out) {
    return qpgpu_pool_create_multi(&device, 1, pack_words, n_words, workers, max_batch, 0, out);
}

int qpgpu_pool_create_multi(const int *devices, unsigned n_devices, const uint64_t *pack_words, size_t n_words, unsigned workers_per_device,
                            unsigned max_batch, unsigned flags, qpgpu_pool **out) {
    if (!out || !pack_words || !devices || n_devices == 0 || n_devices > 64 || workers_per_device == 0 || workers_per_device > 64 ||
        max_batch == 0 || max_batch > 1024 || (flags & ~QPGPU_POOL_HOST_WITNESS)) return QPGPU_EINVAL;
    *out = nullptr;
    qpgpu_pool *p = new qpgpu_pool();
    p->devices.assign(devices, devices + n_devices);
    p->max_batch = max_batch;
    p->done.resize(4096);
    {
        // a queue-intercepting profiler in the process? (rocprofv3 preloads librocprofiler-sdk-tool and sets ROCP_TOOL_LIBRARIES)
        const char *force = getenv("QPGPU_POOL_SERIALIZE");
        if (force && (force[0] == '0' || force[0] == '1')) p->serialize = force[0] == '1';
        else p->serialize = getenv("ROCP_TOOL_LIBRARIES") != nullptr || dlsym(RTLD_DEFAULT, "rocprofiler_configure") != nullptr;
    }
    p->workers.resize((size_t)n_devices * workers_per_device);
    for (size_t i = 0; i < p->workers.size(); i++) {
        Worker &w = p->workers[i];
        w.slot = (int)(i / workers_per_device); w.device = devices[w.slot];
        int rc = qpgpu_ctx_create(w.device, &w.ctx);
        if (rc == QPGPU_OK) rc = qpgpu_circuit_load_batch(w.ctx, pack_words, n_words, max_batch, &w.circ);
        if (rc != QPGPU_OK) { qpgpu_pool_destroy(p); return rc; }
    }
    p->num_pis = qpgpu_circuit_num_public_inputs(p->workers[0].circ);
    p->wit_words = (size_t)pack_words[2] << pack_words[1];          // num_wires << degree_bits (the load above validated the header)
    if ((flags & QPGPU_POOL_HOST_WITNESS) && ensure_workspace(p) != QPGPU_OK) { qpgpu_pool_destroy(p); return QPGPU_EDEVICE; }
    for (size_t i = 0; i < p->workers.size(); i++) p->threads.emplace_back(worker, p, i);
    *out = p;
    return QPGPU_OK;
}

size_t qpgpu_pool_proof_size(const qpgpu_pool *p) { return p && !p->workers.empty() ? qpgpu_proof_size(p->workers[0].circ) : 0; }
unsigned qpgpu_pool_workers(const qpgpu_pool *p) { return p ? (unsigned)p->workers.size() : 0; }
unsigned qpgpu_pool_devices(const qpgpu_pool *p) { return p ? (unsigned)p->devices.size() : 0; }
int qpgpu_pool_serialized(const qpgpu_pool *p) { return p && p->serialize ? 1 : 0; }
const char *qpgpu_pool_last_error(const qpgpu_pool *p) { return p ? p->err.c_str() : "null pool"; }

// qpgpu_circuit_set_witness_check for every worker; call while no job is queued or running
int qpgpu_pool_set_witness_check(qpgpu_pool *p, int on) {
    if (!p) return QPGPU_EINVAL;
    std::lock_guard<std::mutex> lk(p->mu);
    if (!p->queue.empty() || p->next_ticket != p->oldest_live) { p->err = "pool_set_witness_check: jobs in flight"; return QPGPU_EINVAL; }
    for (Worker &w : p->workers) qpgpu_circuit_set_witness_check(w.circ, on);
    return QPGPU_OK;
}

int qpgpu_pool_set_partial_cells(qpgpu_pool *p, const uint64_t *cells, size_t count) { return qpgpu_pool_set_partial_cells_blinded(p, cells, count, 0); }
int qpgpu_pool_set_partial_cells_blinded(qpgpu_pool *p, const uint64_t *cells, size_t count, size_t n_blinding) {
    if (!p || (count && !cells) || n_blinding > count) return QPGPU_EINVAL;
    std::lock_guard<std::mutex> lk(p->mu);
    if (!p->queue.empty() || p->next_ticket != p->oldest_live) { p->err = "pool_set_partial_cells: jobs in flight"; return QPGPU_EINVAL; }
    if (ensure_workspace(p) != QPGPU_OK) return QPGPU_EDEVICE;
    for (Worker &w : p->workers) {
        const int rc = qpgpu_witness_partial_prepare(w.circ, cells, count, p->max_batch);
        if (rc != QPGPU_OK) { p->err = qpgpu_last_error(w.ctx); return rc; }
    }
    p->cells.assign(cells, cells + count);
    p->n_blinding = n_blinding;
    return QPGPU_OK;
}

int qpgpu_pool_submit(qpgpu_pool *p, const uint64_t *d_wires, const uint64_t *public_inputs, uint8_t *out, size_t out_cap, uint64_t *ticket) {
    return qpgpu_pool_submit_on(p, 0, d_wires, public_inputs, out, out_cap, ticket);
}
int qpgpu_pool_submit_on(qpgpu_pool *p, unsigned device_index, const uint64_t *d_wires, const uint64_t *public_inputs, uint8_t *out, size_t out_cap, uint64_t *ticket) {
    const int rc = check_job(p, d_wires, public_inputs, out, out_cap, ticket);
    if (rc != QPGPU_OK) return rc;
    if (device_index >= p->devices.size()) { std::lock_guard<std::mutex> lk(p->mu); p->err = "pool_submit_on: no such device index"; return QPGPU_EINVAL; }
    return submit(p, Job{0, JOB_DEVICE, (int)device_index, d_wires, public_inputs, out, out_cap, {}}, ticket);
}
int qpgpu_pool_submit_host(qpgpu_pool *p, const uint64_t *wires, const uint64_t *public_inputs, uint8_t *out, size_t out_cap, uint64_t *ticket) {
    const int rc = check_job(p, wires, public_inputs, out, out_cap, ticket);
    if (rc != QPGPU_OK) return rc;
    if (!p->host_witness) { std::lock_guard<std::mutex> lk(p->mu); p->err = "pool_submit_host: the pool was created without QPGPU_POOL_HOST_WITNESS"; return QPGPU_EINVAL; }
    return submit(p, Job{0, JOB_HOST, -1, wires, public_inputs, out, out_cap, {}}, ticket);
}
int qpgpu_pool_submit_partial(qpgpu_pool *p, const uint64_t *values, const uint64_t *public_inputs, uint8_t *out, size_t out_cap, uint64_t *ticket) {
    const int rc = check_job(p, values, public_inputs, out, out_cap, ticket, true);
    if (rc != QPGPU_OK) return rc;
    if (p->cells.empty()) { std::lock_guard<std::mutex> lk(p->mu); p->err = "pool_submit_partial: no cell list (qpgpu_pool_set_partial_cells)"; return QPGPU_EINVAL; }
    Job j{0, JOB_PARTIAL, -1, nullptr, public_inputs, out, out_cap, std::vector<uint64_t>(values, values + (p->cells.size() - p->n_blinding))};
    return submit(p, std::move(j), ticket);
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
