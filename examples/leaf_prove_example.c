/*
 * leaf_prove_example.c — the leaf path from plain C, the way the Rust side would drive it ("eight GPUs from
 * Rust"): build the Wormhole leaf circuit (qpgpu_leaf_circuit_build), create ONE proving pool over a list of devices, and push
 * CircuitInputs through it: qpgpu_leaf_commit on the host (WormholeProver::commit, wormhole/prover/src/lib.rs:156-163), then
 * qpgpu_pool_submit_partial — stage s1 (generate_partial_witness) and stages s2..s12 on whichever device takes the job, the
 * proof written into the caller's host buffer. This is the reference bench's timed region, `prover.commit(&inputs).unwrap()
 * .prove()` (wormhole/prover/benches/prover.rs:38), on the reference bench's input (build_dummy_circuit_inputs,
 * wormhole/aggregator/src/dummy_proof.rs:125-170).
 *
 * It is also the torch-free driver of the profiled runs: nothing but libqpgpu.so and the ROCm runtime it links is loaded, so a
 * `rocprofv3 ... -- ./leaf_prove_example ...` process runs ONE ROCm stack; the resolved runtime libraries are
 * printed from /proc/self/maps.
 *
 *   gcc -O2 -I include examples/leaf_prove_example.c -L qp-zk-circuits_amd -lqpgpu -lpthread -Wl,-rpath,$PWD/qp-zk-circuits_amd -o /tmp/leaf_prove_example
 *   /tmp/leaf_prove_example [min_degree_bits=0] [devices=0,0] [workers_per_device=2] [lockstep=4] [steps=2]
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "qpgpu.h"
#include "qpgpu_leaf.h"
#include "qpgpu_verify.h"

static void hex32(const char *h, uint8_t out[32]) { for (int i = 0; i < 32; i++) { unsigned v; sscanf(h + 2 * i, "%2x", &v); out[i] = (uint8_t)v; } }
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

static void print_runtime_stack(void) {
    FILE *f = fopen("/proc/self/maps", "r");
    char line[1024], seen[8][512];
    int n = 0;
    if (!f) return;
    while (fgets(line, sizeof line, f)) {
        char *p = strchr(line, '/');
        if (!p || !(strstr(p, "libamdhip64") || strstr(p, "libhsa-runtime64") || strstr(p, "librocprofiler-sdk") || strstr(p, "libqpgpu"))) continue;
        p[strcspn(p, "\n")] = 0;
        int dup = 0;
        for (int i = 0; i < n; i++) if (!strcmp(seen[i], p)) dup = 1;
        if (!dup && n < 8) { snprintf(seen[n++], 512, "%s", p); printf("runtime: %s\n", p); }
    }
    fclose(f);
}

int main(int argc, char **argv) {
    const unsigned min_degree_bits = argc > 1 ? (unsigned)atoi(argv[1]) : 0;
    int devices[16]; unsigned n_devices = 0;
    { char buf[128]; snprintf(buf, sizeof buf, "%s", argc > 2 ? argv[2] : "0,0"); for (char *t = strtok(buf, ","); t && n_devices < 16; t = strtok(NULL, ",")) devices[n_devices++] = atoi(t); }
    const unsigned workers = argc > 3 ? (unsigned)atoi(argv[3]) : 2, lockstep = argc > 4 ? (unsigned)atoi(argv[4]) : 4, steps = argc > 5 ? (unsigned)atoi(argv[5]) : 2;
    const int hints = argc > 6 ? atoi(argv[6]) : 0;      /* 1: the front-end's hash hints ride along with commit's assignments (qpgpu_leaf.h) */
    char err[QPGPU_LEAF_ERR_CAP];

    /* WormholeCircuit::new(config).build_prover(): host only */
    size_t words = 0;
    uint64_t target_map[QPGPU_LT_COUNT], info[QPGPU_LEAF_CIRCUIT_INFO_WORDS];
    if (qpgpu_leaf_circuit_build(QPGPU_LEAF_FRAGMENT_FULL, min_degree_bits, 0, NULL, NULL, 0, &words, NULL, NULL, err)) { fprintf(stderr, "build: %s\n", err); return 1; }
    uint64_t *pack = malloc(words * 8);
    if (qpgpu_leaf_circuit_build(QPGPU_LEAF_FRAGMENT_FULL, min_degree_bits, 0, NULL, pack, words, &words, target_map, info, err)) { fprintf(stderr, "build: %s\n", err); return 1; }
    printf("leaf circuit: 2^%llu rows (%llu before padding: %llu Arithmetic, %llu BaseSum, %llu Poseidon2, %llu Poseidon), %zu pack words\n",
           (unsigned long long)info[0], (unsigned long long)info[1], (unsigned long long)info[7], (unsigned long long)info[8], (unsigned long long)info[9],
           (unsigned long long)info[10], words);

    /* the reference bench's input */
    qpgpu_leaf_inputs in;
    memset(&in, 0, sizeof in);
    in.volume_fee_bps = 10; in.transfer_count = 4; in.input_amount = 100;
    hex32("4c8587bd422e01d961acdc75e7d66f6761b7af7c9b1864a492f369c9d6724f05", in.secret);
    hex32("ae6e4ff0dca1ef5ede9dccc84365cecfab4e431c6f3086216bc3b819cdf0a893", in.state_root);
    if (qpgpu_leaf_unspendable_account(NULL, 0, in.secret, in.unspendable_account)) return 1;
    {   /* DEFAULT_DIGESTS[0] (wormhole/tests/test-helpers/src/lib.rs:242-248) */
        static const uint8_t head[] = {0x08, 0x06, 0x70, 0x6f, 0x77, 0x5f, 0x80, 0xe9, 0xb6, 0xb7, 0x6b, 0x9e, 0x01, 0x73, 0x13, 0xdb, 0x7e, 0xfd, 0x56, 0x1e, 0xd0, 0xb0, 0x46,
                                       0x15, 0x2d, 0xb4, 0xe5, 0x09, 0x3e, 0x5b, 0x04, 0x06, 0x35, 0xf5, 0x34, 0x30, 0x26, 0x7b, 0xe1, 0x05, 0x70, 0x6f, 0x77, 0x5f, 0x01, 0x01};
        memcpy(in.digest, head, sizeof head);
        in.digest[107] = 0x12; in.digest[108] = 0x4f; in.digest[109] = 0xe2;
    }
    if (qpgpu_leaf_check_constraints(&in, err)) { fprintf(stderr, "inputs: %s\n", err); return 1; }

    /* WormholeProver::commit */
    enum { MAX_ASSIGNMENTS = QPGPU_LT_COUNT + QPGPU_LEAF_HASH_HINTS };
    uint64_t cells[MAX_ASSIGNMENTS], values[MAX_ASSIGNMENTS], pis[QPGPU_LEAF_PUBLIC_INPUTS];
    size_t count = 0, n_hints = 0;
    if (qpgpu_leaf_commit(&in, target_map, cells, values, QPGPU_LT_COUNT, &count, pis, err)) { fprintf(stderr, "commit: %s\n", err); return 1; }
    if (hints) {    /* the hash chains' states, computed here on the host: stage s1 runs the 61 hash rows side by side and checks them */
        if (qpgpu_leaf_circuit_hash_hint_cells(min_degree_bits, 0, NULL, cells + count, QPGPU_LEAF_HASH_HINTS, &n_hints, err) ||
            qpgpu_leaf_hash_hints(&in, values + count, QPGPU_LEAF_HASH_HINTS, &n_hints, err)) { fprintf(stderr, "hash hints: %s\n", err); return 1; }
        count += n_hints;
    }

    /* one pool over all the devices; every worker resolves the cell list once */
    qpgpu_pool *pool = NULL;
    if (qpgpu_pool_create_multi(devices, n_devices, pack, words, workers, lockstep, 0, &pool)) { fprintf(stderr, "no gfx950 device (the library has no CPU fallback) or pool creation failed\n"); return 2; }
    print_runtime_stack();
    if (qpgpu_pool_serialized(pool)) printf("pool: a queue-intercepting profiler is loaded, the workers take turns on the device\n");
    if (qpgpu_pool_set_partial_cells(pool, cells, count)) { fprintf(stderr, "set_partial_cells: %s\n", qpgpu_pool_last_error(pool)); return 3; }
    const size_t cap = qpgpu_pool_proof_size(pool);
    const unsigned per_step = n_devices * workers * lockstep;
    uint8_t *outs = malloc((size_t)per_step * cap);
    uint64_t *tickets = malloc(per_step * sizeof(uint64_t));
    double t0 = 0;
    for (unsigned s = 0; s <= steps; s++) {             /* step 0 warms up */
        if (s == 1) t0 = now();
        for (unsigned i = 0; i < per_step; i++)
            if (qpgpu_pool_submit_partial(pool, values, pis, outs + (size_t)i * cap, cap, &tickets[i])) { fprintf(stderr, "submit: %s\n", qpgpu_pool_last_error(pool)); return 4; }
        for (unsigned i = 0; i < per_step; i++) {
            size_t len = 0;
            if (qpgpu_pool_wait(pool, tickets[i], &len) || len != cap) { fprintf(stderr, "proof %u: %s\n", i, qpgpu_pool_last_error(pool)); return 5; }
            if (memcmp(outs + (size_t)i * cap, outs, cap)) { fprintf(stderr, "proof %u differs from proof 0 of the same inputs\n", i); return 5; }
        }
    }
    const double dt = now() - t0;
    /* the proof's public inputs are the reference's 21, in its order (wormhole/inputs/src/lib.rs:68-80) */
    if (memcmp(outs + cap - 8 * QPGPU_LEAF_PUBLIC_INPUTS, pis, 8 * QPGPU_LEAF_PUBLIC_INPUTS)) { fprintf(stderr, "public inputs differ\n"); return 6; }
    /* a flipped secret byte: that job alone fails, naming the target; its neighbours are proven */
    {
        qpgpu_leaf_inputs bad = in;
        uint64_t bc[QPGPU_LT_COUNT], bv[MAX_ASSIGNMENTS], bp[QPGPU_LEAF_PUBLIC_INPUTS], t_bad, t_good[2];
        size_t bn = 0, len = 0;
        bad.secret[3] ^= 1;
        if (qpgpu_leaf_commit(&bad, target_map, bc, bv, QPGPU_LT_COUNT, &bn, bp, err)) return 7;
        if (hints && qpgpu_leaf_hash_hints(&bad, bv + bn, QPGPU_LEAF_HASH_HINTS, &n_hints, err)) return 7;      /* honest hints of dishonest inputs */
        if (qpgpu_pool_submit_partial(pool, values, pis, outs, cap, &t_good[0]) || qpgpu_pool_submit_partial(pool, bv, bp, outs + cap, cap, &t_bad) ||
            qpgpu_pool_submit_partial(pool, values, pis, outs + 2 * cap, cap, &t_good[1])) return 7;
        if (qpgpu_pool_wait(pool, t_good[0], &len) || len != cap) { fprintf(stderr, "neighbour failed: %s\n", qpgpu_pool_last_error(pool)); return 7; }
        if (qpgpu_pool_wait(pool, t_bad, &len) != QPGPU_EUNSAT || !strstr(qpgpu_pool_last_error(pool), "set twice with different values")) { fprintf(stderr, "unsatisfiable job not reported\n"); return 7; }
        printf("unsatisfiable job alone: %s\n", qpgpu_pool_last_error(pool));
        if (qpgpu_pool_wait(pool, t_good[1], &len) || len != cap || memcmp(outs + 2 * cap, outs, cap)) { fprintf(stderr, "neighbour failed: %s\n", qpgpu_pool_last_error(pool)); return 7; }
    }
    /* the library's host verifier accepts the proof (verifier data = this circuit's constants/sigmas cap) */
    {
        qpgpu_ctx *ctx = NULL; qpgpu_circuit *c = NULL; qpgpu_verifier *v = NULL;
        uint64_t cs_cap[4 << 4];
        char why[QPGPU_VERIFY_ERR_CAP];
        if (qpgpu_ctx_create(devices[0], &ctx) || qpgpu_circuit_load(ctx, pack, words, &c) || qpgpu_circuit_constants_sigmas_cap(c, cs_cap, 4 << 4)) return 8;
        if (qpgpu_verifier_create(pack, words, cs_cap, 4 << 4, 0, NULL, 0, &v, why) || qpgpu_verifier_verify(v, outs, cap, why)) { fprintf(stderr, "verifier: %s\n", why); return 8; }
        qpgpu_verifier_free(v); qpgpu_circuit_free(c); qpgpu_ctx_destroy(ctx);
    }
    qpgpu_pool_destroy(pool);          /* drains every device's workers */
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < cap; i++) h = (h ^ outs[i]) * 1099511628211ull;
    printf("ok devices=%u workers=%u lockstep=%u steps=%u proofs=%u proof_bytes=%zu fnv1a=%016llx commit+prove %.1f proofs/s%s\n", n_devices, workers, lockstep, steps,
           steps * per_step, cap, (unsigned long long)h, steps ? steps * per_step / dt : 0.0, hints ? " (hash hints)" : "");
    free(pack); free(outs); free(tickets);
    return 0;
}
