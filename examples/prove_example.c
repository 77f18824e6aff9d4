/*
 * prove_example.c — the C ABI used from plain C, the way a Rust `extern "C"` block would use it
 * (INTEGRATION.md): build a synthetic leaf-shaped circuit, load it, prove twice, check determinism; then regenerate the
 * witness on the device from its free cells, push eight proofs through a four-worker proving pool, prove four in one
 * lockstep batch, and check a proof with the library's host verifier (verifier data = the circuit handle's cap).
 *
 *   gcc -O2 -I include examples/prove_example.c -L qp-zk-circuits_amd -lqpgpu -Wl,-rpath,$PWD/qp-zk-circuits_amd -o /tmp/prove_example
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "qpgpu.h"
#include "qpgpu_verify.h"

#define CHECK(call) do { int rc_ = (call); if (rc_) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, ctx ? qpgpu_last_error(ctx) : "?"); return 1; } } while (0)

int main(int argc, char **argv) {
    unsigned degree_bits = argc > 1 ? (unsigned)atoi(argv[1]) : 10;
    const unsigned num_wires = 135, num_routed = 80, num_pis = 21, flags = 3; /* Poseidon + BaseSum rows */
    qpgpu_ctx *ctx = NULL;
    if (qpgpu_ctx_create(0, &ctx)) { fprintf(stderr, "no gfx950 device: the library has no CPU fallback\n"); return 2; }

    size_t words = qpgpu_synth_pack_words_ex(degree_bits, num_wires, num_routed, flags), got = 0;
    uint64_t *pack = malloc(words * 8), *wires = malloc((size_t)num_wires * 8 << degree_bits), pis[21];
    CHECK(qpgpu_synth_circuit_ex(degree_bits, num_wires, num_routed, num_pis, 42, flags, pack, words, &got, wires, pis));

    qpgpu_circuit *circuit = NULL;
    CHECK(qpgpu_circuit_load(ctx, pack, got, &circuit));
    size_t cap = qpgpu_proof_size(circuit), len1 = 0, len2 = 0;
    uint8_t *p1 = malloc(cap), *p2 = malloc(cap);
    CHECK(qpgpu_prove(circuit, wires, pis, p1, cap, &len1));
    CHECK(qpgpu_prove(circuit, wires, pis, p2, cap, &len2));
    if (len1 != cap || len2 != cap || memcmp(p1, p2, cap)) { fprintf(stderr, "non-deterministic proof\n"); return 3; }
    /* a too-small buffer is an error, not an overrun */
    if (qpgpu_prove(circuit, wires, pis, p2, cap - 1, &len2) != QPGPU_EBUFSIZE) { fprintf(stderr, "missing EBUFSIZE\n"); return 4; }
    /* stage s1 on the device: keep only the caller-supplied cells, regenerate the rest, same proof */
    {
        const size_t cells = (size_t)num_wires << degree_bits;
        uint8_t *mask = malloc(cells);
        uint64_t *partial = malloc(cells * 8);
        CHECK(qpgpu_witness_free_mask(circuit, mask, cells));
        for (size_t i = 0; i < cells; i++) partial[i] = mask[i] ? wires[i] : 0;
        CHECK(qpgpu_generate_witness(circuit, partial, pis));
        if (memcmp(partial, wires, cells * 8)) { fprintf(stderr, "generated witness differs\n"); return 5; }
        free(mask); free(partial);
    }
    /* throughput: eight proofs through a pool of four workers (own streams and circuit copies inside the library) */
    {
        qpgpu_pool *pool = NULL;
        void *d_wires = NULL;
        const size_t bytes = (size_t)num_wires * 8 << degree_bits;
        CHECK(qpgpu_malloc(ctx, bytes, &d_wires));
        CHECK(qpgpu_memcpy_h2d(ctx, d_wires, wires, bytes));
        if (qpgpu_pool_create(0, pack, got, 4, &pool)) { fprintf(stderr, "pool_create failed\n"); return 6; }
        uint8_t *outs = malloc(8 * cap);
        uint64_t tickets[8];
        for (int i = 0; i < 8; i++)
            if (qpgpu_pool_submit(pool, d_wires, pis, outs + (size_t)i * cap, cap, &tickets[i])) { fprintf(stderr, "submit: %s\n", qpgpu_pool_last_error(pool)); return 6; }
        for (int i = 0; i < 8; i++) {
            size_t len = 0;
            if (qpgpu_pool_wait(pool, tickets[i], &len) || len != cap || memcmp(outs + (size_t)i * cap, p1, cap)) { fprintf(stderr, "pool proof %d differs: %s\n", i, qpgpu_pool_last_error(pool)); return 7; }
        }
        qpgpu_pool_destroy(pool);
        CHECK(qpgpu_free(ctx, d_wires));
        free(outs);
    }
    {   /* four proofs of one circuit in lockstep (every stage launched once for the batch), then host verification */
        qpgpu_circuit *batch = NULL;
        void *d_w = NULL;
        size_t bytes = (size_t)num_wires * ((size_t)1 << degree_bits) * 8;
        CHECK(qpgpu_circuit_load_batch(ctx, pack, got, 4, &batch));
        CHECK(qpgpu_malloc(ctx, bytes, &d_w));
        CHECK(qpgpu_memcpy_h2d(ctx, d_w, wires, bytes));
        const uint64_t *ws[4] = {d_w, d_w, d_w, d_w}, *ps[4] = {pis, pis, pis, pis};
        uint8_t *outs = malloc(4 * cap), *op[4];
        size_t lens[4];
        for (int i = 0; i < 4; i++) op[i] = outs + (size_t)i * cap;
        CHECK(qpgpu_prove_batch_dev(batch, ws, 4, ps, op, cap, lens));
        for (int i = 0; i < 4; i++) if (lens[i] != cap || memcmp(op[i], p1, cap)) { fprintf(stderr, "batch proof %d differs\n", i); return 8; }
        uint64_t cs_cap[4 << 4];
        char why[QPGPU_VERIFY_ERR_CAP];
        qpgpu_verifier *v = NULL;
        CHECK(qpgpu_circuit_constants_sigmas_cap(batch, cs_cap, 4 << 4));
        if (qpgpu_verifier_create(pack, got, cs_cap, 4 << 4, 0, NULL, 0, &v, why)) { fprintf(stderr, "verifier: %s\n", why); return 9; }
        if (qpgpu_verifier_verify(v, p1, cap, why)) { fprintf(stderr, "own proof rejected: %s\n", why); return 9; }
        p2[cap / 3] ^= 1;
        if (qpgpu_verifier_verify(v, p2, cap, why) != QPGPU_EVERIFY) { fprintf(stderr, "tampered proof accepted\n"); return 9; }
        qpgpu_verifier_free(v);
        CHECK(qpgpu_free(ctx, d_w));
        qpgpu_circuit_free(batch);
        free(outs);
    }
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < cap; i++) h = (h ^ p1[i]) * 1099511628211ull;
    printf("ok degree_bits=%u proof_bytes=%zu fnv1a=%016llx\n", degree_bits, cap, (unsigned long long)h);
    qpgpu_circuit_free(circuit);
    qpgpu_ctx_destroy(ctx);
    free(pack); free(wires); free(p1); free(p2);
    return 0;
}
