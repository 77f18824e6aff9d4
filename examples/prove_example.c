/*
 * prove_example.c — the C ABI used from plain C, the way a Rust `extern "C"` block would use it
 * (INTEGRATION.md): build a synthetic leaf-shaped circuit, load it, prove twice, check determinism.
 *
 *   gcc -O2 -I include examples/prove_example.c -L qp-zk-circuits_amd -lqpgpu -Wl,-rpath,$PWD/qp-zk-circuits_amd -o /tmp/prove_example
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "qpgpu.h"

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
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < cap; i++) h = (h ^ p1[i]) * 1099511628211ull;
    printf("ok degree_bits=%u proof_bytes=%zu fnv1a=%016llx\n", degree_bits, cap, (unsigned long long)h);
    qpgpu_circuit_free(circuit);
    qpgpu_ctx_destroy(ctx);
    free(pack); free(wires); free(p1); free(p2);
    return 0;
}
