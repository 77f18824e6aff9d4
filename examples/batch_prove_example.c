/*
 * batch_prove_example.c — the two aggregation layers from plain C, the way the Rust side would drive them (INTEGRATION.md
 * section 2l): CircuitInputs -> two leaf proofs (one real spend, one dummy) -> PrivateBatchProver::commit / prove
 * (wormhole/aggregator/src/private_batch/prover/lib.rs:244-343) -> PublicBatchProver::commit / prove
 * (public_batch/prover/lib.rs:268-305), every circuit built by the library (qpgpu_leaf_circuit_build,
 * qpgpu_wrapper_circuit_build with the complete in-circuit verifier, the layer's logic, the private layer zero-knowledge), every
 * witness generated and every proof made on the device, every proof checked by the host verifier. Nothing but libqpgpu.so.
 *
 *   gcc -O2 -I include examples/batch_prove_example.c -L qp-zk-circuits_amd -lqpgpu -lpthread -Wl,-rpath,$PWD/qp-zk-circuits_amd -o /tmp/batch_prove_example
 *   /tmp/batch_prove_example
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "qpgpu.h"
#include "qpgpu_batch.h"
#include "qpgpu_leaf.h"
#include "qpgpu_verify.h"

#define N_LEAF 2u
#define CHECK(x, what) do { if (x) { fprintf(stderr, "%s failed: %s\n", what, err[0] ? err : (ctx ? qpgpu_last_error(ctx) : "")); return 1; } } while (0)

static void hex32(const char *h, uint8_t out[32]) { for (int i = 0; i < 32; i++) { unsigned v; sscanf(h + 2 * i, "%2x", &v); out[i] = (uint8_t)v; } }

/* the reference bench's input: build_dummy_circuit_inputs (wormhole/aggregator/src/dummy_proof.rs:125-170) */
static void dummy_inputs(qpgpu_leaf_inputs *in) {
    static const uint8_t head[] = {0x08, 0x06, 0x70, 0x6f, 0x77, 0x5f, 0x80, 0xe9, 0xb6, 0xb7, 0x6b, 0x9e, 0x01, 0x73, 0x13, 0xdb, 0x7e, 0xfd, 0x56, 0x1e, 0xd0, 0xb0, 0x46,
                                   0x15, 0x2d, 0xb4, 0xe5, 0x09, 0x3e, 0x5b, 0x04, 0x06, 0x35, 0xf5, 0x34, 0x30, 0x26, 0x7b, 0xe1, 0x05, 0x70, 0x6f, 0x77, 0x5f, 0x01, 0x01};
    memset(in, 0, sizeof *in);
    in->volume_fee_bps = 10; in->transfer_count = 4; in->input_amount = 100;
    hex32("4c8587bd422e01d961acdc75e7d66f6761b7af7c9b1864a492f369c9d6724f05", in->secret);
    hex32("ae6e4ff0dca1ef5ede9dccc84365cecfab4e431c6f3086216bc3b819cdf0a893", in->state_root);
    qpgpu_leaf_unspendable_account(NULL, 0, in->secret, in->unspendable_account);
    memcpy(in->digest, head, sizeof head);
    in->digest[107] = 0x12; in->digest[108] = 0x4f; in->digest[109] = 0xe2;
}

/* a spend that is not a dummy: its leaf in a one-level tree, a header committing to the tree's root, nullifier and block hash as the
 * circuit recomputes them */
static int real_inputs(qpgpu_leaf_inputs *in, char *err) {
    uint8_t leaf[32], sib[3][32], sorted[3][32];
    dummy_inputs(in);
    in->secret[0] ^= 0x5a;
    in->transfer_count = 7; in->input_amount = 300; in->output_amount_1 = 200; in->output_amount_2 = 97;     /* (200 + 97) * 10000 <= 300 * 9990 */
    memset(in->exit_account_1, 4, 32); memset(in->exit_account_2, 7, 32);
    for (int i = 0; i < 32; i += 8) { in->exit_account_1[i + 7] = 0; in->exit_account_2[i + 7] = 0; }
    if (qpgpu_leaf_unspendable_account(NULL, 0, in->secret, in->unspendable_account) || qpgpu_leaf_nullifier(NULL, 0, in->secret, in->transfer_count, in->nullifier)) return -1;
    if (qpgpu_zk_leaf_hash(in->unspendable_account, in->transfer_count, in->asset_id, in->input_amount, leaf)) return -1;
    for (int s = 0; s < 3; s++) for (int i = 0; i < 32; i++) sib[s][i] = (i % 8 == 7) ? 0 : (uint8_t)(17 * s + 3 * i + 1);
    in->zk_merkle_depth = 1;
    if (qpgpu_zk_proof_from_unsorted(leaf, &sib[0][0], 1, &sorted[0][0], in->zk_merkle_positions, in->zk_tree_root, err)) return -1;
    memcpy(in->zk_merkle_siblings[0], sorted, sizeof sorted);
    in->block_number = 2;
    if (qpgpu_leaf_block_hash(NULL, 0, in->parent_hash, in->block_number, in->state_root, in->extrinsics_root, in->zk_tree_root, in->digest, in->block_hash)) return -1;
    return qpgpu_leaf_check_constraints(in, err);
}

/* one circuit on the device with its verifier data */
typedef struct { uint64_t *pack; size_t words; qpgpu_circuit *circ; qpgpu_verifier *ver; uint64_t cs_cap[4 << 4]; size_t proof_size, wit_words; uint64_t *d_wires; } level_t;

static int level_load(qpgpu_ctx *ctx, level_t *l, char *err) {
    if (qpgpu_circuit_load(ctx, l->pack, l->words, &l->circ) || qpgpu_circuit_constants_sigmas_cap(l->circ, l->cs_cap, 4 << 4)) return -1;
    if (qpgpu_verifier_create(l->pack, l->words, l->cs_cap, 4 << 4, 0, NULL, 0, &l->ver, err)) return -1;
    l->proof_size = qpgpu_proof_size(l->circ);
    l->wit_words = (size_t)l->pack[2] << l->pack[1];
    return qpgpu_malloc(ctx, l->wit_words * 8, (void **)&l->d_wires);
}

/* a wrapper level: circuit + the wire cell of every logical target (+ the blinding cells behind them) */
typedef struct { level_t l; uint64_t *map; size_t map_count, T, Q, n_blind; unsigned n; } wrapper_t;

static int wrapper_build(qpgpu_ctx *ctx, const level_t *inner, unsigned n, unsigned routed, unsigned flags, wrapper_t *w, char *err) {
    uint64_t info[QPGPU_WRAPPER_CIRCUIT_INFO_WORDS];
    memset(w, 0, sizeof *w);
    w->n = n;
    if (qpgpu_wrapper_circuit_build(inner->pack, inner->words, inner->cs_cap, 4 << 4, n, routed, 0, 0, flags, NULL, 0, &w->l.words, NULL, 0, &w->map_count, NULL, err)) return -1;
    w->l.pack = malloc(w->l.words * 8); w->map = malloc(w->map_count * 8);
    if (qpgpu_wrapper_circuit_build(inner->pack, inner->words, inner->cs_cap, 4 << 4, n, routed, 0, 0, flags, w->l.pack, w->l.words, &w->l.words, w->map, w->map_count,
                                    &w->map_count, info, err)) return -1;
    w->T = (size_t)info[2]; w->Q = (size_t)info[3];
    w->n_blind = w->map_count - (size_t)n * (w->T + 4 + w->Q);
    printf("  wrapper over %u proofs: 2^%llu rows (%llu before padding, %llu blinding), %llu public inputs\n", n, (unsigned long long)info[0], (unsigned long long)info[1],
           (unsigned long long)info[11], (unsigned long long)info[9]);
    return level_load(ctx, &w->l, err);
}

/* fill_*_batch_witness -> stage s1 (blinding wires drawn on the device, public inputs computed by the circuit) -> proof */
static int wrapper_prove(qpgpu_ctx *ctx, const level_t *inner, wrapper_t *w, const uint8_t *const *proofs, const uint64_t *preimages, uint8_t *out, uint64_t *pis_out, char *err) {
    const size_t logical = (size_t)w->n * (w->T + 4), npis = (size_t)w->l.pack[9];
    size_t lens[64], cnt = 0, out_len = 0;
    uint32_t *ids = malloc(logical * 4);
    uint64_t *vals = malloc(logical * 8), *cells = malloc((logical + w->n_blind) * 8), *cvals = malloc(logical * 8);
    int rc = -1, status = 0;
    for (unsigned i = 0; i < w->n; i++) lens[i] = inner->proof_size;
    if (qpgpu_batch_fill_proof_targets(inner->pack, inner->words, proofs, lens, w->n, w->n, preimages, w->n, w->n, "inner proof", ids, vals, logical, &cnt, err)) goto done;
    const size_t k = qpgpu_leaf_map_targets(ids, vals, cnt, w->map, (size_t)w->n * (w->T + 4 + w->Q), cells, cvals);
    memcpy(cells + k, w->map + (size_t)w->n * (w->T + 4 + w->Q), w->n_blind * 8);          /* the blinding cells: no values, drawn on the device */
    if (qpgpu_generate_witness_partial_batch_blinded_dev(w->l.circ, cells, k + w->n_blind, w->n_blind, cvals, NULL, NULL, 1, w->l.d_wires, &status)) {
        snprintf(err, QPGPU_BATCH_ERR_CAP, "%s", qpgpu_last_error(ctx)); goto done;
    }
    if (qpgpu_witness_public_inputs_dev(w->l.circ, w->l.d_wires, 1, pis_out) || qpgpu_prove_dev(w->l.circ, w->l.d_wires, pis_out, out, w->l.proof_size, &out_len)) {
        snprintf(err, QPGPU_BATCH_ERR_CAP, "%s", qpgpu_last_error(ctx)); goto done;
    }
    rc = qpgpu_verifier_verify(w->l.ver, out, out_len, err);
    (void)npis;
done:
    free(ids); free(vals); free(cells); free(cvals);
    return rc;
}

int main(void) {
    char err[QPGPU_BATCH_ERR_CAP] = "";
    qpgpu_ctx *ctx = NULL;
    if (qpgpu_ctx_create(0, &ctx)) { fprintf(stderr, "no gfx950 device (the library has no CPU fallback)\n"); return 2; }

    /* ---- leaf level: WormholeCircuit::new, two proofs from CircuitInputs ---- */
    level_t leaf;
    uint64_t target_map[QPGPU_LT_COUNT];
    memset(&leaf, 0, sizeof leaf);
    CHECK(qpgpu_leaf_circuit_build(QPGPU_LEAF_FRAGMENT_FULL, 0, 0, NULL, NULL, 0, &leaf.words, NULL, NULL, err), "leaf circuit");
    leaf.pack = malloc(leaf.words * 8);
    CHECK(qpgpu_leaf_circuit_build(QPGPU_LEAF_FRAGMENT_FULL, 0, 0, NULL, leaf.pack, leaf.words, &leaf.words, target_map, NULL, err), "leaf circuit");
    CHECK(level_load(ctx, &leaf, err), "leaf load");
    qpgpu_leaf_inputs in[N_LEAF];
    CHECK(real_inputs(&in[0], err), "real inputs");
    dummy_inputs(&in[1]);
    uint8_t *leaf_proofs = malloc(N_LEAF * leaf.proof_size);
    uint64_t leaf_pis[N_LEAF][QPGPU_LEAF_PUBLIC_INPUTS];
    for (unsigned i = 0; i < N_LEAF; i++) {
        uint64_t cells[QPGPU_LT_COUNT], values[QPGPU_LT_COUNT];
        size_t count = 0, len = 0;
        CHECK(qpgpu_leaf_commit(&in[i], target_map, cells, values, QPGPU_LT_COUNT, &count, leaf_pis[i], err), "commit");
        CHECK(qpgpu_generate_witness_partial_dev(leaf.circ, cells, values, count, leaf_pis[i], leaf.d_wires), "leaf witness");
        CHECK(qpgpu_prove_dev(leaf.circ, leaf.d_wires, leaf_pis[i], leaf_proofs + i * leaf.proof_size, leaf.proof_size, &len), "leaf prove");
        CHECK(qpgpu_verifier_verify(leaf.ver, leaf_proofs + i * leaf.proof_size, len, err), "leaf verify");
    }
    printf("leaf proofs: %u x %zu bytes\n", N_LEAF, leaf.proof_size);

    /* ---- private batch: admission checks, padding + shuffle + preimages, proof ---- */
    wrapper_t priv;
    CHECK(wrapper_build(ctx, &leaf, N_LEAF, 60, QPGPU_WRAPPER_TRANSCRIPT | QPGPU_WRAPPER_VERIFY | QPGPU_WRAPPER_PRIVATE_BATCH | QPGPU_WRAPPER_ZERO_KNOWLEDGE, &priv, err), "private-batch circuit");
    CHECK(qpgpu_private_batch_preflight(&leaf_pis[0][0], 1, N_LEAF, err), "preflight");          /* one real proof supplied; the dummy is the padding template */
    uint32_t slot_source[N_LEAF];
    uint64_t preimages[4 * N_LEAF];
    CHECK(qpgpu_private_batch_arrange(1, N_LEAF, NULL, slot_source, preimages, err), "arrange");
    const uint8_t *slots[N_LEAF];
    uint64_t slot_rows[N_LEAF][QPGPU_LEAF_PUBLIC_INPUTS];
    for (unsigned s = 0; s < N_LEAF; s++) {
        const unsigned src = slot_source[s] == UINT32_MAX ? 1u : slot_source[s];
        slots[s] = leaf_proofs + src * leaf.proof_size;
        memcpy(slot_rows[s], leaf_pis[src], sizeof slot_rows[s]);
    }
    uint8_t *pb = malloc(priv.l.proof_size);
    const size_t n1 = qpgpu_private_batch_pi_len(N_LEAF);
    uint64_t *pis1 = malloc(n1 * 8), *want1 = malloc(n1 * 8);
    CHECK(wrapper_prove(ctx, &leaf, &priv, slots, preimages, pb, pis1, err), "private-batch prove");
    CHECK(qpgpu_private_batch_outputs(&slot_rows[0][0], N_LEAF, preimages, want1, err), "private-batch outputs");
    if (memcmp(pis1, want1, n1 * 8)) { fprintf(stderr, "the circuit's public inputs differ from the host restatement's\n"); return 3; }
    {
        qpgpu_private_batch_public_inputs hdr; qpgpu_exit_slot ex[2 * N_LEAF]; uint8_t nul[32 * N_LEAF];
        CHECK(qpgpu_private_batch_public_inputs_parse(pis1, n1, &hdr, ex, nul, err), "parse");
        unsigned paid = 0, sum = 0;
        for (unsigned i = 0; i < 2 * N_LEAF; i++) if (ex[i].summed_output_amount) { paid++; sum += ex[i].summed_output_amount; }
        printf("private batch: %zu-byte proof, dummy in slot %u, block number %u, %u paid exit slots, %u paid out\n", priv.l.proof_size, slot_source[0] == UINT32_MAX ? 0u : 1u,
               hdr.block_number, paid, sum);
        if (paid != 2 || sum != 297 || hdr.block_number != 2 || memcmp(hdr.block_hash, in[0].block_hash, 32)) { fprintf(stderr, "unexpected private-batch public inputs\n"); return 3; }
    }
    /* a tampered leaf proof: no witness */
    {
        uint8_t *bad = malloc(leaf.proof_size);
        const uint8_t *bs[N_LEAF] = {slots[0], bad};
        memcpy(bad, slots[1], leaf.proof_size); bad[leaf.proof_size / 2] ^= 1;
        if (wrapper_prove(ctx, &leaf, &priv, bs, preimages, pb + 0, want1, err) == 0 || !strstr(err, "set twice with different values")) { fprintf(stderr, "tampered inner proof not refused (%s)\n", err); return 4; }
        printf("tampered leaf proof: %s\n", err);
        free(bad);
        CHECK(wrapper_prove(ctx, &leaf, &priv, slots, preimages, pb, pis1, err), "private-batch prove");
    }

    /* ---- public batch over the one private-batch proof (M = 1), bound to an aggregator address ---- */
    wrapper_t pub;
    CHECK(wrapper_build(ctx, &priv.l, 1, 80, QPGPU_WRAPPER_TRANSCRIPT | QPGPU_WRAPPER_VERIFY | QPGPU_WRAPPER_PUBLIC_BATCH, &pub, err), "public-batch circuit");
    CHECK(qpgpu_public_batch_preflight(pis1, 1, n1, 1, err), "public preflight");
    uint8_t address[32];
    memset(address, 3, 32);
    uint64_t addr_felts[4];
    qpgpu_bytes_to_digest(address, addr_felts);
    const uint8_t *inner1[1] = {pb};
    uint8_t *root = malloc(pub.l.proof_size);
    const size_t n2 = qpgpu_public_batch_pi_len(1, N_LEAF);
    uint64_t *pis2 = malloc(n2 * 8), *want2 = malloc(n2 * 8);
    CHECK(wrapper_prove(ctx, &priv.l, &pub, inner1, addr_felts, root, pis2, err), "public-batch prove");          /* the aggregator address rides in the first preimage slot */
    CHECK(qpgpu_public_batch_outputs(pis1, 1, N_LEAF, address, want2, err), "public-batch outputs");
    if (memcmp(pis2, want2, n2 * 8)) { fprintf(stderr, "the public-batch circuit's public inputs differ from the host restatement's\n"); return 5; }
    {
        qpgpu_public_batch_public_inputs hdr; qpgpu_exit_slot ex[2 * N_LEAF]; uint8_t nul[32 * N_LEAF];
        CHECK(qpgpu_public_batch_public_inputs_parse(pis2, n2, 1, N_LEAF, &hdr, ex, nul, err), "parse");
        if (memcmp(hdr.aggregator_address, address, 32) || hdr.total_exit_slots != 2 * N_LEAF || hdr.block_number != 2) { fprintf(stderr, "unexpected public-batch public inputs\n"); return 5; }
        printf("public batch: %zu-byte proof bound to aggregator 03..03, %u exit slots, block number %u\n", pub.l.proof_size, hdr.total_exit_slots, hdr.block_number);
    }
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n2 * 8; i++) h = (h ^ ((const uint8_t *)pis2)[i]) * 1099511628211ull;
    printf("ok leaves=%u private_batch_pis=%zu public_batch_pis=%zu fnv1a(public inputs)=%016llx\n", N_LEAF, n1, n2, (unsigned long long)h);
    return 0;
}
