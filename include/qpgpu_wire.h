/*
 * qpgpu_wire.h — on-disk / wire formats the reference's provers exchange (host only, no GPU), SURVEY.md section 8 row f4,
 * and the circuit-pack validator (row f1):
 *
 *   proof hand-off: hex::encode(proof.to_bytes())               wormhole/tests/src/aggregator/aggregator_tests.rs:350-394
 *   config.json (CircuitBinsConfig, serde_json pretty)           wormhole/aggregator/src/config.rs:20-88
 *     { "num_leaf_proofs": N, "num_private_batch_proofs": M | null }, legacy key "num_layer0_proofs" accepted on load,
 *     both counts in 1..=64 (qp_wormhole_inputs::MAX_PROOF_COUNT, wormhole/inputs/src/lib.rs:46-65)
 *   artifact file names of a bins directory                      wormhole/circuit-builder/src/lib.rs:78-80,
 *                                                                wormhole/aggregator/src/private_batch/circuit/build.rs:110-123,
 *                                                                wormhole/aggregator/src/public_batch/circuit/build.rs:118-119
 *     plus the one file this backend adds per level: the prover pack (csrc/circuit.hpp) the exporter writes next to them
 *   artifact size cap                                            wormhole/aggregator/src/common/utils.rs:33 (64 MiB; the
 *                                                                prover pack is exempt: it holds the sigma polynomials)
 */
#ifndef QPGPU_WIRE_H
#define QPGPU_WIRE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QPGPU_MAX_PROOF_COUNT 64
#define QPGPU_MAX_ARTIFACT_FILE_BYTES (64ull * 1024 * 1024)
#define QPGPU_WIRE_ERR_CAP 200
#define QPGPU_CONFIG_ERR_CAP 400     /* err buffers of the CircuitConfig policy functions (the reference's messages are long) */

/* ---- proof hex (the CLI hand-off format) ---- */
/* hex::encode: lowercase, no prefix; returns 2 * len, or 0 when out_cap < 2 * len + 1 (the output is NUL-terminated) */
size_t qpgpu_hex_encode(const uint8_t *in, size_t len, char *out, size_t out_cap);
/* hex::decode: upper or lower case, even length, nothing else; returns the byte count or (size_t)-1 */
size_t qpgpu_hex_decode(const char *in, size_t len, uint8_t *out, size_t out_cap);

/* ---- config.json ---- */
typedef struct {
    uint64_t num_leaf_proofs;
    int has_num_private_batch_proofs;      /* 0: null (private batch only) */
    uint64_t num_private_batch_proofs;
} qpgpu_bins_config;
/* CircuitBinsConfig::load on the file's text: parse + validate. Unknown keys are ignored (serde default); giving both the
 * current and the legacy key is a duplicate field. Returns 0, or -1 with the reason in err (QPGPU_WIRE_ERR_CAP bytes). */
int qpgpu_bins_config_parse(const char *json, size_t len, qpgpu_bins_config *out, char *err);
/* serde_json::to_string_pretty layout; returns the length (without the NUL) or 0 when out_cap is too small / invalid config */
size_t qpgpu_bins_config_write(const qpgpu_bins_config *cfg, char *out, size_t out_cap);
int qpgpu_bins_config_validate(const qpgpu_bins_config *cfg, char *err);

/* ---- artifact names ---- */
enum { QPGPU_LEVEL_LEAF = 0, QPGPU_LEVEL_PRIVATE_BATCH = 1, QPGPU_LEVEL_PUBLIC_BATCH = 2 };
enum { QPGPU_ARTIFACT_COMMON = 0, QPGPU_ARTIFACT_VERIFIER = 1, QPGPU_ARTIFACT_DUMMY_PROOF = 2, QPGPU_ARTIFACT_PROVER_PACK = 3, QPGPU_ARTIFACT_CONFIG = 4 };
/* e.g. (PRIVATE_BATCH, COMMON) -> "private_batch_common.bin"; NULL where the reference has no such file
 * (the public batch has no dummy template) */
const char *qpgpu_artifact_name(int level, int kind);

/* ---- circuit-pack validation with reasons (the loader's checks and the structural ones an exporter can get wrong) ----
 * Returns 0 when the pack would load and prove, else -1 with one line in err naming the first problem found:
 * bad magic / truncated sections / header fields out of range, unknown gate id, selector groups that overlap or leave a
 * gate out, a selector column value that names no gate of its group, sigma not a permutation of the routed cells (value
 * outside every wire coset, or two cells mapping to one), k_is not distinct cosets, FRI reduction schedule inconsistent
 * with degree_bits / cap_height, public-input cells outside the routed trace, hint cells outside the routed trace. */
int qpgpu_pack_validate(const uint64_t *pack_words, size_t n_words, char *err);

/* ---- CircuitConfig policy (common/src/circuit.rs:378-571) ----
 * plonky2's CircuitConfig as plain data, the three canonical Wormhole configs, the structural validation every Wormhole
 * circuit constructor applies before building (validate_circuit_config, with its messages), and the canonicality check of
 * the artifact loaders (ensure_config_is_canonical, aggregator/src/common/utils.rs:248-262) applied to a circuit pack's
 * header — what the exporter runs before handing a pack to the prover. */
typedef struct {
    uint64_t num_wires, num_routed_wires, num_constants, security_bits, num_challenges, max_quotient_degree_factor;
    int use_base_arithmetic_gate, zero_knowledge;
    uint64_t rate_bits, cap_height, proof_of_work_bits, num_query_rounds;     /* fri_config */
    uint64_t reduction_arity_bits, reduction_final_poly_bits;                  /* FriReductionStrategy::ConstantArityBits(4, 5) */
} qpgpu_circuit_config;
/* level: QPGPU_LEVEL_LEAF / _PUBLIC_BATCH = standard_recursion_config; _PRIVATE_BATCH = standard_recursion_zk_config with
 * 135 wires, 60 routed. Returns -1 for an unknown level. */
int qpgpu_wormhole_circuit_config(int level, qpgpu_circuit_config *out);
int qpgpu_validate_circuit_config(const qpgpu_circuit_config *cfg, char *err);     /* err: QPGPU_CONFIG_ERR_CAP bytes */
/* The config fields a pack header carries (wires, routed wires, challenges, quotient degree factor, rate, cap height, proof
 * of work, query rounds, zero knowledge) against the level's canonical config. */
int qpgpu_pack_config_is_canonical(const uint64_t *pack_words, size_t n_words, int level, char *err);   /* err: QPGPU_CONFIG_ERR_CAP bytes */

#ifdef __cplusplus
}
#endif
#endif
