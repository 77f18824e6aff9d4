/*
 * qpgpu_batch.h — the host side of the two aggregation levels, on either side of `prove()` (host only, no GPU):
 * SURVEY.md section 8 rows a3 / a4 and the public-input formats of row f4.
 *
 *   public-input layouts and parsers                    wormhole/inputs/src/lib.rs:25-33,68-80,187-290,365-700
 *     leaf (21 felts), private batch (21 N + 8), public batch (12 + 14 M N)
 *   admission checks of PrivateBatchProver::commit      wormhole/aggregator/src/private_batch/prover/lib.rs:244-343,372-460
 *   dummy leaf template sentinel                        wormhole/aggregator/src/private_batch/prover/lib.rs:478-530
 *   padding, uniform shuffle, dummy-nullifier preimages wormhole/aggregator/src/private_batch/prover/lib.rs:296-316,537-541,
 *                                                       wormhole/aggregator/src/dummy_proof.rs:178-187
 *   admission checks of PublicBatchProver::commit       wormhole/aggregator/src/public_batch/prover/lib.rs:268-300,321-445
 *   dummy private-batch template sentinel               wormhole/aggregator/src/public_batch/prover/lib.rs:454-510
 *   what the two wrapper circuits output, natively      wormhole/aggregator/src/private_batch/circuit/circuit_logic.rs:170-523,
 *                                                       wormhole/aggregator/src/public_batch/circuit/circuit_logic.rs:167-330
 *     (first non-dummy slot as block / fee reference, dummy exits masked to zero, exit accounts grouped with the first
 *     occurrence carrying the sum and later ones zeroed, real nullifiers pairwise distinct, dummy nullifiers replaced by
 *     H(H(preimage)) under Poseidon2, the nullifier region sorted, zero padding to 21 N + 8)
 *
 * Cryptographic verification of the inner proofs — the other half of the reference's admission checks — needs a verifier
 * and is not part of this header: callers run theirs between `*_preflight` and proving.
 *
 * Every function returns 0, or -1 with one line in err (QPGPU_BATCH_ERR_CAP bytes, the reference's message where it has
 * one), or -4 (QPGPU_EUNSAT) from the `*_outputs` functions when the inputs violate a constraint of the wrapper circuit
 * (the proof could not be generated).
 */
#ifndef QPGPU_BATCH_H
#define QPGPU_BATCH_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QPGPU_LEAF_PI_LEN 21
#define QPGPU_BATCH_HEADER_LEN 8
#define QPGPU_PUBLIC_BATCH_HEADER_LEN 12
#define QPGPU_EXIT_SLOT_LEN 5
#define QPGPU_BATCH_MAX_PROOFS 64
#define QPGPU_BATCH_ERR_CAP 400

/* leaf public inputs: asset_id, output_amount_1, output_amount_2, volume_fee_bps, nullifier(4), exit_account_1(4),
 * exit_account_2(4), block_hash(4), block_number */
typedef struct {
    uint32_t asset_id, output_amount_1, output_amount_2, volume_fee_bps;
    uint8_t nullifier[32], exit_account_1[32], exit_account_2[32], block_hash[32];
    uint32_t block_number;
} qpgpu_leaf_public_inputs;

typedef struct {
    uint32_t summed_output_amount;
    uint8_t exit_account[32];
} qpgpu_exit_slot;

/* header of a private-batch proof's public inputs; account_data has 2 * n_leaf entries, nullifiers n_leaf */
typedef struct {
    uint32_t num_exit_slots, asset_id, volume_fee_bps;
    uint8_t block_hash[32];
    uint32_t block_number;
    uint32_t n_leaf;
} qpgpu_private_batch_public_inputs;

/* header of a public-batch proof's public inputs; account_data has total_exit_slots entries, nullifiers M * N */
typedef struct {
    uint8_t aggregator_address[32];
    uint32_t asset_id, volume_fee_bps;
    uint8_t block_hash[32];
    uint32_t block_number;
    uint32_t total_exit_slots;
} qpgpu_public_batch_public_inputs;

/* validate_proof_count: 1..=64 */
int qpgpu_validate_proof_count(uint64_t count, const char *label, char *err);
size_t qpgpu_private_batch_pi_len(size_t n_leaf);                              /* 21 N + 8 */
/* 12 + M * 2N * 5 + M * N * 4; 0 when a count is outside 1..=64 */
size_t qpgpu_public_batch_pi_len(size_t num_private_batch_proofs, size_t num_leaf_proofs);

/* PublicCircuitInputs::try_from_u64_slice */
int qpgpu_leaf_public_inputs_parse(const uint64_t *pis, size_t n, qpgpu_leaf_public_inputs *out, char *err);
/* PrivateBatchPublicInputs::try_from_u64_slice; slots: 2 * n_leaf entries (cap 128), nullifiers: n_leaf * 32 bytes (cap 64) */
int qpgpu_private_batch_public_inputs_parse(const uint64_t *pis, size_t n, qpgpu_private_batch_public_inputs *out,
                                            qpgpu_exit_slot *slots, uint8_t *nullifiers, char *err);
/* PublicBatchPublicInputs::try_from_u64_slice; slots: M * 2N entries, nullifiers: M * N * 32 bytes */
int qpgpu_public_batch_public_inputs_parse(const uint64_t *pis, size_t n, uint64_t num_private_batch_proofs, uint64_t num_leaf_proofs,
                                           qpgpu_public_batch_public_inputs *out, qpgpu_exit_slot *slots, uint8_t *nullifiers,
                                           char *err);

/* ---- private batch (client side) ---- */
/* PrivateBatchProver::commit's checks on the supplied leaves (count rows of 21 canonical felts), before padding:
 * non-empty, at most num_leaf_proofs, asset_id 0 when padding will be needed, one asset everywhere, one block hash and
 * fee rate among the non-dummy ones, pairwise distinct nullifiers among them, at least one non-dummy. */
int qpgpu_private_batch_preflight(const uint64_t *leaf_pis, size_t count, size_t num_leaf_proofs, char *err);
/* verify_dummy_leaf_template's sentinel part: zero block hash, zero amounts, asset 0, zero exit accounts */
int qpgpu_dummy_leaf_template_check(const uint64_t *pis, size_t n, char *err);
/* Pads `count` supplied proofs to num_leaf_proofs slots with the dummy template, shuffles uniformly and draws one dummy
 * nullifier preimage per slot. slot_source[s] = index of the supplied proof in slot s, or UINT32_MAX for the dummy
 * template; preimages: num_leaf_proofs * 4 canonical felts. seed32: NULL = operating-system entropy (getrandom); a
 * 32-byte seed makes the arrangement reproducible (tests). */
int qpgpu_private_batch_arrange(size_t count, size_t num_leaf_proofs, const uint8_t *seed32, uint32_t *slot_source,
                                uint64_t *preimages, char *err);
/* The public inputs the private-batch circuit emits for these slots (n_leaf rows of 21 felts in slot order, n_leaf
 * preimages of 4 felts): out has 21 n_leaf + 8 words. -4 when a constraint of the circuit is violated. */
int qpgpu_private_batch_outputs(const uint64_t *leaf_pis, size_t n_leaf, const uint64_t *dummy_preimages, uint64_t *out, char *err);

/* ---- public batch (miner side) ---- */
/* preflight_private_batch_proofs without the cryptographic part: count rows of pi_len felts; non-empty, at most
 * num_private_batch_proofs, pi_len == 21 N + 8 for an N in 1..=64, one (block hash, asset, fee) among the non-dummy
 * proofs, at least one non-dummy. */
int qpgpu_public_batch_preflight(const uint64_t *inner_pis, size_t count, size_t pi_len, size_t num_private_batch_proofs, char *err);
/* verify_dummy_private_batch_template's sentinel part */
int qpgpu_dummy_private_batch_template_check(const uint64_t *pis, size_t n, char *err);
/* The public inputs the public-batch circuit emits: m rows of 21 n_leaf + 8 felts in order (no shuffle: forwarding is
 * order-preserving), the aggregator address as 32 bytes (4 felts, 8 bytes each, canonical); out has
 * qpgpu_public_batch_pi_len(m, n_leaf) words. -4 when a constraint of the circuit is violated. */
int qpgpu_public_batch_outputs(const uint64_t *inner_pis, size_t m, size_t n_leaf, const uint8_t aggregator_address[32],
                               uint64_t *out, char *err);

/* ---- inner-proof targets: the witness of a recursive wrapper circuit -----------------
 * fill_private_batch_witness (wormhole/aggregator/src/private_batch/prover/witness.rs:15-77) assigns N complete inner proofs
 * to the wrapper's proof targets with `pw.set_proof_with_pis_target`, after ensure_proof_shape_matches_targets
 * (wormhole/aggregator/src/common/utils.rs:295-540) has compared every vector length of the proof with its target, and then the
 * N x 4 dummy-nullifier preimage elements. A proof target is one virtual target per field element of a proof of the INNER
 * circuit (`add_virtual_proof_with_pis(common_data)`); here the tree is flattened into LOGICAL TARGETS 0 .. T-1 per proof slot,
 * in this order (the order `set_proof_with_pis_target` visits them in upstream plonky2; any enumeration would do, the exporter
 * and this header only have to agree):
 *     public_inputs[ ]
 *     wires_cap, plonk_zs_partial_products_cap, quotient_polys_cap                  (4 elements per digest)
 *     openings at zeta:  constants, plonk_sigmas, wires, plonk_zs, partial_products, quotient_polys   (2 per extension element)
 *     openings at g*zeta: plonk_zs_next                                                       (lookup vectors are empty)
 *     opening_proof.pow_witness
 *     opening_proof.final_poly coefficients                                                     (2 each)
 *     opening_proof.commit_phase_merkle_caps
 *     per query round: per initial oracle: evals (base elements), siblings (digests); per reduction step: evals (extension), siblings
 * The id of target j of the proof in slot i is i * T + j; the preimage element `limb` of slot i is N * T + 4 i + limb. The
 * circuit-pack exporter records which wire cell each logical target became (integration/qpgpu_backend.rs: target map), as for
 * the leaf circuit (include/qpgpu_leaf.h): qpgpu_leaf_map_targets turns (id, value) pairs into the (cell, value) list of
 * qpgpu_generate_witness_partial_dev.
 *
 * A SHAPE is the list of a proof's vector lengths as 32-bit words, in the order the reference's check visits them:
 *   public_inputs, wires_cap, plonk_zs_partial_products_cap, quotient_polys_cap, openings.{constants, plonk_sigmas, wires,
 *   plonk_zs, plonk_zs_next, partial_products, quotient_polys, lookup_zs, lookup_zs_next}, number of commit-phase caps and each
 *   cap's length, number of query rounds and per round { number of initial oracles, (evals, siblings) each, number of steps,
 *   (evals, siblings) each }, final_poly. */
/* the shape of the proof targets of `inner_pack`'s circuit; out may be NULL to ask for the word count */
int qpgpu_proof_target_shape(const uint64_t *inner_pack, size_t n_words, uint32_t *out, size_t cap, size_t *count, char *err);
/* the shape `ProofWithPublicInputs::from_bytes(bytes, common_data)` yields: the bytes carry only two kinds of length, the Merkle
 * paths' (one byte each) and, implicitly, the number of public inputs (whatever follows pow_witness); -1 when the bytes end
 * inside a vector or hold a non-canonical element */
int qpgpu_proof_shape_of_bytes(const uint64_t *inner_pack, size_t n_words, const uint8_t *proof, size_t len, uint32_t *out, size_t cap,
                               size_t *count, char *err);
/* ensure_proof_shape_matches_targets: 0, or -1 with the reference's message
 * "{label} at slot {slot} is malformed: {what} has length {actual}, but the circuit expects {expected}" for the first mismatch */
int qpgpu_ensure_proof_shape_matches_targets(const uint32_t *target_shape, size_t n_target, const uint32_t *proof_shape, size_t n_proof,
                                             size_t slot, const char *label, char *err);
size_t qpgpu_proof_target_count(const uint64_t *inner_pack, size_t n_words);      /* T; 0 for a bad pack */
/* one proof's values in logical-target order (shape-checked first); values_out may be NULL to ask for the count */
int qpgpu_proof_target_values(const uint64_t *inner_pack, size_t n_words, const uint8_t *proof, size_t len, size_t slot, const char *label,
                              uint64_t *values_out, size_t cap, size_t *count, char *err);
/* fill_private_batch_witness: its three count checks with its messages (num_proof_targets / num_preimage_targets are what the
 * wrapper circuit was built for), the shape check of every proof, then the assignments: num_proofs * (T + 4) pairs
 * (logical target id, value) in the reference's order — all proofs, then all preimages (4 canonical elements per slot).
 * targets_out / values_out may be NULL to ask for the count. */
int qpgpu_batch_fill_proof_targets(const uint64_t *inner_pack, size_t n_words, const uint8_t *const *proofs, const size_t *proof_lens, size_t num_proofs,
                                   size_t num_proof_targets, const uint64_t *dummy_nullifier_preimages, size_t num_preimages, size_t num_preimage_targets,
                                   const char *label, uint32_t *targets_out, uint64_t *values_out, size_t cap, size_t *count, char *err);

/* ---- the recursive verifier and the two batch circuits (csrc/wrapper_circuit.cpp) -------------------------------------------
 * add_recursive_verifiers (wormhole/aggregator/src/common/recursive.rs:74-102) restated on the library's native builder:
 * `num_proofs` proof targets of the circuit `inner_pack` (logical targets in the order above) and verify_proof on each. Without
 * flags: the commitment half — for every proof and every query round the in-circuit Merkle verification of its four opened rows
 * and of every FRI step's coset of evaluations (hash the row with PoseidonGate rows, one permute_swapped per path level,
 * RandomAccessGate look-up of the cap entry) against the inner circuit's constants/sigmas cap (`inner_cs_cap`, 4 << cap_height
 * words: qpgpu_circuit_constants_sigmas_cap of the loaded inner circuit; it becomes constants of the wrapper) and against the caps
 * the proof carries, the query indices being inputs (qpgpu_verifier_query_indices); the inner public inputs are forwarded. The
 * flags below add the transcript, the arithmetic half (with both: everything VerifierCircuitData::verify checks), the two batch
 * layers' own constraints and the private layer's zero-knowledge configuration. A flipped byte of an inner proof makes the
 * wrapper's witness unsatisfiable (QPGPU_EUNSAT at witness generation, naming the target). examples/batch_prove_example.c drives
 * both layers from C.
 * Logical targets of the wrapper, target_map_out[...] = wire cell or UINT64_MAX: proof slot i target j -> i * T + j (T =
 * qpgpu_proof_target_count); dummy-nullifier preimage limb -> N * T + 4 i + limb (assigned by fill_private_batch_witness, unused
 * here); query index q of slot i -> N * (T + 4) + i * Q + q. num_routed_wires: 80 (public batch) or 60 (private batch),
 * 0 = 80. info_out (QPGPU_WRAPPER_CIRCUIT_INFO_WORDS, may be NULL): degree_bits, rows before padding, T, Q, PoseidonGate rows,
 * RandomAccessGate rows, BaseSumGate rows, ArithmeticGate rows, ConstantGate rows, public inputs, rows of the verification part,
 * blinding rows (zero-knowledge circuits). */
#define QPGPU_WRAPPER_CIRCUIT_INFO_WORDS 12
/* flags: QPGPU_WRAPPER_TRANSCRIPT — the Fiat-Shamir transcript of every inner proof is replayed IN-CIRCUIT (RecursiveChallenger on
 * PoseidonGate rows: circuit digest, public-input hash, caps, openings, FRI caps, final polynomial, proof-of-work witness, in the
 * prover's order), the proof-of-work response is range-checked (fri_verify_proof_of_work) and the 28 query indices are the low
 * bits of the transcript's challenges instead of inputs: the query-index logical targets then map to UINT64_MAX and need no
 * assignment. The Plonk / FRI challenges are derived too; QPGPU_WRAPPER_VERIFY consumes them. */
#define QPGPU_WRAPPER_TRANSCRIPT 1u
/* QPGPU_WRAPPER_PRIVATE_BATCH — the private-batch layer's own logic on top (build_private_batch_constraints,
 * wormhole/aggregator/src/private_batch/circuit/circuit_logic.rs:171-477): the inner circuit must have the leaf's 21 public inputs;
 * dummy flags from the zero block hash, references from the first real slot, block / asset / fee consistency, exit-account
 * grouping with duplicates zeroed and sums range-checked, pairwise distinct real nullifiers, dummy nullifiers replaced by
 * H(H(preimage)) (Poseidon2 gate rows), all nullifiers through the sorting network. The circuit's public inputs are then the
 * 21 N + 8 words qpgpu_private_batch_outputs computes on the host (the witness is unsatisfiable when they differ).
 * QPGPU_WRAPPER_PUBLIC_BATCH — the public-batch layer's (build_public_batch_constraints, public_batch/circuit/circuit_logic.rs:
 * 167-317): the inner circuit's public inputs are a private batch's (21 N + 8); the aggregator address is assigned through the
 * FIRST preimage slot's four targets; public inputs = what qpgpu_public_batch_outputs computes. The two exclude each other. */
#define QPGPU_WRAPPER_PRIVATE_BATCH 2u
#define QPGPU_WRAPPER_PUBLIC_BATCH 4u
/* QPGPU_WRAPPER_VERIFY (needs QPGPU_WRAPPER_TRANSCRIPT) — the arithmetic half of verify_proof in-circuit as well, on
 * ArithmeticExtensionGate rows: the openings against the vanishing polynomial at zeta (every gate of the inner circuit, the
 * permutation argument, Z(1) = 1; the same generic expressions the host verifier evaluates, csrc/verify_math.hpp) equal to
 * Z_H(zeta) * quotient(zeta) for every challenge; per query round the opened rows reduced with the FRI alpha and divided by
 * (x - zeta) / (x - g zeta), every reduction step's coset interpolated at its beta (and holding the previous evaluation at the
 * index's position), the final polynomial at the last point. With it the wrapper enforces everything
 * VerifierCircuitData::verify checks on an inner proof.
 * QPGPU_WRAPPER_ZERO_KNOWLEDGE — the circuit is built zero-knowledge as the reference's private-batch layer is
 * (wormhole_private_batch_circuit_config, common/src/circuit.rs:396-402; pass num_routed_wires = 60 for its wire budget):
 * CircuitBuilder::blind's rows are added before padding (upstream plonky2's counts; the fork's "row blinding" mode is
 * un-vendored), the pack says zero_knowledge = 1 (the prover salts the Merkle leaves of the wires / Z / quotient oracles). The
 * wire cells that need a FRESH RANDOM value per proof follow the query-index block of the target map, as cells (no logical
 * id); the caller assigns them with the PartialWitness — qpgpu_random_field_elements draws them. */
#define QPGPU_WRAPPER_VERIFY 8u
#define QPGPU_WRAPPER_ZERO_KNOWLEDGE 16u
/* n uniform field elements from ChaCha20 keyed with seed32, or with 32 bytes of operating-system entropy when seed32 is NULL
 * (RandomValueGenerator's F::rand(), for the blinding cells of a zero-knowledge circuit) */
int qpgpu_random_field_elements(const uint8_t *seed32, uint64_t *out, size_t n, char *err);

/* ---- gadget circuits: the builder's gadgets one at a time, for tests (csrc/gadget_circuits.cpp) ---------------------------------
 * A small circuit that applies ONE family of gadgets of the native builder to free inputs: pack_out is its circuit pack,
 * cells_out its input cells followed by its output cells (row * num_wires + wire), n_inputs / n_outputs their counts. A test
 * assigns the inputs, lets a witness generator run and compares the outputs with the gadget's definition computed independently
 * (tests/test_builder_gadgets.py). kind:
 *   0  extension arithmetic     in: a, b, c (2 each)                     out: a b, a b + c, a - b, a / b
 *   1  reducing                 in: alpha (2), 100 base terms, 40 extension terms     out: sum t_i alpha^i for both lists
 *   2  coset interpolation      in: shift, 16 extension values, point (2)            out: the interpolant through (shift w^i, v_i) at point
 *   3  bits                     in: x (< 2^10), y (any)                   out: 7^x (exp_from_bits_const_base), le_sum(split_le(x, 10)), the 64 bits of y
 *   4  selection                in: index (< 16), 16 values, b (0/1), u, v            out: values[index], select(b, u, v), is_equal(u, v)
 *   5  digest order             in: 5 digests (4 each)                    out: the digests sorted (sort_digests4), digest_eq(d0, d1)
 *   6  constant comparisons     in: r8 (< 2^8), r1 (< 2), x (any)          out: 3 < r8 (width 8), 0 < r1 (width 1), 0 < x, 1 < x, p - 2 < x, p - 1 < x
 *                                                                           (width 64: is_const_less_than's canonical-half path, gadgets.rs:40-97)
 *   7  0 < x forced true        in: x                                      out: x          (x = 0 has no witness: gadgets.rs:393-412)
 *   8  a comparison 65 bits wide: refused when the circuit is built ("exceeds 64 bits", gadgets.rs:414-421)
 *   3000 + K: K unconstrained public inputs (a stand-in inner circuit for the batch layers' logic tests; in = out = the K values)
 *   1000 + seed: a random program of 40-80 gadget applications over 6 inputs (differential tests of builder, stage s1 and prover) */
/* gates sort_digests4 adds over n virtual digests in a builder with num_routed_wires routed wires — the reference pins this cost
 * (common/src/gadgets.rs:423-458: 900 gates for n = 8, 57 000 for n = 64 under the private-batch config's 60 routed wires, "measured
 * cost + 15 %"), which makes it a check of the native builder's slot packing against plonky2's */
int qpgpu_builder_sort_gate_cost(unsigned n, unsigned num_routed_wires, size_t *gates, char *err);
int qpgpu_builder_gadget_circuit(unsigned kind, uint64_t *pack_out, size_t pack_cap_words, size_t *pack_words, uint64_t *cells_out, size_t cells_cap,
                                 size_t *n_inputs, size_t *n_outputs, char *err);
int qpgpu_wrapper_circuit_build(const uint64_t *inner_pack, size_t inner_words, const uint64_t *inner_cs_cap, size_t cap_words, unsigned num_proofs,
                                unsigned num_routed_wires, unsigned min_degree_bits, int inner_hasher, unsigned flags, uint64_t *pack_out, size_t pack_cap_words,
                                size_t *pack_words, uint64_t *target_map_out, size_t map_cap, size_t *map_count, uint64_t *info_out, char *err);

#ifdef __cplusplus
}
#endif
#endif
