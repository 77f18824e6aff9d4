/*
 * qpgpu_leaf.h — C ABI of the native leaf-witness front-end (host only, no GPU): what the reference does between
 * `CircuitInputs` and the PartialWitness of the Wormhole leaf circuit, i.e. SURVEY.md section 8 row a1 / f2:
 *
 *   WormholeProver::commit -> fill_witness                     wormhole/prover/src/lib.rs:156-163,187-221
 *     Nullifier::fill_targets                                   wormhole/circuit/src/nullifier.rs:345-354
 *     UnspendableAccount::fill_targets                          wormhole/circuit/src/unspendable_account.rs:239-248
 *     ZkMerkleProofData::try_from / fill_targets                wormhole/circuit/src/zk_merkle_proof.rs:424-470,628-700
 *     DualExitAccount::fill_targets                             wormhole/circuit/src/substrate_account.rs:152-165
 *     BlockHeader::fill_targets, HeaderInputs::new              wormhole/circuit/src/block_header/mod.rs:130-147, header.rs:110-131
 *   byte <-> felt codecs                                        common/src/serialization.rs:62-247
 *   public-input order (21 felts)                               wormhole/inputs/src/lib.rs:68-80
 *
 * fill_witness is pure encoding: every value it assigns is a re-encoding of an input field (the hashes are inputs; the
 * circuit recomputes them). It is therefore pinned by the reference's encoding anchors alone
 * (wormhole/prover/src/lib.rs:262-271, common/src/serialization.rs:92-97) and does not depend on the Poseidon2 constants.
 *
 * The helpers that DERIVE hashes natively (nullifier from its preimage, unspendable account from the secret, block hash)
 * run the fork's Poseidon2 sponge: `input || 1 || 0*` to a multiple of the rate 8
 * (wormhole/circuit/tests/heap_zeroization.rs:133-160), every block added into the rate part of the state. qp-poseidon-core
 * 3.1.0's constants are not in the reference tree; the parameter set the library carries
 * (qpgpu_poseidon2_qp_params: Plonky3's new_from_rng_128 on ChaCha20Rng::seed_from_u64(0x3141592653589793), MDSMat4,
 * MATRIX_DIAG_12_GOLDILOCKS) was found by search and is PINNED by all seven of the reference's known-answer vectors
 * (5 addresses, wormhole/tests/src/circuit/unspendable_account_tests.rs:9-24; 2 block hashes,
 * wormhole/tests/test-helpers/src/lib.rs:210-219; tests/test_leaf_witness.py). Pass params = NULL, n_words = 0 to use it.
 *
 * Assignments are reported against LOGICAL targets of the leaf circuit (enum below), in the order fill_witness sets
 * them; the circuit-pack exporter (integration/) records which wire cell each logical target became, and
 * qpgpu_leaf_map_targets turns the pair list into the (cell, value) list qpgpu_generate_witness_partial_dev takes.
 */
#ifndef QPGPU_LEAF_H
#define QPGPU_LEAF_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QPGPU_LEAF_PUBLIC_INPUTS 21          /* wormhole/inputs/src/lib.rs:33 */
#define QPGPU_LEAF_MAX_DEPTH 16              /* common/src/zk_merkle.rs:65 */
#define QPGPU_LEAF_DIGEST_LOGS_SIZE 110      /* wormhole/circuit/src/block_header/header.rs:14 */
#define QPGPU_LEAF_DIGEST_LOGS_FELTS 28
#define QPGPU_LEAF_ERR_CAP 160

/* CircuitInputs = PublicCircuitInputs + PrivateCircuitInputs (wormhole/circuit/src/inputs.rs:30-83). 32-byte fields are
 * BytesDigest: four little-endian 8-byte limbs, each below the Goldilocks modulus (wormhole/inputs/src/lib.rs:148-167). */
typedef struct {
    /* public */
    uint32_t asset_id, output_amount_1, output_amount_2, volume_fee_bps;
    uint8_t nullifier[32], exit_account_1[32], exit_account_2[32], block_hash[32];
    uint32_t block_number;
    /* private */
    uint8_t secret[32];
    uint64_t transfer_count;
    uint8_t unspendable_account[32], parent_hash[32], state_root[32], extrinsics_root[32];
    uint8_t digest[QPGPU_LEAF_DIGEST_LOGS_SIZE];
    uint32_t input_amount;
    uint8_t zk_tree_root[32];
    uint32_t zk_merkle_depth;                                   /* siblings.len() == positions.len() */
    uint8_t zk_merkle_siblings[QPGPU_LEAF_MAX_DEPTH][3][32];    /* sorted order, current hash excluded */
    uint8_t zk_merkle_positions[QPGPU_LEAF_MAX_DEPTH];          /* 0..3 */
} qpgpu_leaf_inputs;

/* Logical targets of CircuitTargets, grouped as fill_witness visits them. A target id is base + index. */
enum {
    QPGPU_LT_NULLIFIER_HASH = 0,            /* 4  nullifier.rs:350 */
    QPGPU_LT_NULLIFIER_SECRET = 4,          /* 4 */
    QPGPU_LT_NULLIFIER_TRANSFER_COUNT = 8,  /* 2 */
    QPGPU_LT_UNSPENDABLE_ACCOUNT_ID = 10,   /* 4  unspendable_account.rs:244 */
    QPGPU_LT_UNSPENDABLE_SECRET = 14,       /* 4 */
    QPGPU_LT_ZK_ROOT_HASH = 18,             /* 4  zk_merkle_proof.rs:654 */
    QPGPU_LT_ZK_DEPTH = 22,                 /* 1 */
    QPGPU_LT_ZK_SIBLINGS = 23,              /* 16 x 3 x 4, level major */
    QPGPU_LT_ZK_POSITIONS = 215,            /* 16 */
    QPGPU_LT_LEAF_TO_ACCOUNT = 231,         /* 4 */
    QPGPU_LT_LEAF_TRANSFER_COUNT = 235,     /* 2 */
    QPGPU_LT_LEAF_ASSET_ID = 237,
    QPGPU_LT_LEAF_INPUT_AMOUNT = 238,
    QPGPU_LT_LEAF_OUTPUT_AMOUNT_1 = 239,
    QPGPU_LT_LEAF_OUTPUT_AMOUNT_2 = 240,
    QPGPU_LT_LEAF_VOLUME_FEE_BPS = 241,
    QPGPU_LT_EXIT_ACCOUNT_1 = 242,          /* 4  substrate_account.rs:157 */
    QPGPU_LT_EXIT_ACCOUNT_2 = 246,          /* 4 */
    QPGPU_LT_BLOCK_HASH = 250,              /* 4  block_header/mod.rs:135 */
    QPGPU_LT_HEADER_PARENT_HASH = 254,      /* 4 */
    QPGPU_LT_HEADER_BLOCK_NUMBER = 258,
    QPGPU_LT_HEADER_STATE_ROOT = 259,       /* 4 */
    QPGPU_LT_HEADER_EXTRINSICS_ROOT = 263,  /* 4 */
    QPGPU_LT_HEADER_ZK_TREE_ROOT = 267,     /* 4 */
    QPGPU_LT_HEADER_DIGEST = 271,           /* 28 */
    QPGPU_LT_COUNT = 299
};

/* ---- codecs (common/src/serialization.rs) ---- */
/* bytes_to_felts: 4 bytes per element little-endian after appending the terminator 0x01 and zero padding; returns the
 * number of elements (len / 4 + 1), or 0 when out_cap is too small or len exceeds 1 MiB (MAX_SERIALIZED_BYTES). */
size_t qpgpu_bytes_to_felts(const uint8_t *in, size_t len, uint64_t *out, size_t out_cap);
/* inverse; returns the byte count, or (size_t)-1 for a missing / misplaced terminator or an element above 32 bits */
size_t qpgpu_felts_to_bytes(const uint64_t *in, size_t n, uint8_t *out, size_t out_cap);
/* 32 bytes <-> 4 elements, 8 bytes per element little-endian (from_noncanonical_u64: reduced mod p on the way in) */
void qpgpu_bytes_to_digest(const uint8_t in[32], uint64_t out[4]);
void qpgpu_digest_to_bytes(const uint64_t in[4], uint8_t out[32]);
/* BytesDigest::try_from: 1 when every 8-byte limb is canonical (< p) */
int qpgpu_bytes_digest_is_canonical(const uint8_t in[32]);
/* u64 / u128 as 32-bit limbs, most significant first (u128: hi 64 bits then lo 64 bits) */
void qpgpu_u64_to_felts(uint64_t v, uint64_t out[2]);
void qpgpu_u128_to_felts(uint64_t hi, uint64_t lo, uint64_t out[4]);

/* ---- fill_witness ---- */
/* public_inputs_out: the 21 public inputs in registration order (asset_id, output_amount_1, output_amount_2,
 * volume_fee_bps, nullifier x4, exit_account_1 x4, exit_account_2 x4, block_hash x4, block_number).
 * targets_out / values_out: QPGPU_LT_COUNT (299) assignments in fill_witness order. Returns 0, or -1 with a message in
 * err (QPGPU_LEAF_ERR_CAP bytes) for inputs the reference rejects: depth above 16, a position above 3, a 32-byte field
 * with a non-canonical limb. */
int qpgpu_leaf_fill_witness(const qpgpu_leaf_inputs *in, uint64_t public_inputs_out[QPGPU_LEAF_PUBLIC_INPUTS],
                            uint32_t *targets_out, uint64_t *values_out, size_t cap, size_t *count, char *err);
/* is_not_dummy as ZkMerkleProofData::try_from derives it (block_hash == 0 and both outputs == 0 mean dummy) */
int qpgpu_leaf_is_not_dummy(const qpgpu_leaf_inputs *in);
/* logical targets -> wire cells through the exporter's target map (target_map[id] = row * num_wires + wire, or
 * UINT64_MAX for a target the builder optimised away); returns the number of (cell, value) pairs written */
size_t qpgpu_leaf_map_targets(const uint32_t *targets, const uint64_t *values, size_t count, const uint64_t *target_map,
                              size_t map_len, uint64_t *cells_out, uint64_t *cell_values_out);

/* ---- Poseidon2 sponge of the fork: params = NULL, n_words = 0 -> qp-poseidon-core's pinned set; otherwise a caller's block
 * (146 words, layout of qpgpu_ctx_set_hasher) ---- */
/* the pinned set as 146 words (for qpgpu_ctx_set_hasher, should the fork's proof-system hasher be Poseidon2); returns 146 */
size_t qpgpu_poseidon2_qp_params(uint64_t *out, size_t cap);
int qpgpu_poseidon2_permute(const uint64_t *params, size_t n_words, uint64_t state[12]);
/* Poseidon2Hash::hash_no_pad of the fork: pads `|| 1 || 0*` to a multiple of 8, additive absorption, 4 outputs */
int qpgpu_poseidon2_hash_pad10(const uint64_t *params, size_t n_words, const uint64_t *in, size_t n, uint64_t out[4]);
/* hash_no_pad_bytes: the same, output as 32 bytes (digest_to_bytes) */
int qpgpu_poseidon2_hash_bytes(const uint64_t *params, size_t n_words, const uint64_t *in, size_t n, uint8_t out[32]);
/* H(H(felts("wormhole") || secret)): UnspendableAccount::from_secret (unspendable_account.rs:63-94) */
int qpgpu_leaf_unspendable_account(const uint64_t *params, size_t n_words, const uint8_t secret[32], uint8_t out[32]);
/* H(H(felts("~nullif~") || secret || transfer_count)): Nullifier::from_preimage (nullifier.rs:103-128) */
int qpgpu_leaf_nullifier(const uint64_t *params, size_t n_words, const uint8_t secret[32], uint64_t transfer_count, uint8_t out[32]);
/* HeaderInputs::block_hash over the 45-element preimage (block_header/header.rs:132-141) */
int qpgpu_leaf_block_hash(const uint64_t *params, size_t n_words, const uint8_t parent_hash[32], uint32_t block_number,
                          const uint8_t state_root[32], const uint8_t extrinsics_root[32], const uint8_t zk_tree_root[32],
                          const uint8_t digest[QPGPU_LEAF_DIGEST_LOGS_SIZE], uint8_t out[32]);

/* ---- the chain's 4-ary ZK Merkle tree, natively (common/src/zk_merkle.rs) ----
 * Leaves: Poseidon2 of (to_account x4 at 8 bytes per element, transfer_count as two 32-bit limbs, asset_id, input_amount)
 * (wormhole/circuit/src/zk_merkle_proof.rs:103-112,222-262,503-504). Internal nodes: Poseidon2 over the four children's 16
 * limbs (8 bytes per element), children in sorted byte order; a proof carries the three siblings of every level in sorted
 * order plus the position (0..3) where the running hash is inserted, so that neither the verifier nor the circuit sorts.
 * Hash bytes must be canonical (every 8-byte limb below p): a non-canonical alias would hash like the genuine child. */
#define QPGPU_ZK_ARITY 4
int qpgpu_zk_leaf_hash(const uint8_t to_account[32], uint64_t transfer_count, uint32_t asset_id, uint32_t input_amount, uint8_t out[32]);
/* hash_node_presorted: children = 4 x 32 bytes as given; -1 for a non-canonical child */
int qpgpu_zk_hash_node_presorted(const uint8_t *children, uint8_t out[32]);
/* hash_node: sorts the children first (order independent) */
int qpgpu_zk_hash_node(const uint8_t *children, uint8_t out[32]);
/* insert_at_position: out = 4 x 32 bytes; -1 when position > 3 */
int qpgpu_zk_insert_at_position(const uint8_t current[32], const uint8_t *sorted_siblings, unsigned position, uint8_t *out);
/* ZkMerkleProof::verify (= verify_with_positions): siblings = depth x 3 x 32 bytes, positions = depth bytes. Returns 1 when
 * the proof leads to root, 0 otherwise (depth above 16, a position above 3, a non-canonical hash, a different root). */
int qpgpu_zk_proof_verify(const uint8_t leaf_hash[32], const uint8_t *siblings, const uint8_t *positions, size_t depth, const uint8_t root[32]);
/* ZkMerkleProof::from_unsorted: sorts every level's siblings, computes the position hints and the root the path leads to.
 * sorted_out: depth x 3 x 32 bytes, positions_out: depth bytes. -1 with the reference's message for a depth above 16 or
 * non-canonical hash bytes. */
int qpgpu_zk_proof_from_unsorted(const uint8_t leaf_hash[32], const uint8_t *unsorted_siblings, size_t depth, uint8_t *sorted_out,
                                 uint8_t *positions_out, uint8_t root_out[32], char *err);

/* ---- the leaf circuit's constraints, natively ----
 * What WormholeCircuit constrains about a CircuitInputs (wormhole/circuit/src/circuit.rs:233-323 and the fragments it wires:
 * unspendable_account.rs:215-237, nullifier.rs:285-325, block_header/mod.rs:93-108, zk_merkle_proof.rs:480-626), evaluated on
 * the host before any proving: the unspendable account is H(H("wormhole" || secret)); the fee relation
 * (out_1 + out_2) * 10000 <= input * (10000 - fee_bps) with fee_bps <= 10000; depth <= 16 and positions <= 3; and, unless the
 * inputs are a dummy (zero block hash and zero outputs): the nullifier is H(H("~nullif~" || secret || transfer_count)), the
 * block hash is the Poseidon2 hash of the header, and the Merkle path from the leaf hash leads to the header's ZK tree
 * root. Returns 0 when a proof can be generated, -4 naming the first constraint that cannot hold (what the reference's
 * negative tests observe as a failed prove), -1 for malformed inputs (as qpgpu_leaf_fill_witness). */
int qpgpu_leaf_check_constraints(const qpgpu_leaf_inputs *in, char *err);

/* ---- the leaf circuit itself, natively ----
 * WormholeCircuit::new(wormhole_leaf_circuit_config()) + build_prover() (wormhole/circuit/src/circuit.rs:115-152,210-212),
 * restated statement by statement on a native restatement of plonky2's CircuitBuilder (qp-zk-circuits_amd/csrc/builder.hpp,
 * leaf_circuit.cpp): the five fragments' targets in CircuitTargets::new's order (= the 21 public inputs' order), the
 * unspendable-account double hash, the 32 / 14 / 48-bit range checks and the fee relation, the leaf hash, the depth bound, the
 * 16-level 4-ary Merkle walk over position hints and selects, the block-number range check, and connect_shared_targets
 * (shared secret / transfer count / account, the in-circuit dummy flag, the conditional nullifier / block-hash / tree-root
 * bindings). Host only. Output: a circuit pack for qpgpu_circuit_load* and target_map_out[QPGPU_LT_COUNT] = the wire cell
 * (row * num_wires + wire) of every logical target above, for qpgpu_leaf_map_targets / qpgpu_leaf_commit.
 *   min_degree_bits: NoopGate padding up to 2^min_degree_bits rows (0: the next power of two above the gates used);
 *   inner_hasher: 0 Poseidon, 1 Poseidon2 — the gate the public-input hash is built from; must be the hasher of the context
 *     that proves (qpgpu_ctx_set_hasher; for Poseidon2: qp-poseidon-core's parameter set, which is also the gate's);
 *   p2_layout: the ten words of the pack trailer "P2GL1" (csrc/circuit.hpp), NULL = the default layout. LAYOUT UNPINNED.
 * What cannot match the fork offline: the order in which qp-plonky2's builder lays gates out, its Poseidon2 sponge wiring and its
 * circuit_digest (here: a hash of the pack's contents). The STATEMENT proven and the gate set are the reference's; the verifier
 * data are this builder's.
 * info_out (QPGPU_LEAF_CIRCUIT_INFO_WORDS words, may be NULL): degree_bits, rows before padding, gates after target creation,
 * gates added by UnspendableAccount::circuit, by ZkMerkleProofData::circuit, by the block-number range check, by
 * connect_shared_targets (the reference's GateProfiler checkpoints, wormhole/circuit/src/profile.rs), then rows of
 * ArithmeticGate, BaseSumGate, Poseidon2 gate, PoseidonGate, ConstantGate, PublicInputGate, NoopGate, free-standing generators,
 * selector polynomials.
 * Call with pack_out = NULL to learn the size. Returns 0, or a qpgpu.h error code with a message in err. */
#define QPGPU_LEAF_CIRCUIT_INFO_WORDS 16
/* fragment: the whole WormholeCircuit, or one CircuitFragment composed alone with its unconditional binding, the way the
 * reference's fragment tests build them (wormhole/tests/src/circuit/block_header_tests.rs:8-19, unspendable_account_tests.rs:26-40,
 * nullifier_tests.rs): BlockHeader::circuit (public inputs: block_hash x4, block_number), UnspendableAccount::circuit (none),
 * Nullifier::circuit (hash x4). Logical targets a fragment does not have map to UINT64_MAX. */
#define QPGPU_LEAF_FRAGMENT_FULL 0u
#define QPGPU_LEAF_FRAGMENT_BLOCK_HEADER 1u
#define QPGPU_LEAF_FRAGMENT_UNSPENDABLE_ACCOUNT 2u
#define QPGPU_LEAF_FRAGMENT_NULLIFIER 3u
/* build_fake_leaf_circuit (wormhole/tests/test-helpers/src/fake_leaf.rs): 21 free public inputs in the leaf layout + the three 32-bit
 * range checks; no logical targets (the public inputs are the whole witness) — for tests of the batch layers on arbitrary leaf values */
#define QPGPU_LEAF_FRAGMENT_FAKE_LEAF 4u
int qpgpu_leaf_circuit_build(unsigned fragment, unsigned min_degree_bits, int inner_hasher, const uint64_t *p2_layout, uint64_t *pack_out, size_t pack_cap_words,
                             size_t *pack_words, uint64_t *target_map_out, uint64_t *info_out, char *err);
/* WormholeProver::commit (wormhole/prover/src/lib.rs:156-163) against such a circuit: qpgpu_leaf_fill_witness followed by
 * qpgpu_leaf_map_targets. cells_out / values_out (room for QPGPU_LT_COUNT) are what qpgpu_generate_witness_partial_dev and
 * qpgpu_pool_submit_partial take. Returns 0 or -1 with the reference's message in err. */
int qpgpu_leaf_commit(const qpgpu_leaf_inputs *in, const uint64_t *target_map, uint64_t *cells_out, uint64_t *values_out, size_t cap,
                      size_t *count, uint64_t public_inputs_out[QPGPU_LEAF_PUBLIC_INPUTS], char *err);

/* ---- hash hints: the serial part of the leaf witness, handed in by the front-end ------------------------------------------------
 * The leaf circuit hashes in chains: 61 Poseidon2 gate rows at 8 call sites, every Merkle level waiting for the one below it. On the
 * device a row's generator is one dependent chain (about 19 us with its launch), 45 levels deep; a host core computes the same
 * permutation in under a microsecond. So the front-end, which knows every preimage, may hand the rows' outputs in as EXTRA assignments:
 * a PartialWitness may set targets that generators also produce (plonky2 checks them: "set twice with different values"), stage s1 then
 * builds its plan for that assignment list, runs all 61 rows in one level and CHECKS each against its hint — 120 dependency levels
 * become a few dozen. Optional: a caller that does not append the hints gets the same witness, later.
 * Call sites in the order of their tags (the order of the cells and of the values): */
enum qpgpu_leaf_hash_site {
    QPGPU_LEAF_HASH_UNSPENDABLE_INNER = 0,   /* H(salt "wormhole" || secret)            1 row   unspendable_account.rs:229-231 */
    QPGPU_LEAF_HASH_UNSPENDABLE_OUTER = 1,   /* H(inner)                                1 row */
    QPGPU_LEAF_HASH_ZK_LEAF = 2,             /* H(to, transfer_count, asset, amount)    2 rows  zk_merkle_proof.rs:482 */
    QPGPU_LEAF_HASH_MERKLE_LEVEL_0 = 3,      /* .. + 15: H(four children)               3 rows each  zk_merkle_proof.rs:504-606 */
    QPGPU_LEAF_HASH_NULLIFIER_INNER = 19,    /* H(salt "~nullif~" || secret || count)   2 rows  nullifier.rs:298-299 */
    QPGPU_LEAF_HASH_NULLIFIER_OUTER = 20,    /* H(inner)                                1 row */
    QPGPU_LEAF_HASH_BLOCK_HEADER = 21        /* H(header, 45 elements)                  6 rows  block_header/header.rs */
};
#define QPGPU_LEAF_HASH_ROWS 61
/* the whole 12-element state after every permutation, then the Merkle walk's running hash (4 elements) after each of its 16 levels */
#define QPGPU_LEAF_HASH_HINTS (12 * QPGPU_LEAF_HASH_ROWS + 4 * 16)
/* the cells (row * num_wires + wire) of those 796 values in the FULL leaf circuit built with the same arguments as
 * qpgpu_leaf_circuit_build (cells_out may be NULL to ask for the count) */
int qpgpu_leaf_circuit_hash_hint_cells(unsigned min_degree_bits, int inner_hasher, const uint64_t *p2_layout, uint64_t *cells_out, size_t cap, size_t *count, char *err);
/* the 796 values for one set of inputs, computed on the host exactly as the rows' generators would from the same assignments (field
 * arithmetic on the inputs as given: inconsistent inputs give hints that disagree with nothing but the targets the circuit itself
 * would refuse). values_out: room for QPGPU_LEAF_HASH_HINTS. Returns 0, or -1 with a message (inputs qpgpu_leaf_fill_witness refuses).
 * The values are as sensitive as qpgpu_leaf_commit's (the first sponge states are functions of the spend secret alone): wipe them after
 * the submit, as the library wipes its own copies. */
int qpgpu_leaf_hash_hints(const qpgpu_leaf_inputs *in, uint64_t *values_out, size_t cap, size_t *count, char *err);

#ifdef __cplusplus
}
#endif
#endif
