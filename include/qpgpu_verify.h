/*
 * qpgpu_verify.h — proof verification on the host (no GPU): the check every caller of the proving path applies to the proofs
 * it receives and to the proofs it made. SURVEY.md section 8 row f3.
 *
 *   leaf verification at the private batch's commit      wormhole/aggregator/src/private_batch/prover/lib.rs:274-281
 *   private-batch verification at the public batch       wormhole/aggregator/src/public_batch/prover/lib.rs:338-349
 *   self-verification after proving                      wormhole/aggregator/src/aggregator.rs:224-225
 *   WormholeVerifier::verify                             wormhole/verifier/src/lib.rs
 *
 * All of those call qp-plonky2's `VerifierCircuitData::verify` (plonk::verifier::verify_with_challenges +
 * fri::verifier::verify_fri_proof). This is that algorithm over a circuit pack: the Fiat-Shamir transcript replayed, the
 * vanishing polynomial (permutation argument + the selector-filtered constraints of the fourteen gate types) evaluated at
 * zeta in the quadratic extension and compared with Z_H(zeta) * quotient(zeta), the proof-of-work response, and for every
 * query round the Merkle paths of the four initial oracles, the reduced opening, each FRI reduction (coset interpolation at
 * beta) with its Merkle path, and the final polynomial. A verifier is verifier data (constants/sigmas cap + circuit digest +
 * common data), so it is created once per circuit and shared.
 */
#ifndef QPGPU_VERIFY_H
#define QPGPU_VERIFY_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QPGPU_EVERIFY (-6)        /* the proof is not accepted; err names the first failing check */
#define QPGPU_VERIFY_ERR_CAP 200

typedef struct qpgpu_verifier qpgpu_verifier;

/* pack: the circuit pack (csrc/circuit.hpp). cs_cap: the constants/sigmas Merkle cap, 4 << cap_height words — what
 * qpgpu_circuit_constants_sigmas_cap returns for a loaded circuit (VerifierOnlyCircuitData::constants_sigmas_cap); NULL
 * (cap_words 0): computed here from the pack's constants/sigmas columns on one host core (LDE + Merkle tree: a fraction of a
 * second at 2^10 rows, several seconds at 2^13 — prefer the cap of the GPU handle). hasher_kind / params: the proof system's
 * permutation as in qpgpu_ctx_set_hasher (QPGPU_HASH_POSEIDON, or QPGPU_HASH_POSEIDON2 with a parameter block; NULL, 0
 * selects the built-in qp-poseidon-core set). Returns 0 or QPGPU_EINVAL with the reason in err. */
int qpgpu_verifier_create(const uint64_t *pack_words, size_t n_words, const uint64_t *cs_cap, size_t cap_words, int hasher_kind,
                          const uint64_t *hasher_params, size_t n_params, qpgpu_verifier **out, char *err);
void qpgpu_verifier_free(qpgpu_verifier *v);
size_t qpgpu_verifier_proof_size(const qpgpu_verifier *v);      /* the one length a proof of this circuit has */
/* the verifier data's cap (4 << cap_height words), e.g. to compare with a pinned value */
int qpgpu_verifier_constants_sigmas_cap(const qpgpu_verifier *v, uint64_t *out, size_t out_words);
/* 0: accepted. QPGPU_EVERIFY: rejected, err says why (size, non-canonical element, proof of work, quotient identity,
 * Merkle path of oracle k at query q, FRI round consistency, final polynomial). Thread-safe on a shared verifier. */
int qpgpu_verifier_verify(const qpgpu_verifier *v, const uint8_t *proof, size_t len, char *err);

/* The FRI query indices of a proof (num_query_rounds words): the transcript replayed up to and including the proof of work,
 * everything it absorbs checked on the way; the query rounds themselves are not looked at, so a proof whose opened rows or
 * Merkle paths were tampered with still yields its indices. They are what a recursive verifier derives in-circuit; a wrapper
 * circuit that checks the Merkle paths only (qpgpu_wrapper_circuit_build, include/qpgpu_batch.h) takes them as inputs. */
int qpgpu_verifier_query_indices(const qpgpu_verifier *v, const uint8_t *proof, size_t len, uint64_t *out, size_t cap, char *err);

/* The same for `count` proofs of the circuit on up to `threads` host threads (0 = all cores): results[i] = 0 or
 * QPGPU_EVERIFY per proof; returns 0 when all are accepted, else QPGPU_EVERIFY with the first rejected proof's index and
 * reason in err. What PrivateBatchProver::commit does to its leaf proofs one after the other. */
int qpgpu_verifier_verify_many(const qpgpu_verifier *v, const uint8_t *const *proofs, const size_t *lens, size_t count, unsigned threads,
                               int *results, char *err);

#ifdef __cplusplus
}
#endif
#endif
