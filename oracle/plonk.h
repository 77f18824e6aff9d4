/*
 * oracle/plonk.h — CPU restatement of qp-plonky2 1.5.5 `plonk::prover::prove`, `fri::prover` and the
 * matching verifier, over a circuit pack (format: qp-zk-circuits_amd/csrc/circuit.hpp, "QPCP1").
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/gl.h). The algorithm is the one reached from the reference's
 * prove call sites (wormhole/prover/src/lib.rs:171-175 etc.); it lives in an un-vendored crate, so it is
 * restated from the published upstream algorithm. PARITY UNPINNED
 * for proof bytes: the reference holds no golden proof; what pins this file is (a) the
 * Poseidon / field / FFT vectors, (b) orc_verify accepting what orc_prove and the GPU prover emit.
 */
#ifndef ORACLE_PLONK_H
#define ORACLE_PLONK_H
#include "gl.h"

enum { OG_NOOP = 0, OG_CONSTANT = 1, OG_PUBLIC_INPUT = 2, OG_ARITHMETIC = 3, OG_POSEIDON = 4, OG_BASE_SUM = 5, OG_ARITHMETIC_EXT = 6, OG_MUL_EXT = 7,
       OG_REDUCING = 8, OG_REDUCING_EXT = 9, OG_RANDOM_ACCESS = 10, OG_EXPONENTIATION = 11, OG_POSEIDON_MDS = 12, OG_COSET_INTERP = 13, OG_POSEIDON2 = 14 };

typedef struct { uint64_t type, param0, param1, selector_index, group_start, group_end, num_constraints, reserved; } orc_gate;

typedef struct {
    size_t ncols, n, lde_n;      /* polynomials, degree, lde size */
    size_t width;                /* leaf width = ncols + salt columns (4 when blinding) */
    unsigned log_n, rate_bits, cap_height;
    gl_t *coeffs;                /* [ncols][n] */
    gl_t *leaves;                /* [lde_n][ncols] leaf order (leaf j = point bitrev(j)) */
    gl_t *digests;               /* level-ordered */
    gl_t *cap;                   /* 2^cap_height x 4 */
} orc_batch;

typedef struct {
    uint64_t degree_bits, num_wires, num_routed, num_constants, num_selectors, num_challenges, qdf, num_pp,
        num_pis, rate_bits, cap_height, pow_bits, num_queries, zk, num_gate_constraints;
    size_t n_arity; uint64_t arity[16];
    size_t n_gates; orc_gate *gates;
    gl_t *k_is; gl_t digest[4];
    uint64_t p2_layout[10];      /* wire layout of the Poseidon2 gate: pack trailer "P2GL1", or the default (oracle/poseidon2_gate.c) */
    gl_t *cs_values;             /* [ncs][n] */
    orc_batch cs;                /* constants_sigmas commitment (setup) */
} orc_circuit;

orc_circuit *orc_circuit_load(const uint64_t *words, size_t n_words);
void orc_circuit_free(orc_circuit *c);
/* returns 0 and writes proof bytes (ProofWithPublicInputs::to_bytes order); -1 buffer too small */
int orc_prove(const orc_circuit *c, const gl_t *wires, const gl_t *public_inputs, uint8_t *out, size_t cap, size_t *len);
/* zero-knowledge circuits: salts are drawn from a counter-mode generator keyed by `seed` (the reference uses thread_rng) */
int orc_prove_many_entry(const orc_circuit *c, const gl_t *wires, const gl_t *public_inputs, uint8_t *out, size_t cap, size_t *len);
int orc_prove_seeded(const orc_circuit *c, const gl_t *wires, const gl_t *public_inputs, uint64_t seed, uint8_t *out, size_t cap, size_t *len);
gl_t orc_salt_value(uint64_t seed, unsigned oracle_index, unsigned column, uint64_t leaf);
/* 0 = accepted; otherwise a positive stage code saying what failed */
int orc_verify(const orc_circuit *c, const uint8_t *proof, size_t len);
size_t orc_proof_size(const orc_circuit *c);

/* stage trace of the last orc_prove call in this process (not thread-safe; tests only) */
size_t orc_trace_len(const char *name);              /* number of u64 words, 0 if absent */
int orc_trace_get(const char *name, uint64_t *out);  /* copies the words */

/* the qp fork's Poseidon2 gate (oracle/poseidon2_gate.c): unfiltered constraints at one point; returns how many */
size_t orc_p2_gate_num_constraints(const uint64_t lay[10]);
size_t orc_p2_gate_base(const uint64_t lay[10], const gl_t *w, gl_t *out);
size_t orc_p2_gate_ext(const uint64_t lay[10], const gl2_t *w, gl2_t *out);

/* hashing primitives from poseidon.c */
void orc_poseidon_permute(gl_t s[12]);
void orc_hash_no_pad(const gl_t *in, size_t n, gl_t out[4]);
size_t orc_merkle_build(const gl_t *leaves, size_t n_leaves, size_t width, unsigned cap_height, gl_t *digests_out, gl_t *cap_out);
size_t orc_merkle_path(const gl_t *digests, size_t n_leaves, unsigned cap_height, size_t index, gl_t *path_out);
void orc_two_to_one(const gl_t l[4], const gl_t r[4], gl_t out[4]);
void orc_hash_or_noop(const gl_t *in, size_t n, gl_t out[4]);
void orc_fft(gl_t *a, unsigned log_n);
void orc_ifft(gl_t *a, unsigned log_n);
void orc_coset_fft(gl_t *a, unsigned log_n, gl_t shift);
void orc_coset_ifft(gl_t *a, unsigned log_n, gl_t shift);
#endif
