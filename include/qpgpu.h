/*
 * qpgpu.h — C ABI of libqpgpu, the MI355X (gfx950) backend for the qp-wormhole proving hot path.
 *
 * The reference has no FFI: its "operator API" is the qp-plonky2 1.5.5 type surface reached from
 *   wormhole/prover/src/lib.rs:171-175                         (leaf  prove)
 *   wormhole/aggregator/src/private_batch/prover/lib.rs:326-330 (private batch prove)
 *   wormhole/aggregator/src/public_batch/prover/lib.rs:301-305  (public batch prove)
 * The entry points below are what a patched qp-plonky2 `plonk::prover::prove` would bind, stage by
 * stage, plus the all-in-one qpgpu_prove. INTEGRATION.md shows the
 * Rust `extern "C"` block. Plain pointers and sizes only; the caller owns every host buffer; device
 * memory is owned by the library behind opaque handles (or raw device pointers the caller got from
 * qpgpu_malloc / its own allocator). All field elements are little-endian u64; inputs may be any
 * u64 (reduced on load), outputs are canonical (< p), as `to_canonical_u64` would serialise them
 * (reference common/src/serialization.rs:40-43).
 *
 * Every function returns 0 on success or a negative QPGPU_E* code; qpgpu_last_error(ctx) gives the
 * text. Nothing aborts the process. A ctx is bound to one GPU and one HIP stream and is not
 * thread-safe; use one ctx per in-flight proof stream.
 */
#ifndef QPGPU_H
#define QPGPU_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct qpgpu_ctx qpgpu_ctx;
typedef struct qpgpu_circuit qpgpu_circuit;   /* a loaded circuit pack + its device residents and workspace */

enum {
    QPGPU_OK = 0,
    QPGPU_EINVAL = -1,      /* bad argument (size, null pointer, unsupported log_n) */
    QPGPU_EDEVICE = -2,     /* HIP runtime error; see qpgpu_last_error */
    QPGPU_ENOMEM = -3,
    QPGPU_EUNSAT = -4,      /* witness does not satisfy the circuit (quotient not divisible by Z_H) */
    QPGPU_EBUFSIZE = -5     /* caller buffer too small */
};

/* ---- context ---- */
int qpgpu_ctx_create(int device, qpgpu_ctx **out);
void qpgpu_ctx_destroy(qpgpu_ctx *ctx);
const char *qpgpu_last_error(const qpgpu_ctx *ctx);
/* Use an existing HIP stream (hipStream_t passed as void*); NULL = the ctx's own stream. */
int qpgpu_ctx_set_stream(qpgpu_ctx *ctx, void *hip_stream);
int qpgpu_sync(qpgpu_ctx *ctx);
const char *qpgpu_version(void);
/* PCI address of the context's device as "dddd:bb:dd.f" (NUL-terminated, 13 bytes): the key under /sys/bus/pci/devices a
 * caller reads the card's engine clock and power from while it measures (bench.py's `engine_clock_mhz`). */
int qpgpu_ctx_pci_bus_id(const qpgpu_ctx *ctx, char *out, size_t out_len);
/* Post-mortem evidence (csrc/crash_trace.cpp): with QPGPU_CRASH_TRACE=1 (stderr) or =<file> in the environment when the
 * library is loaded, a fatal signal writes its address, the mapping that holds it and the native backtrace of the faulting
 * thread before the previous handler (e.g. python -X faulthandler) or the default action runs. 1 = armed. */
int qpgpu_crash_trace_armed(void);

/* ---- measurement: per-kernel HIP-event timing on the ctx stream (off by default) ---- */
int qpgpu_profile_enable(qpgpu_ctx *ctx, int on);   /* on: clears the counters */
/* kernel: "ntt_pass_strided", "ntt_pass_rows", "ntt_pass_single", ... ; sums since enable */
int qpgpu_profile_read(qpgpu_ctx *ctx, const char *kernel, double *total_ms, uint64_t *launches);

/* ---- device memory plumbing ---- */
int qpgpu_malloc(qpgpu_ctx *ctx, size_t bytes, void **dptr);
int qpgpu_free(qpgpu_ctx *ctx, void *dptr);
/* for buffers that held a witness: overwritten with zeros on the ctx stream, then released (the reference's zeroization
 * policy, wormhole/circuit/src/sensitive.rs:36-44) */
int qpgpu_free_scrubbed(qpgpu_ctx *ctx, void *dptr, size_t bytes);
int qpgpu_memcpy_h2d(qpgpu_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);
int qpgpu_memcpy_d2h(qpgpu_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);
int qpgpu_memcpy_d2d(qpgpu_ctx *ctx, void *dst_dev, const void *src_dev, size_t bytes);   /* asynchronous on the ctx stream */

/* ---- proof-system hasher -------------------------------------------------------------------------------------
 * Which permutation backs the fork's `PoseidonGoldilocksConfig` hasher (Merkle trees, Fiat-Shamir challenger, public-input
 * hash, proof of work) cannot be told from the reference, so it is a plug, and a property of the
 * context: kind 0 = plonky2's Poseidon (default; constants derived at start-up), kind 1 = Poseidon2 (width 12, x^7, 4+22+4
 * rounds) with caller-supplied parameters: 96 external round constants (round major), 22 internal round constants, the 12
 * diagonal entries d of the internal matrix J + diag(d), and the 4x4 block M4 (row major) of the external matrix
 * circ(2 M4, M4, M4) — 146 words. qpgpu_ctx_set_hasher must come before the first circuit or oracle is created on the
 * context (QPGPU_EINVAL afterwards); everything created on the context then hashes with it, and contexts with different
 * hashers coexist in one process. */
#define QPGPU_HASH_POSEIDON 0
#define QPGPU_HASH_POSEIDON2 1
#define QPGPU_POSEIDON2_PARAM_WORDS 146
int qpgpu_ctx_set_hasher(qpgpu_ctx *ctx, int kind, const uint64_t *params, size_t n_words);
int qpgpu_ctx_get_hasher(const qpgpu_ctx *ctx);
/* Deprecated process-wide default: what a new context (and the context-free helpers: synthetic circuits,
 * qpgpu_challenger_*, the workers of a proving pool) starts with. Existing contexts are not affected. */
int qpgpu_set_hasher(int kind, const uint64_t *params, size_t n_words);
int qpgpu_get_hasher(void);

/* ---- stage s2: plonky2::field::fft ---- */
enum {
    QPGPU_NTT_FORWARD = 0,        /* fft:  out[i] = P(w^i), natural order in and out */
    QPGPU_NTT_INVERSE = 1,        /* ifft: coefficients from values, includes the 1/n */
    QPGPU_NTT_OUT_BITREV = 2      /* flag: store out[bitrev(i)] (reverse_index_bits order) */
};
/*
 * Batched transform of `batch` polynomials of 2^log_n elements, polynomial c at data + c*2^log_n.
 * coset_shift: 0 or 1 = plain; otherwise forward = coset_fft(shift) (coefficients scaled by
 * shift^i first). Replaces fft_with_options / ifft_with_options / coset_fft. log_n <= 23; an LDE beyond 2^20 points
 * needs rate_bits <= 3 (the input must be at least 1/8 of the output).
 * Host-buffer form (copies in and out):
 */
int qpgpu_ntt_batch(qpgpu_ctx *ctx, uint64_t *data, unsigned log_n, size_t batch, int flags,
                    uint64_t coset_shift);
/* Device-buffer form; d_in == d_out is allowed. Asynchronous on the ctx stream. */
int qpgpu_ntt_batch_dev(qpgpu_ctx *ctx, const uint64_t *d_in, uint64_t *d_out, unsigned log_n,
                        size_t batch, int flags, uint64_t coset_shift);
/*
 * Low-degree extension: coefficients (2^log_n per polynomial) -> values of the zero-padded polynomial
 * on the coset shift*<w_{n<<rate_bits}>; PolynomialBatch::from_coeffs minus the Merkle step.
 * flags: 0 or QPGPU_NTT_OUT_BITREV. d_out holds batch << (log_n + rate_bits) elements.
 */
int qpgpu_lde_batch_dev(qpgpu_ctx *ctx, const uint64_t *d_coeffs, uint64_t *d_out, unsigned log_n,
                        unsigned rate_bits, size_t batch, int flags, uint64_t coset_shift);

/* ---- stage s3: plonky2::hash::{poseidon, merkle_tree} ---- */
/* Poseidon permutation (width 12) on n states of 12 elements each, in place. */
int qpgpu_poseidon_permute_dev(qpgpu_ctx *ctx, uint64_t *d_states, size_t n);
/* The APPLICATION hash of the Wormhole circuits on the device: Poseidon2Hash::hash_no_pad of the qp fork (reference call
 * sites wormhole/circuit/src/unspendable_account.rs:87-88, nullifier.rs:119-120, block_header/header.rs:140,
 * common/src/zk_merkle.rs:53-58) for `count` preimages of `len` elements each, row-major at d_in: padded `|| 1 || 0*` to a
 * multiple of the rate 8 (wormhole/circuit/tests/heap_zeroization.rs:133-160), blocks added into the rate part, 4 outputs
 * per preimage at d_out. params = NULL, n_words = 0: qp-poseidon-core's parameter set, pinned by the reference's seven
 * known-answer vectors (include/qpgpu_leaf.h); otherwise a 146-word block as for qpgpu_ctx_set_hasher. Independent of the
 * context's proof-system hasher. Asynchronous on the ctx stream with the built-in set. */
int qpgpu_poseidon2_hash_pad10_dev(qpgpu_ctx *ctx, const uint64_t *params, size_t n_words, const uint64_t *d_in, size_t len,
                                   size_t count, uint64_t *d_out);
/* Digests stored by a tree: level 0 (2^log_leaves leaf digests), then each parent level down to the
 * cap level (2^cap_height digests), concatenated; 4 elements per digest. */
size_t qpgpu_merkle_digest_count(unsigned log_leaves, unsigned cap_height);
/*
 * MerkleTree::new(leaves, cap_height) for leaves given column-major and already in leaf order:
 * leaf j = (d_cols[c*col_stride + j])_{c < n_cols}; leaf digest = hash_or_noop (rows of <= 4 elements
 * are copied, longer rows go through the overwrite-mode sponge, rate 8); node = two_to_one.
 * d_digests receives qpgpu_merkle_digest_count() digests; h_cap_out (host, optional) the cap.
 */
int qpgpu_merkle_build_dev(qpgpu_ctx *ctx, const uint64_t *d_cols, uint64_t col_stride, uint32_t n_cols,
                           unsigned log_leaves, unsigned cap_height, uint64_t *d_digests, uint64_t *h_cap_out);
/* Same for row-major leaves of `width` contiguous elements (FRI round trees). */
int qpgpu_merkle_build_rows_dev(qpgpu_ctx *ctx, const uint64_t *d_rows, uint32_t width, unsigned log_leaves,
                                unsigned cap_height, uint64_t *d_digests, uint64_t *h_cap_out);

/* ---- the whole hot path: ProverCircuitData::prove after witness generation (stages s2..s12) ---- */
/*
 * Load a circuit pack ("QPCP1", qp-zk-circuits_amd/csrc/circuit.hpp: CommonCircuitData + the prover-only
 * constants/sigmas, as a Rust-side exporter would dump them). Computes the constants_sigmas commitment on the
 * GPU (what ProverOnlyCircuitData::constants_sigmas_commitment holds) and allocates the per-proof workspace,
 * so qpgpu_prove* performs no allocation. Replaces the prover-side use of the data built by
 * wormhole/circuit/src/circuit.rs:210-212 (`builder.build_prover()`).
 */
int qpgpu_circuit_load(qpgpu_ctx *ctx, const uint64_t *pack_words, size_t n_words, qpgpu_circuit **out);
/* The same with a per-proof workspace for up to max_batch proofs proven in lockstep (qpgpu_prove_batch_dev): about
 * 260 MB per proof at 2^13 rows x 135 wires. The constants/sigmas commitment is shared by the batch. */
int qpgpu_circuit_load_batch(qpgpu_ctx *ctx, const uint64_t *pack_words, size_t n_words, unsigned max_batch, qpgpu_circuit **out);
unsigned qpgpu_circuit_max_batch(const qpgpu_circuit *c);
size_t qpgpu_circuit_num_public_inputs(const qpgpu_circuit *c);
/* Releases the handle; every device region that held witness-derived data is overwritten first. */
void qpgpu_circuit_free(qpgpu_circuit *c);
/* Overwrites the witness-derived workspace now (witness copy, Z / quotient values, coefficients, LDEs, salts, FRI
 * workspace, openings). qpgpu_prove does this after every proof; the _dev and pool entries leave the workspace as it is
 * until the next proof or the free (the caller owns the witness buffer it passed and clears that itself). */
int qpgpu_circuit_scrub(qpgpu_circuit *c);
int qpgpu_circuit_constants_sigmas_cap(const qpgpu_circuit *c, uint64_t *out, size_t out_words);
/* Zero-knowledge packs (standard_recursion_zk_config, reference common/src/circuit.rs:396-402): the wires,
 * Z/partial-products and quotient oracles carry 4 salt columns per leaf: a ChaCha20 stream keyed with 256 bits that are
 * drawn from the OS entropy source (getrandom) for every proof (the reference uses thread_rng, also a CSPRNG): opened
 * salts say nothing about unopened ones. TEST HOOK: a seed set here applies to the NEXT prove call only (proof b of a
 * batch derives its key from seed + b) and makes its bytes reproducible; a proof made under a known seed is not
 * zero-knowledge against whoever knows the seed. */
int qpgpu_circuit_set_blinding_seed(qpgpu_circuit *c, uint64_t seed);
/* Optional witness check (off by default): before the quotient stage, evaluate every filtered gate constraint on the
 * trace rows and the closing of the permutation product; a violation makes qpgpu_prove* return QPGPU_EUNSAT with the
 * offending row in qpgpu_last_error. plonky2 only finds an unsatisfied witness through debug assertions or a proof that
 * fails to verify (reference tests catch the panic, wormhole/tests/src/circuit/nullifier_tests.rs:53-58). Costs one
 * extra pass over the trace and one stream sync. */
int qpgpu_circuit_set_witness_check(qpgpu_circuit *c, int on);
size_t qpgpu_proof_size(const qpgpu_circuit *c);   /* bytes written by qpgpu_prove for this circuit */
/*
 * prove(): wires = the full witness matrix (num_wires x 2^degree_bits, column-major, as
 * `generate_partial_witness(...).full_witness()` holds it), public_inputs on the host. Writes
 * ProofWithPublicInputs::to_bytes() order into out: wires_cap, zs_partial_products_cap, quotient_cap, openings,
 * FRI commit caps, query rounds, final polynomial, pow_witness, then the public inputs. The proof-of-work nonce
 * is the minimum valid one (the reference's rayon find_any may return any).
 * The _dev form takes wires already resident in HBM and leaves them untouched.
 */
int qpgpu_prove(qpgpu_circuit *c, const uint64_t *wires, const uint64_t *public_inputs, uint8_t *out, size_t out_cap, size_t *out_len);
int qpgpu_prove_dev(qpgpu_circuit *c, const uint64_t *d_wires, const uint64_t *public_inputs, uint8_t *out, size_t out_cap, size_t *out_len);
/* `batch` (<= the handle's max_batch) independent proofs of the circuit in lockstep: every stage is launched once for all
 * of them, the Fiat-Shamir transcripts advance together on the host, one stream synchronisation per stage for the batch.
 * d_wires[b] / public_inputs[b] / outs[b] (each out_cap bytes) / out_lens[b] belong to proof b; witnesses laid out back to
 * back are used in place, others are gathered into the workspace first. Each proof's bytes equal what qpgpu_prove_dev
 * gives for the same witness. An error concerns the whole batch (an unsatisfied witness names its proof). */
int qpgpu_prove_batch_dev(qpgpu_circuit *c, const uint64_t *const *d_wires, uint32_t batch, const uint64_t *const *public_inputs,
                          uint8_t *const *outs, size_t out_cap, size_t *out_lens);

/* ---- stage s1: witness generation on the device ------------------------------------------------------------------
 * `iop::generator::generate_partial_witness` for the gate-attached generators of the supported gate set (Constant,
 * Arithmetic, ArithmeticExtension, MulExtension, BaseSum split, Poseidon, Reducing*, RandomAccess, Exponentiation,
 * PoseidonMds, CosetInterpolation) plus copy-constraint propagation. The dependency order is resolved once per circuit
 * from the pack (copy classes are the cycles of the sigma permutation); generation is one kernel launch per dependency
 * level. The caller supplies the cells no generator produces — plonky2's PartialWitness; qpgpu_witness_free_mask marks
 * them (1 byte per cell, num_wires x 2^degree_bits, column-major like the wire matrix) — every other cell is
 * overwritten. Generators that are not attached to a gate (Copy, Equality, WireSplit, extension Quotient, Constant,
 * NonzeroTest, LowHigh) run too when the pack carries them in its hint trailer (csrc/circuit.hpp); outputs of generators
 * the pack does not describe count as caller-supplied cells. */
int qpgpu_witness_info(qpgpu_circuit *c, uint64_t *num_generators, uint64_t *num_levels, uint64_t *num_free_cells);
int qpgpu_witness_free_mask(qpgpu_circuit *c, uint8_t *mask, size_t mask_len);
/* in place on a device-resident wire matrix / on a host matrix (uploaded, generated, downloaded, device copy scrubbed) */
int qpgpu_generate_witness_dev(qpgpu_circuit *c, uint64_t *d_wires, const uint64_t *public_inputs);
int qpgpu_generate_witness(qpgpu_circuit *c, uint64_t *wires, const uint64_t *public_inputs);
/* `batch` witnesses of the same circuit at once (wire matrices back to back, public inputs back to back): the dependency
 * levels are walked once for all of them, which is how the device pays off — a single witness is latency-bound by the
 * circuit's dependency depth, exactly the part a host core does well. */
/* the trace rows of one gate type (the QPCP gate type codes of csrc/circuit.hpp: 4 PoseidonGate, 14 the Poseidon2 gate, ...), for tools
 * and tests; rows_out may be NULL to ask for the count */
int qpgpu_circuit_gate_rows(const qpgpu_circuit *c, unsigned gate_type, uint32_t *rows_out, size_t cap, size_t *count);
int qpgpu_generate_witness_batch_dev(qpgpu_circuit *c, uint64_t *d_wires, uint32_t batch, const uint64_t *public_inputs);
/* The same from a sparse PartialWitness: `count` assignments (cell = row * num_wires + wire, value), as the reference
 * builds them with `pw.set_target` (wormhole/prover/src/lib.rs:187-221). d_wires (num_wires x 2^degree_bits, device) is
 * cleared, the assignments and the public inputs (at the pack's public-input cells, trailer "PUBI1") are written, the
 * generators run. A target set twice with different values — two assignments inside one copy class, or an assignment that
 * disagrees with the value a generator or the public-input argument gives that target — returns QPGPU_EUNSAT and names
 * the target in qpgpu_last_error: plonky2's "set twice with different values" panic, which six reference tests expect
 * (wormhole/tests/src/circuit/block_header_tests.rs:34-95, nullifier_tests.rs:53-58). Unassigned free cells stay zero.
 * A generator that divides by zero — QuotientGeneratorExtension's denominator, InterpolationGenerator's coset shift; plonky2's
 * Field::inverse panics with "Tried to invert zero" — returns QPGPU_EUNSAT too and names the zero target (every witness entry
 * point; it is reported before a "set twice" the value written in its place may have caused). */
int qpgpu_generate_witness_partial_dev(qpgpu_circuit *c, const uint64_t *cells, const uint64_t *values, size_t count,
                                       const uint64_t *public_inputs, uint64_t *d_wires);
/* `batch` PartialWitnesses over the SAME cell list at once (every proof of one circuit assigns the same targets: the 299 of
 * wormhole/prover/src/lib.rs:187-221): values is batch x count, public_inputs batch x num_public_inputs, d_wires batch wire
 * matrices back to back. The cell list is resolved against the circuit once and kept (copy classes, which assignments seed a
 * free class and which land in a generated one); per call the values go up in one piece, the dependency levels are walked
 * once for all witnesses, and one read-back tells which witnesses had a target set twice with different values — a caller's
 * assignment against a generated value, two assignments of one class, or two generators of one class that disagree (plonky2
 * allows several generators per partition as long as they agree: `connect(computed, claimed)`). status (may be NULL): per
 * witness QPGPU_OK or QPGPU_EUNSAT; the return value is QPGPU_EUNSAT when any witness failed, with the first one's target
 * in qpgpu_last_error. */
int qpgpu_generate_witness_partial_batch_dev(qpgpu_circuit *c, const uint64_t *cells, size_t count, const uint64_t *values,
                                             const uint64_t *public_inputs, uint32_t batch, uint64_t *d_wires, int *status);
/* Resolve a cell list and size every buffer for up to max_batch witnesses now (on the loading thread), so that the generate
 * calls themselves neither allocate nor free. Optional: the generate calls do it on first use. */
int qpgpu_witness_partial_prepare(qpgpu_circuit *c, const uint64_t *cells, size_t count, uint32_t max_batch);
/* The same with RandomValueGenerator targets drawn ON THE DEVICE: the LAST n_blinding cells of `cells` (the blinding rows' wires
 * of a zero-knowledge circuit, CircuitBuilder::blind; qpgpu_wrapper_circuit_build lists them) get one uniform field element
 * each, per witness, from ChaCha20 keyed with seeds[b] (32 bytes per witness; NULL: 32 bytes of operating-system entropy per
 * witness, the key wiped after use). values: [batch][count - n_blinding]. A blinding cell must be a free cell of its own
 * (QPGPU_EINVAL otherwise). Saves drawing and uploading ~0.6 M values per private-batch proof on the host. */
int qpgpu_generate_witness_partial_batch_blinded_dev(qpgpu_circuit *c, const uint64_t *cells, size_t count, size_t n_blinding, const uint64_t *values,
                                                     const uint8_t *seeds, const uint64_t *public_inputs, uint32_t batch, uint64_t *d_wires, int *status);
/* ProverCircuitData::prove takes its public inputs OUT of the generated witness (`get_targets(&prover_data.public_inputs)`). The
 * same here: pass public_inputs = NULL to the two functions above (allowed when the circuit's PublicInputGate wires are
 * copy-connected to the in-circuit hash of the public-input targets, as CircuitBuilder::build wires them: exported circuits and
 * the library's own builder; QPGPU_EINVAL otherwise) — the public-input targets are then whatever the caller's assignments and the
 * generators make of them — and read them back with this call ([batch][num_public_inputs]) for qpgpu_prove*_dev. */
int qpgpu_witness_public_inputs_dev(qpgpu_circuit *c, const uint64_t *d_wires, uint32_t batch, uint64_t *public_inputs_out);

/* ---- stage-level entry points: the circuit-independent parts of prove() ---------------------------------------
 * For a patched `qp-plonky2::plonk::prover::prove` that keeps witness generation, partial products and the quotient
 * evaluation (which depend on the gate set) in Rust and moves everything else to the GPU — polynomial commitments
 * (s2/s3), opening evaluations (s7) and the whole FRI opening proof (s8..s11). Replaces, call for call:
 *   PolynomialBatch::from_values / from_coeffs  -> qpgpu_oracle_commit
 *   batch.merkle_tree.cap                       -> qpgpu_oracle_cap
 *   batch.polynomials[i].to_extension().eval(z) -> qpgpu_oracle_eval           (OpeningSet::new)
 *   batch.get_lde_values(..)                    -> qpgpu_oracle_read / qpgpu_oracle_device_ptrs (quotient inputs)
 *   Challenger::{observe_*, get_*}              -> qpgpu_challenger_*         (host; optional, the Rust one works too)
 *   PolynomialBatch::prove_openings             -> qpgpu_fri_prove            (returns write_fri_proof bytes)
 * reached in the reference through wormhole/prover/src/lib.rs:171-175 and the aggregator call sites listed above. */
typedef struct qpgpu_oracle qpgpu_oracle;
#define QPGPU_ORACLE_VALUES 0u        /* input polynomials are values on the subgroup (from_values) */
#define QPGPU_ORACLE_COEFFS 1u        /* input polynomials are coefficients (from_coeffs) */
#define QPGPU_ORACLE_BLINDING 2u      /* append 4 salt elements to every leaf (zero-knowledge configs) */
#define QPGPU_ORACLE_DEVICE_INPUT 4u  /* `polys` is a device pointer */
/* polys: num_polys x 2^degree_bits, column-major (polynomial j at polys + j*2^degree_bits). With QPGPU_ORACLE_BLINDING the
 * salts are a ChaCha20 stream: blinding_seed 0 = keyed with 256 fresh bits from the OS entropy source (production);
 * non-zero = key derived from the seed (reproducible, tests only). blinding_stream selects the stream under the key
 * (the all-in-one prover uses 1, 2, 3 for wires, Z/partial products, quotient). */
int qpgpu_oracle_commit(qpgpu_ctx *ctx, const uint64_t *polys, uint32_t num_polys, unsigned degree_bits, unsigned rate_bits,
                        unsigned cap_height, unsigned flags, uint64_t blinding_seed, uint32_t blinding_stream, qpgpu_oracle **out);
void qpgpu_oracle_free(qpgpu_oracle *o);   /* overwrites the device copies before releasing them */
int qpgpu_oracle_cap(const qpgpu_oracle *o, uint64_t *out, size_t out_words);          /* 4 << cap_height words */
/* evaluations of polynomials [first, first+count) at an extension point; out: count x 2 words */
int qpgpu_oracle_eval(qpgpu_oracle *o, const uint64_t point[2], uint32_t first, uint32_t count, uint64_t *out);
#define QPGPU_ORACLE_READ_COEFFS 0u   /* count x 2^degree_bits words */
#define QPGPU_ORACLE_READ_LDE 1u      /* count x 2^(degree_bits+rate_bits) words, slot s = value at g*w^bitrev(s): the Merkle leaf order */
int qpgpu_oracle_read(qpgpu_oracle *o, unsigned what, uint32_t first, uint32_t count, uint64_t *out);
/* device pointers of the resident data, same layouts as qpgpu_oracle_read (digests: leaf layer first, cap last) */
int qpgpu_oracle_device_ptrs(const qpgpu_oracle *o, const uint64_t **d_coeffs, const uint64_t **d_lde, const uint64_t **d_digests);

/* plonky2::iop::challenger::Challenger as plain data: buffers are Vec<F> contents, lengths their len() */
typedef struct {
    uint64_t sponge_state[12];
    uint64_t input_buffer[8];
    uint64_t output_buffer[8];
    uint32_t input_len, output_len;
} qpgpu_challenger;
void qpgpu_challenger_init(qpgpu_challenger *c);
/* under the context's hasher; QPGPU_EINVAL for a state plonky2 cannot be in (input_len >= 8 or output_len > 8) */
int qpgpu_ctx_challenger_observe(const qpgpu_ctx *ctx, qpgpu_challenger *c, const uint64_t *elements, size_t n);
int qpgpu_ctx_challenger_get(const qpgpu_ctx *ctx, qpgpu_challenger *c, uint64_t *out);
/* under the process-default hasher (first ABI); an invalid state is left untouched and get returns 0 */
void qpgpu_challenger_observe(qpgpu_challenger *c, const uint64_t *elements, size_t n);
uint64_t qpgpu_challenger_get(qpgpu_challenger *c);

typedef struct { uint32_t oracle, first, count; } qpgpu_fri_range;          /* FriPolynomialInfo::from_range */
typedef struct { uint64_t point[2]; uint32_t num_ranges; qpgpu_fri_range ranges[8]; } qpgpu_fri_batch;   /* FriBatchInfo */
typedef struct {                                                            /* FriParams */
    uint32_t rate_bits, cap_height, proof_of_work_bits, num_query_rounds;
    uint32_t num_reduction_rounds; uint32_t reduction_arity_bits[16];
} qpgpu_fri_params;
size_t qpgpu_fri_proof_size(qpgpu_oracle *const *oracles, uint32_t num_oracles, const qpgpu_fri_params *params);
/* PolynomialBatch::prove_openings(instance, oracles, challenger, fri_params): `challenger` is the transcript state after
 * the openings were observed and comes back as prove_openings leaves it. Writes the FriProof in write_fri_proof order
 * (commit-phase caps, query rounds, final polynomial, pow witness; minimum valid nonce). */
int qpgpu_fri_prove(qpgpu_ctx *ctx, qpgpu_oracle *const *oracles, uint32_t num_oracles, const qpgpu_fri_batch *batches,
                    uint32_t num_batches, const qpgpu_fri_params *params, qpgpu_challenger *challenger,
                    uint8_t *out, size_t out_cap, size_t *out_len);

/* ---- proving pool: several proofs of one circuit in flight on one GPU ------------------------------------------
 * The native counterpart of the reference's dedicated proving worker (wormhole/aggregator/src/aggregator.rs:14-43):
 * `workers` host threads, each with its own context (HIP stream) and loaded copy of the circuit, fed from one queue.
 * One proof leaves most of the GPU idle between its latency-bound stages; four workers reach the throughput plateau.
 * submit() never blocks; wait() blocks until that proof is written (tickets may be waited for in any order, once). The
 * witness matrices and output buffers must stay valid until their ticket has been waited for. */
typedef struct qpgpu_pool qpgpu_pool;
int qpgpu_pool_create(int device, const uint64_t *pack_words, size_t n_words, unsigned workers, qpgpu_pool **out);
/* Workers that prove in lockstep: each takes up to max_batch queued proofs at once (qpgpu_prove_batch_dev). Two workers of
 * eight keep the GPU busy while one of them is in a host-side stage. The workers' contexts use the process-default hasher. */
int qpgpu_pool_create_batched(int device, const uint64_t *pack_words, size_t n_words, unsigned workers, unsigned max_batch, qpgpu_pool **out);
/* The same over several GPUs of one process — north_star's "proofs shard one-per-GPU across the node ... invoked from Rust
 * through a thin C-ABI": devices[0..n_devices) (a device may be named twice: two worker sets on one GPU), workers_per_device
 * workers on each, ONE queue. Proofs are independent, so there is no data-path collective: whichever
 * worker takes a job writes the proof into the caller's host buffer, which is the whole "gather of proof bytes" inside one
 * process (RCCL is for bench.py's one-process-per-GPU launch). The level schedule of an aggregation tree
 * (wormhole/aggregator/src/aggregator.rs:187-227) stays the caller's: submit a level's proofs, wait for them, build the next.
 * flags: QPGPU_POOL_HOST_WITNESS gives every worker a workspace of max_batch wire matrices, which qpgpu_pool_submit_host needs
 * (qpgpu_pool_set_partial_cells allocates it too). */
#define QPGPU_POOL_HOST_WITNESS 1u
int qpgpu_pool_create_multi(const int *devices, unsigned n_devices, const uint64_t *pack_words, size_t n_words, unsigned workers_per_device,
                            unsigned max_batch, unsigned flags, qpgpu_pool **out);
/* A job's arguments are checked at submit (null pointers: QPGPU_EINVAL; out_cap below qpgpu_pool_proof_size:
 * QPGPU_EBUFSIZE). Jobs of different callers share a lockstep batch: when a batch fails, its jobs are proven again one at
 * a time, so only the failing job's wait() returns the error (with that job's own message). */
void qpgpu_pool_destroy(qpgpu_pool *p);          /* drains the queue first; witness workspaces are overwritten before release */
/* qpgpu_circuit_set_witness_check on every worker's circuit; only while no ticket is outstanding (QPGPU_EINVAL otherwise) */
int qpgpu_pool_set_witness_check(qpgpu_pool *p, int on);
size_t qpgpu_pool_proof_size(const qpgpu_pool *p);
unsigned qpgpu_pool_workers(const qpgpu_pool *p);
unsigned qpgpu_pool_devices(const qpgpu_pool *p);
/* 1 when the pool's workers take turns on the device: a queue-intercepting profiler (rocprofv3) was found in the process at creation
 * (its interceptor has faulted under concurrent submission from several threads), or QPGPU_POOL_SERIALIZE=1 */
int qpgpu_pool_serialized(const qpgpu_pool *p);
const char *qpgpu_pool_last_error(const qpgpu_pool *p);
/* Three forms of a job's witness. (1) a full wire matrix resident on a device: only that device's workers can read it —
 * qpgpu_pool_submit_on names the device by its index in `devices` (qpgpu_pool_submit = index 0). */
int qpgpu_pool_submit(qpgpu_pool *p, const uint64_t *d_wires, const uint64_t *public_inputs, uint8_t *out, size_t out_cap, uint64_t *ticket);
int qpgpu_pool_submit_on(qpgpu_pool *p, unsigned device_index, const uint64_t *d_wires, const uint64_t *public_inputs, uint8_t *out, size_t out_cap, uint64_t *ticket);
/* (2) a full wire matrix in host memory — what a patched qp-plonky2 prove() holds after
 * `generate_partial_witness(..).full_witness()` — taken by any worker of any device and uploaded on that worker's stream
 * (about 9 MB per proof at 2^13 rows; over PCIe). The matrix must stay valid until the ticket has been waited for. */
int qpgpu_pool_submit_host(qpgpu_pool *p, const uint64_t *wires, const uint64_t *public_inputs, uint8_t *out, size_t out_cap, uint64_t *ticket);
/* (3) a PartialWitness: the values of the cell list set once with qpgpu_pool_set_partial_cells (for the leaf circuit:
 * qpgpu_leaf_commit's cells / values, i.e. WormholeProver::commit, wormhole/prover/src/lib.rs:156-163). The worker that takes
 * the job runs stage s1 for its whole lockstep batch (qpgpu_generate_witness_partial_batch_dev) and then stages s2..s12: this
 * is the reference bench's timed region, `prover.commit(&inputs).unwrap().prove()` (wormhole/prover/benches/prover.rs:38),
 * from the committed assignments on. `values` (as many words as the cell list) are copied at submit and wiped after use; an
 * unsatisfiable witness fails its own ticket with QPGPU_EUNSAT and the target's name, the rest of its batch is proven. */
int qpgpu_pool_set_partial_cells(qpgpu_pool *p, const uint64_t *cells, size_t count);
/* The same for a zero-knowledge circuit: the LAST n_blinding cells are the blinding rows' random wires, drawn on the device for
 * every proof (qpgpu_generate_witness_partial_batch_blinded_dev, operating-system entropy); a job's `values` then has
 * count - n_blinding words. public_inputs may be NULL in qpgpu_pool_submit_partial: the job is proven with the public inputs its
 * witness holds (they are the last num_public_inputs words of the proof) — the batch layers, whose public inputs the circuit
 * computes; public inputs that ARE handed in are compared with the witness's (QPGPU_EUNSAT when they differ). */
int qpgpu_pool_set_partial_cells_blinded(qpgpu_pool *p, const uint64_t *cells, size_t count, size_t n_blinding);
int qpgpu_pool_submit_partial(qpgpu_pool *p, const uint64_t *values, const uint64_t *public_inputs, uint8_t *out, size_t out_cap, uint64_t *ticket);
int qpgpu_pool_wait(qpgpu_pool *p, uint64_t ticket, size_t *out_len);

/* Hash constants the library derives at start-up (host only, no GPU): the 360 Poseidon round constants and plonky2's
 * FAST_PARTIAL_* tables flattened as FIRST[12] | RC[22] | VS[22][11] | W_HATS[22][11] | INIT[11][11] (row c of INIT
 * produces element 1+c). Returns the number of words of the second table. */
size_t qpgpu_poseidon_constants(uint64_t *round_constants_360, uint64_t *fast_partial, size_t fast_partial_cap);

/* ---- synthetic circuits (stand-in for reference rows a1/a6 while no Rust exporter exists) ---- */
/*
 * Builds a satisfied plonky2-shaped circuit (PublicInput / Constant / Arithmetic / Noop gates wired by copy
 * constraints, standard_recursion_config parameters) and its witness. pack_out receives the circuit pack
 * ("QPCP1", see qp-zk-circuits_amd/csrc/circuit.hpp); wires_out num_wires x 2^degree_bits column-major;
 * pis_out num_public_inputs elements. Host-only; no GPU needed.
 */
size_t qpgpu_synth_pack_words(unsigned degree_bits, unsigned num_wires, unsigned num_routed);
/* flags bit 0: every 8th row is a PoseidonGate row (135 wires, 123 constraints of degree 7, its own selector group),
 * the gate that dominates the recursive (aggregator) circuits. bit 1: every 8th row is a BaseSumGate<2> row (the leaf
 * circuit's range checks, reference wormhole/circuit/src/zk_merkle_proof.rs:486-504). bit 2: every 8th row alternates
 * ArithmeticExtensionGate / MulExtensionGate (quadratic-extension arithmetic of the recursive verifier circuits).
 * bit 3: every 8th row cycles through ReducingGate, ReducingExtensionGate, RandomAccessGate(4 bits), ExponentiationGate
 * and PoseidonMdsGate, the rest of the gates plonky2's in-circuit verifier uses (needs >= 48 routed wires). bit 4: some
 * operation inputs come from generators that are not attached to a gate (Equality, LowHigh, NonzeroTest, Constant, Copy,
 * WireSplit, extension quotient): the pack gets a hint trailer (circuit.hpp) and qpgpu_synth_pack_words_ex is an upper bound.
 * bit 6: the LEAF PROFILE — the Wormhole leaf circuit's application hashes as rows of the qp fork's Poseidon2 gate (gate type 14,
 * qp-zk-circuits_amd/csrc/circuit.hpp): pad-10 sponge chains over 7, 4, 9, 4, 8, 45 and 16 x 16 elements (the eight
 * `hash_n_to_hash_no_pad_p2` call sites: wormhole/circuit/src/unspendable_account.rs:229-231, nullifier.rs:298-299,
 * zk_merkle_proof.rs:482,504,606, block_header/mod.rs:66; 61 permutations), one gate row per permutation and one ArithmeticGate
 * row of eight additions per further block, as many as the rows hold, plus four free-standing permutation rows; the pack
 * carries the gate's wire layout ("P2GL1"). bit 7 (with bit 6): a deliberately different wire layout (no swap wires, other block
 * order) to exercise the layout table. qpgpu_synth_p2_sites lists the hash sites (3 words each: preimage length, gate rows,
 * first slot; gate row k of a site is row 8 * (slot + k) + 3, the additions for block k + 1 are operations 0..7 of the row after
 * it, message element i is input wire i of the first gate row or the addend wire 4 (i % 8) + 2 of the add row before block i / 8,
 * the digest is the first four output wires of the last gate row). */
size_t qpgpu_synth_p2_sites(unsigned degree_bits, unsigned num_public_inputs, unsigned flags, uint64_t *out, size_t cap_words);
size_t qpgpu_synth_pack_words_ex(unsigned degree_bits, unsigned num_wires, unsigned num_routed, unsigned flags);
int qpgpu_synth_circuit_ex(unsigned degree_bits, unsigned num_wires, unsigned num_routed, unsigned num_public_inputs,
                           uint64_t seed, unsigned flags, uint64_t *pack_out, size_t pack_cap_words, size_t *pack_words,
                           uint64_t *wires_out, uint64_t *pis_out);
int qpgpu_synth_circuit(unsigned degree_bits, unsigned num_wires, unsigned num_routed, unsigned num_public_inputs,
                        uint64_t seed, uint64_t *pack_out, size_t pack_cap_words, size_t *pack_words,
                        uint64_t *wires_out, uint64_t *pis_out);

#ifdef __cplusplus
}
#endif
#endif
