//! qpgpu_backend.rs — the Rust side of the MI355X prover backend: FFI declarations for `libqpgpu.so`, the in-process
//! circuit-pack exporter, the public-input / target maps, and the `prove` hook.
//!
//! STATUS: NOT COMPILED in the build environment of this repository (it has no Rust toolchain and no copy of
//! `qp-plonky2 =1.5.5`). It is written against the plonky2 1.x public API the fork re-exports (field and type names as
//! in upstream `plonk::circuit_data`, `iop::target`, `gates::selectors`); every place that depends on a detail of the
//! fork that this repository could not read is marked `FORK:`. The C side it talks to is tested without it
//! (`tests/`, through the same C ABI from Python and from C).
//!
//! Where it goes: a module of the patched `qp-plonky2` (`src/gpu/mod.rs`), or a small crate next to
//! `wormhole/prover` that depends on `qp-plonky2`. Nothing in the wormhole crates changes their signatures:
//!   * `WormholeProver::new` (wormhole/prover/src/lib.rs:112-140) builds the circuit from source as today and then calls
//!     `GpuCircuit::from_circuit_data(&circuit_data)` once;
//!   * `WormholeProver::prove` (wormhole/prover/src/lib.rs:171-175) calls `gpu.prove(&self.circuit_data, self.partial_witness)`
//!     in place of `self.circuit_data.prove(..)`; same `anyhow::Result<ProofWithPublicInputs<F, C, D>>`;
//!   * the batch provers (wormhole/aggregator/src/private_batch/prover/lib.rs:326-343,
//!     public_batch/prover/lib.rs:301-305) do the same with their own circuit data.
//!
//! Policy note (wormhole/circuit/src/circuit.rs:5-16, wormhole/circuit-builder/src/lib.rs:34-36): the reference never
//! ships prover data as a file ("a poisoned prover artifact cannot exfiltrate witness data through the proof's
//! public-input list"). The pack below is therefore derived IN PROCESS from the circuit the prover has just built and
//! handed to the library as a byte buffer; it is not written to the bins directory. (`QPGPU_ARTIFACT_PROVER_PACK` file
//! names in include/qpgpu_wire.h exist for test fixtures and benchmarks only.)

use anyhow::{anyhow, bail, Result};
use core::ffi::{c_char, c_void};
use plonky2::field::extension::Extendable;
use plonky2::field::polynomial::PolynomialCoeffs;
use plonky2::field::types::PrimeField64;
use plonky2::hash::hash_types::RichField;
use plonky2::iop::target::Target;
use plonky2::iop::witness::{PartialWitness, PartitionWitness};
use plonky2::plonk::circuit_data::{CircuitData, CommonCircuitData, ProverOnlyCircuitData};
use plonky2::plonk::config::GenericConfig;
use plonky2::plonk::proof::ProofWithPublicInputs;

// ---------------------------------------------------------------------------------------------------------------------
// FFI: mirrors include/qpgpu.h (the subset the all-in-one path needs; the stage-level entry points are in INTEGRATION.md)
// ---------------------------------------------------------------------------------------------------------------------
#[repr(C)] pub struct QpgpuCtx { _p: [u8; 0] }
#[repr(C)] pub struct QpgpuCircuit { _p: [u8; 0] }

pub const QPGPU_OK: i32 = 0;
pub const QPGPU_EUNSAT: i32 = -4;

#[link(name = "qpgpu")]
extern "C" {
    pub fn qpgpu_ctx_create(device: i32, out: *mut *mut QpgpuCtx) -> i32;
    pub fn qpgpu_ctx_destroy(ctx: *mut QpgpuCtx);
    pub fn qpgpu_last_error(ctx: *const QpgpuCtx) -> *const c_char;
    pub fn qpgpu_ctx_pci_bus_id(ctx: *const QpgpuCtx, out: *mut c_char, out_len: usize) -> i32;
    pub fn qpgpu_ctx_set_hasher(ctx: *mut QpgpuCtx, kind: i32, params: *const u64, n_words: usize) -> i32;
    pub fn qpgpu_malloc(ctx: *mut QpgpuCtx, bytes: usize, dptr: *mut *mut c_void) -> i32;
    pub fn qpgpu_free(ctx: *mut QpgpuCtx, dptr: *mut c_void) -> i32;
    pub fn qpgpu_free_scrubbed(ctx: *mut QpgpuCtx, dptr: *mut c_void, bytes: usize) -> i32;
    pub fn qpgpu_memcpy_h2d(ctx: *mut QpgpuCtx, dst: *mut c_void, src: *const c_void, bytes: usize) -> i32;
    pub fn qpgpu_circuit_load_batch(ctx: *mut QpgpuCtx, pack: *const u64, n_words: usize, max_batch: u32, out: *mut *mut QpgpuCircuit) -> i32;
    pub fn qpgpu_circuit_free(c: *mut QpgpuCircuit);
    pub fn qpgpu_circuit_scrub(c: *mut QpgpuCircuit) -> i32;
    pub fn qpgpu_circuit_set_witness_check(c: *mut QpgpuCircuit, on: i32) -> i32;
    pub fn qpgpu_proof_size(c: *const QpgpuCircuit) -> usize;
    pub fn qpgpu_prove(c: *mut QpgpuCircuit, wires: *const u64, public_inputs: *const u64, out: *mut u8, out_cap: usize, out_len: *mut usize) -> i32;
    pub fn qpgpu_generate_witness_partial_dev(c: *mut QpgpuCircuit, cells: *const u64, values: *const u64, count: usize,
                                              public_inputs: *const u64, d_wires: *mut u64) -> i32;
    pub fn qpgpu_prove_dev(c: *mut QpgpuCircuit, d_wires: *const u64, public_inputs: *const u64, out: *mut u8, out_cap: usize, out_len: *mut usize) -> i32;
    pub fn qpgpu_prove_batch_dev(c: *mut QpgpuCircuit, d_wires: *const *const u64, batch: u32, public_inputs: *const *const u64,
                                 outs: *const *mut u8, out_cap: usize, out_lens: *mut usize) -> i32;
    pub fn qpgpu_pack_validate(pack: *const u64, n_words: usize, err: *mut c_char) -> i32;
    pub fn qpgpu_circuit_constants_sigmas_cap(c: *const QpgpuCircuit, out: *mut u64, out_words: usize) -> i32;
    // include/qpgpu_verify.h — host-side verification of the bytes the GPU produced (debug self-check; production keeps
    // plonky2's own `VerifierCircuitData::verify`, which is what consumers of the proof run anyway)
    pub fn qpgpu_verifier_create(pack: *const u64, n_words: usize, cs_cap: *const u64, cap_words: usize, hasher_kind: i32,
                                 hasher_params: *const u64, n_params: usize, out: *mut *mut QpgpuVerifier, err: *mut c_char) -> i32;
    pub fn qpgpu_verifier_free(v: *mut QpgpuVerifier);
    pub fn qpgpu_verifier_verify(v: *const QpgpuVerifier, proof: *const u8, len: usize, err: *mut c_char) -> i32;
    // include/qpgpu.h — the proving pool over one GPU or several: one queue, a worker set per device,
    // proofs written into the caller's host buffers
    pub fn qpgpu_pool_create_multi(devices: *const i32, n_devices: u32, pack: *const u64, n_words: usize, workers_per_device: u32,
                                   max_batch: u32, flags: u32, out: *mut *mut QpgpuPool) -> i32;
    pub fn qpgpu_pool_destroy(p: *mut QpgpuPool);
    pub fn qpgpu_pool_proof_size(p: *const QpgpuPool) -> usize;
    pub fn qpgpu_pool_last_error(p: *const QpgpuPool) -> *const c_char;
    pub fn qpgpu_pool_submit_host(p: *mut QpgpuPool, wires: *const u64, public_inputs: *const u64, out: *mut u8, out_cap: usize, ticket: *mut u64) -> i32;
    pub fn qpgpu_pool_set_partial_cells(p: *mut QpgpuPool, cells: *const u64, count: usize) -> i32;
    pub fn qpgpu_pool_set_partial_cells_blinded(p: *mut QpgpuPool, cells: *const u64, count: usize, n_blinding: usize) -> i32;
    pub fn qpgpu_pool_submit_partial(p: *mut QpgpuPool, values: *const u64, public_inputs: *const u64, out: *mut u8, out_cap: usize, ticket: *mut u64) -> i32;
    pub fn qpgpu_pool_wait(p: *mut QpgpuPool, ticket: u64, out_len: *mut usize) -> i32;
    // include/qpgpu.h, include/qpgpu_batch.h — zero-knowledge circuits (the private-batch layer): the blinding rows' random wires are
    // the last `n_blinding` cells of the assignment list and are drawn on the device, one ChaCha20 key per witness (seeds = null:
    // operating-system entropy); qpgpu_random_field_elements is the host-side form
    pub fn qpgpu_generate_witness_partial_batch_blinded_dev(c: *mut QpgpuCircuit, cells: *const u64, count: usize, n_blinding: usize, values: *const u64,
                                                            seeds: *const u8, public_inputs: *const u64, batch: u32, d_wires: *mut u64, status: *mut i32) -> i32;
    pub fn qpgpu_random_field_elements(seed32: *const u8, out: *mut u64, n: usize, err: *mut c_char) -> i32;
    // with stage s1 on the device the public inputs are read out of the device witness, the way `prove` reads them out of the
    // partition witness: pass public_inputs = null to the generate_witness_partial* calls, then this, then qpgpu_prove*_dev
    pub fn qpgpu_circuit_gate_rows(c: *const QpgpuCircuit, gate_type: u32, rows_out: *mut u32, cap: usize, count: *mut usize) -> i32;   // introspection
    pub fn qpgpu_witness_public_inputs_dev(c: *mut QpgpuCircuit, d_wires: *const u64, batch: u32, public_inputs_out: *mut u64) -> i32;
    // include/qpgpu_leaf.h, include/qpgpu_batch.h — the circuits built by the library itself: only for
    // deployments that take BOTH prover and verifier data from it; with exported packs of the fork's own circuits these are not used
    pub fn qpgpu_leaf_circuit_build(fragment: u32, min_degree_bits: u32, inner_hasher: i32, p2_layout: *const u64, pack_out: *mut u64, pack_cap_words: usize,
                                    pack_words: *mut usize, target_map_out: *mut u64, info_out: *mut u64, err: *mut c_char) -> i32;
    // optional: the leaf circuit's hash-chain states as extra assignments (796 cells / values appended to qpgpu_leaf_commit's) — the
    // same witness in 14 dependency levels instead of 120 (one proof 4.6 -> 3.7 ms); a hint that disagrees is QPGPU_EUNSAT
    pub fn qpgpu_leaf_circuit_hash_hint_cells(min_degree_bits: u32, inner_hasher: i32, p2_layout: *const u64, cells_out: *mut u64, cap: usize,
                                              count: *mut usize, err: *mut c_char) -> i32;
    pub fn qpgpu_leaf_hash_hints(inputs: *const c_void, values_out: *mut u64, cap: usize, count: *mut usize, err: *mut c_char) -> i32;
    pub fn qpgpu_wrapper_circuit_build(inner_pack: *const u64, inner_words: usize, inner_cs_cap: *const u64, cap_words: usize, num_proofs: u32,
                                       num_routed_wires: u32, min_degree_bits: u32, inner_hasher: i32, flags: u32, pack_out: *mut u64, pack_cap_words: usize,
                                       pack_words: *mut usize, target_map_out: *mut u64, map_cap: usize, map_count: *mut usize, info_out: *mut u64,
                                       err: *mut c_char) -> i32;
}
#[repr(C)] pub struct QpgpuVerifier { _private: [u8; 0] }
#[repr(C)] pub struct QpgpuPool { _private: [u8; 0] }
pub const QPGPU_POOL_HOST_WITNESS: u32 = 1;

/// All the GPUs of a node behind one handle: what `ProvingContext` (wormhole/aggregator/src/aggregator.rs:187-227) or a miner's
/// leaf-proving loop holds instead of a `ProverCircuitData`. Proofs are independent, so there is no collective: any worker of any
/// device takes the next job and writes the proof bytes into the caller's buffer.
pub struct GpuProvingPool { pool: *mut QpgpuPool, proof_size: usize }
unsafe impl Send for GpuProvingPool {}
unsafe impl Sync for GpuProvingPool {}      // submit / wait lock inside the library

impl GpuProvingPool {
    /// `devices`: HIP device ordinals, e.g. `&[0, 1, 2, 3, 4, 5, 6, 7]`; three workers of 32 lockstep proofs per device saturate one.
    pub fn new(pack: &[u64], devices: &[i32], workers_per_device: u32, lockstep: u32) -> Result<Self> {
        let mut pool = std::ptr::null_mut();
        let rc = unsafe { qpgpu_pool_create_multi(devices.as_ptr(), devices.len() as u32, pack.as_ptr(), pack.len(), workers_per_device, lockstep,
                                                  QPGPU_POOL_HOST_WITNESS, &mut pool) };
        if rc != 0 { bail!("qpgpu_pool_create_multi failed ({rc})") }
        Ok(Self { pool, proof_size: unsafe { qpgpu_pool_proof_size(pool) } })
    }
    /// `full_witness`: the wire matrix `generate_partial_witness(..).full_witness()` holds (num_wires x n, column-major, canonical u64).
    /// Returns a ticket; the matrix and `out` must outlive `wait`.
    pub fn submit(&self, full_witness: &[u64], public_inputs: &[u64], out: &mut [u8]) -> Result<u64> {
        let mut ticket = 0u64;
        let rc = unsafe { qpgpu_pool_submit_host(self.pool, full_witness.as_ptr(), public_inputs.as_ptr(), out.as_mut_ptr(), out.len(), &mut ticket) };
        if rc != 0 { bail!("qpgpu_pool_submit_host: {}", unsafe { std::ffi::CStr::from_ptr(qpgpu_pool_last_error(self.pool)).to_string_lossy() }) }
        Ok(ticket)
    }
    /// Blocks until that proof is written; an unsatisfied witness fails its own ticket only ("Failed to prove: ...").
    pub fn wait(&self, ticket: u64) -> Result<usize> {
        let mut len = 0usize;
        let rc = unsafe { qpgpu_pool_wait(self.pool, ticket, &mut len) };
        if rc != 0 { bail!("Failed to prove: {}", unsafe { std::ffi::CStr::from_ptr(qpgpu_pool_last_error(self.pool)).to_string_lossy() }) }
        Ok(len)
    }
    pub fn proof_size(&self) -> usize { self.proof_size }

    /// A circuit whose witness is a PartialWitness (stage s1 on the device too): `cells` is the assignment list resolved once —
    /// for an exported circuit the wire cells of the targets `fill_witness` / `fill_private_batch_witness` set, in their order;
    /// the last `n_blinding` of them are the `RandomValueGenerator` targets of a zero-knowledge circuit (`CircuitBuilder::blind`),
    /// which the device draws per proof.
    pub fn set_partial_cells(&self, cells: &[u64], n_blinding: usize) -> Result<()> {
        let rc = unsafe { qpgpu_pool_set_partial_cells_blinded(self.pool, cells.as_ptr(), cells.len(), n_blinding) };
        if rc != 0 { bail!("qpgpu_pool_set_partial_cells: {}", unsafe { std::ffi::CStr::from_ptr(qpgpu_pool_last_error(self.pool)).to_string_lossy() }) }
        Ok(())
    }
    /// `values`: one per cell of the list, without the blinding ones. `public_inputs = None`: what `ProverCircuitData::prove` does —
    /// they are read out of the generated witness (the batch layers, whose public inputs the circuit computes) and come back as the
    /// last `num_public_inputs` words of the proof.
    pub fn submit_partial(&self, values: &[u64], public_inputs: Option<&[u64]>, out: &mut [u8]) -> Result<u64> {
        let mut ticket = 0u64;
        let pis = public_inputs.map_or(std::ptr::null(), |p| p.as_ptr());
        let rc = unsafe { qpgpu_pool_submit_partial(self.pool, values.as_ptr(), pis, out.as_mut_ptr(), out.len(), &mut ticket) };
        if rc != 0 { bail!("qpgpu_pool_submit_partial: {}", unsafe { std::ffi::CStr::from_ptr(qpgpu_pool_last_error(self.pool)).to_string_lossy() }) }
        Ok(ticket)
    }
}
impl Drop for GpuProvingPool { fn drop(&mut self) { unsafe { qpgpu_pool_destroy(self.pool) } } }

fn last_error(ctx: *const QpgpuCtx) -> String {
    unsafe { std::ffi::CStr::from_ptr(qpgpu_last_error(ctx)).to_string_lossy().into_owned() }
}

// ---------------------------------------------------------------------------------------------------------------------
// Circuit pack ("QPCP1", qp-zk-circuits_amd/csrc/circuit.hpp): everything `prove` reads after witness generation
// ---------------------------------------------------------------------------------------------------------------------
const QPCP_MAGIC: u64 = 0x0000_0031_5043_5051;
const PUBI_MAGIC: u64 = 0x0000_0031_4942_5550;
const P2GL_MAGIC: u64 = 0x0000_0031_4C47_3250;     // "P2GL1": wire layout of the Poseidon2 gate (csrc/circuit.hpp)
const P2_NO_SWAP: u64 = 0xFFFF_FFFF;

/// Gate id string -> (type code, param0, param1, param2) of the pack format. An id this table does not know aborts the
/// export: the backend refuses unknown gates rather than skipping their constraints.
fn gate_code(id: &str) -> Result<(u64, u64, u64, u64)> {
    // ids are the `Debug` renderings plonky2 uses as gate identity, e.g. "ArithmeticGate { num_ops: 20 }"
    let num = |key: &str| -> Result<u64> {
        let at = id.find(key).ok_or_else(|| anyhow!("gate id `{id}` has no `{key}`"))? + key.len();
        let digits: String = id[at..].chars().skip_while(|c| !c.is_ascii_digit()).take_while(|c| c.is_ascii_digit()).collect();
        digits.parse::<u64>().map_err(|_| anyhow!("gate id `{id}`: no number after `{key}`"))
    };
    Ok(if id.starts_with("NoopGate") { (0, 0, 0, 0) }
    else if id.starts_with("ConstantGate") { (1, num("num_consts")?, 0, 0) }
    else if id.starts_with("PublicInputGate") { (2, 0, 0, 0) }
    else if id.starts_with("ArithmeticGate") { (3, num("num_ops")?, 0, 0) }
    else if id.starts_with("PoseidonGate") { (4, 0, 0, 0) }
    else if id.starts_with("BaseSumGate") {
        if !id.ends_with("Base: 2") { bail!("only BaseSumGate<2> is implemented on the device (`{id}`)") }
        (5, num("num_limbs")?, 2, 0)
    }
    else if id.starts_with("ArithmeticExtensionGate") { (6, num("num_ops")?, 0, 0) }
    else if id.starts_with("MulExtensionGate") { (7, num("num_ops")?, 0, 0) }
    else if id.starts_with("ReducingGate") { (8, num("num_coeffs")?, 0, 0) }
    else if id.starts_with("ReducingExtensionGate") { (9, num("num_coeffs")?, 0, 0) }
    else if id.starts_with("RandomAccessGate") {
        // num_extra_constants = min(num_constants, num_routed_wires - (2 + 2^bits) * num_copies): derived by the caller
        (10, num("bits")?, num("num_copies")?, num("num_extra_constants").unwrap_or(0))
    }
    else if id.starts_with("ExponentiationGate") { (11, num("num_power_bits")?, 0, 0) }
    else if id.starts_with("PoseidonMdsGate") { (12, 0, 0, 0) }
    else if id.starts_with("CosetInterpolationGate") { (13, num("subgroup_bits")?, num("degree")?, 0) }
    // FORK: the qp fork's Poseidon2 gate (wormhole/circuit reaches it through `hash_n_to_hash_no_pad_p2::<Poseidon2Hash>`).
    // The device evaluates it as gate type 14 with the wire layout this exporter writes into the "P2GL1" trailer
    // (`poseidon2_gate_layout` below); adjust the prefix if the fork's `Gate::id()` renders differently.
    else if id.starts_with("Poseidon2Gate") { (14, 0, 0, 0) }
    else { bail!("gate `{id}` is not implemented by the MI355X backend") })
}

/// The ten words of the "P2GL1" trailer, read off the fork's gate: first input wire, first output wire, swap wire (or
/// P2_NO_SWAP), first delta wire, first S-box-input wire of the first half's full rounds, of the 22 partial rounds, of the
/// second half's full rounds, whether round 0 of the first half is recorded too, constraint order (0 = swap boolean, deltas,
/// first-half rounds, partial rounds, second-half rounds, outputs), one past the last wire used.
/// FORK: written against the accessor names of upstream's `PoseidonGate` (`wire_input`, `wire_output`, `WIRE_SWAP`,
/// `wire_delta`, `wire_full_sbox_0`, `wire_partial_sbox`, `wire_full_sbox_1`, `end`) carried over to `Poseidon2Gate`; if the
/// fork names or orders them differently, this function is the single place to change — the device, the host verifier and the
/// oracle all read the table (`tests/test_poseidon2_gate*.py` exercise a second layout to prove it). If the fork's gate
/// constrains in another ORDER than upstream's PoseidonGate, add a new `constraint_order` value on both sides.
fn poseidon2_gate_layout() -> [u64; 10] {
    use plonky2::gates::poseidon2::Poseidon2Gate;       // FORK: module path
    type G = Poseidon2Gate<plonky2::field::goldilocks_field::GoldilocksField, 2>;
    let first_round_recorded = 0u64;                    // upstream PoseidonGate records S-box inputs from round 1 on
    [G::wire_input(0) as u64, G::wire_output(0) as u64, G::WIRE_SWAP as u64, G::wire_delta(0) as u64,
     G::wire_full_sbox_0(1, 0) as u64, G::wire_partial_sbox(0) as u64, G::wire_full_sbox_1(0, 0) as u64,
     first_round_recorded, 0, G::end() as u64]
}

/// Index of a target in plonky2's flat target space (`Target::index`): wires first, virtual targets after the trace.
fn target_index(t: Target, num_wires: usize, degree: usize) -> usize {
    match t {
        Target::Wire(w) => w.row * num_wires + w.column,
        Target::VirtualTarget { index } => degree * num_wires + index,
    }
}

/// For every copy class (partition of `representative_map`) one ROUTED wire cell of the trace, if the class has one.
/// `representative_map[i]` is the representative of target i (ProverOnlyCircuitData, built by `CircuitBuilder::build`).
fn class_cells<F: RichField + Extendable<D>, C: GenericConfig<D, F = F>, const D: usize>(
    prover: &ProverOnlyCircuitData<F, C, D>, common: &CommonCircuitData<F, D>,
) -> std::collections::HashMap<usize, u64> {
    let (nw, nr, n) = (common.config.num_wires, common.config.num_routed_wires, common.degree());
    let mut cell_of_rep = std::collections::HashMap::new();
    for row in 0..n {
        for col in 0..nr {
            let idx = row * nw + col;
            cell_of_rep.entry(prover.representative_map[idx]).or_insert((row * nw + col) as u64);
        }
    }
    cell_of_rep
}

/// The wire cell (row * num_wires + column) that carries target `t`, or `None` for a target the builder never routed
/// into the trace (an unused virtual target).
fn cell_of_target<F: RichField + Extendable<D>, C: GenericConfig<D, F = F>, const D: usize>(
    t: Target, prover: &ProverOnlyCircuitData<F, C, D>, common: &CommonCircuitData<F, D>,
    cell_of_rep: &std::collections::HashMap<usize, u64>,
) -> Option<u64> {
    let (nw, n) = (common.config.num_wires, common.degree());
    if let Target::Wire(w) = t {
        if w.column < common.config.num_routed_wires { return Some((w.row * nw + w.column) as u64); }
        return Some((w.row * nw + w.column) as u64);      // an unrouted wire is its own cell
    }
    cell_of_rep.get(&prover.representative_map[target_index(t, nw, n)]).copied()
}

/// Serialise what `prove` needs. Little-endian u64 words, layout documented in csrc/circuit.hpp.
pub fn circuit_pack<F: RichField + Extendable<D>, C: GenericConfig<D, F = F>, const D: usize>(
    data: &CircuitData<F, C, D>,
) -> Result<Vec<u64>> {
    let (c, p, cfg) = (&data.common, &data.prover_only, &data.common.config);
    if c.num_lookup_polys != 0 { bail!("lookup arguments are not implemented by the MI355X backend") }
    let sel = &c.selectors_info;
    let num_selectors = sel.num_selectors();
    let n = c.degree();
    let mut w: Vec<u64> = vec![QPCP_MAGIC];
    w.extend([c.degree_bits(), cfg.num_wires, cfg.num_routed_wires, c.num_constants - num_selectors, num_selectors,
              cfg.num_challenges, c.quotient_degree_factor, c.num_partial_products, c.num_public_inputs,
              cfg.fri_config.rate_bits, cfg.fri_config.cap_height, cfg.fri_config.proof_of_work_bits as usize,
              cfg.fri_config.num_query_rounds,
              // FORK: `zero_knowledge` = salted Merkle leaves (upstream). If the fork's "RowBlinding" mode only randomises
              // padding rows and does not salt leaves (common/src/circuit.rs:382-402), write 0 here.
              cfg.zero_knowledge as usize,
              c.num_gate_constraints, c.gates.len(), c.fri_params.reduction_arity_bits.len()].map(|x| x as u64));
    w.extend(c.fri_params.reduction_arity_bits.iter().map(|&b| b as u64));
    for (i, g) in c.gates.iter().enumerate() {
        let (ty, p0, p1, mut p2) = gate_code(&g.0.id())?;
        if ty == 10 && p2 == 0 {   // RandomAccessGate::num_extra_constants
            let used = (2 + (1usize << p0)) * p1 as usize;
            p2 = core::cmp::min(cfg.num_constants, cfg.num_routed_wires.saturating_sub(used)) as u64;
        }
        let group = &sel.groups[sel.selector_indices[i]];
        w.extend([ty, p0, p1, sel.selector_indices[i] as u64, group.start as u64, group.end as u64, g.0.num_constraints() as u64, p2]);
    }
    w.extend(c.k_is.iter().map(|k| k.to_canonical_u64()));
    w.extend(data.verifier_only.circuit_digest.elements.iter().map(|e| e.to_canonical_u64()));
    // constants_sigmas oracle, column order = plonky2's: selectors, constants, sigmas; VALUES on the subgroup in natural order
    for poly in &p.constants_sigmas_commitment.polynomials {
        let values = PolynomialCoeffs::new(poly.coeffs.clone()).fft().values;
        if values.len() != n { bail!("constants/sigmas polynomial of unexpected length") }
        w.extend(values.iter().map(|v| v.to_canonical_u64()));
    }
    // public-input cells: where `prover_only.public_inputs` (the registered targets, in registration order) sit in the trace
    if c.num_public_inputs > 0 {
        let cells = class_cells(p, c);
        w.push(PUBI_MAGIC);
        w.push(c.num_public_inputs as u64);
        for &t in &p.public_inputs {
            w.push(cell_of_target(t, p, c, &cells).ok_or_else(|| anyhow!("a public-input target is not routed into the trace"))?);
        }
    }
    // the Poseidon2 gate's wire layout, when the circuit uses that gate
    if c.gates.iter().any(|g| g.0.id().starts_with("Poseidon2Gate")) {
        w.push(P2GL_MAGIC);
        w.push(10);
        w.extend(poseidon2_gate_layout());
        let _ = P2_NO_SWAP;      // (a fork gate without swap wires writes P2_NO_SWAP as word 2)
    }
    // (the hint trailer — generators that are not attached to a gate — is only needed for stage s1 on the device; without
    // it qpgpu_prove takes the full witness plonky2's own generate_partial_witness produced. See INTEGRATION.md section 2d.)
    Ok(w)
}

/// Logical-target map of the leaf circuit for `qpgpu_leaf_map_targets` (include/qpgpu_leaf.h): the cell of every target
/// `fill_witness` sets, in the QPGPU_LT_* order. `targets` is `wormhole_circuit::circuit::CircuitTargets`.
/// A target that reached no wire maps to `u64::MAX` (qpgpu_leaf_map_targets skips it).
pub fn leaf_target_map<F: RichField + Extendable<D>, C: GenericConfig<D, F = F>, const D: usize>(
    data: &CircuitData<F, C, D>, flat_targets: &[Target],      // CircuitTargets flattened in QPGPU_LT_* order by the caller
) -> Vec<u64> {
    let cells = class_cells(&data.prover_only, &data.common);
    flat_targets.iter().map(|&t| cell_of_target(t, &data.prover_only, &data.common, &cells).unwrap_or(u64::MAX)).collect()
}

/// Target map of a recursive wrapper (private / public batch) for `qpgpu_batch_fill_proof_targets` (include/qpgpu_batch.h):
/// the cell of every logical target, slot-major, in the header's documented order — per proof target the public inputs, the
/// three caps, the openings of the zeta batch and of the zeta-next batch, pow_witness, the final polynomial, the commit-phase
/// caps and the query rounds — followed by the N x 4 dummy-nullifier preimage targets. `proof_targets` are the wrapper's
/// `ProofWithPublicInputsTarget`s (`PrivateBatchCircuitTargets::leaf_proofs`, private_batch/circuit/circuit_logic.rs),
/// `preimages` its `dummy_nullifier_pre_images`.
pub fn proof_target_map<F: RichField + Extendable<D>, C: GenericConfig<D, F = F>, const D: usize>(
    data: &CircuitData<F, C, D>, proof_targets: &[plonky2::plonk::proof::ProofWithPublicInputsTarget<D>], preimages: &[[Target; 4]],
) -> Vec<u64> {
    let cells = class_cells(&data.prover_only, &data.common);
    let mut flat: Vec<Target> = Vec::new();
    for pt in proof_targets {
        flat.extend(pt.public_inputs.iter().copied());
        let p = &pt.proof;
        for cap in [&p.wires_cap, &p.plonk_zs_partial_products_cap, &p.quotient_polys_cap] {
            for h in &cap.0 { flat.extend(h.elements); }
        }
        let o = &p.openings;
        for batch in [&o.constants, &o.plonk_sigmas, &o.wires, &o.plonk_zs, &o.partial_products, &o.quotient_polys, &o.plonk_zs_next] {
            for e in batch.iter() { flat.extend(e.to_target_array()); }
        }
        let f = &p.opening_proof;
        flat.push(f.pow_witness);
        for e in &f.final_poly.0 { flat.extend(e.to_target_array()); }
        for cap in &f.commit_phase_merkle_caps { for h in &cap.0 { flat.extend(h.elements); } }
        for q in &f.query_round_proofs {
            for (evals, path) in &q.initial_trees_proof.evals_proofs {
                flat.extend(evals.iter().copied());
                for h in &path.siblings { flat.extend(h.elements); }
            }
            for st in &q.steps {
                for e in &st.evals { flat.extend(e.to_target_array()); }
                for h in &st.merkle_proof.siblings { flat.extend(h.elements); }
            }
        }
    }
    for pre in preimages { flat.extend(pre.iter().copied()); }
    flat.iter().map(|&t| cell_of_target(t, &data.prover_only, &data.common, &cells).unwrap_or(u64::MAX)).collect()
}

// ---------------------------------------------------------------------------------------------------------------------
// The prove hook
// ---------------------------------------------------------------------------------------------------------------------
pub struct GpuCircuit { ctx: *mut QpgpuCtx, circuit: *mut QpgpuCircuit }
unsafe impl Send for GpuCircuit {}     // one GpuCircuit per proving worker thread (a ctx owns one HIP stream)

impl GpuCircuit {
    /// Once per circuit, right after `builder.build()` / `build_prover()`.
    pub fn from_circuit_data<F: RichField + Extendable<D>, C: GenericConfig<D, F = F>, const D: usize>(
        device: i32, data: &CircuitData<F, C, D>, max_batch: u32,
    ) -> Result<Self> {
        let pack = circuit_pack(data)?;
        let mut err = [0 as c_char; 200];
        if unsafe { qpgpu_pack_validate(pack.as_ptr(), pack.len(), err.as_mut_ptr()) } != 0 {
            bail!("circuit pack rejected: {}", unsafe { std::ffi::CStr::from_ptr(err.as_ptr()) }.to_string_lossy());
        }
        let mut ctx = core::ptr::null_mut();
        if unsafe { qpgpu_ctx_create(device, &mut ctx) } != QPGPU_OK { bail!("no MI355X device {device}") }
        // FORK: if `PoseidonGoldilocksConfig::Hasher` is Poseidon2 in the fork, select it here BEFORE loading:
        //   qpgpu_ctx_set_hasher(ctx, 1, qp_poseidon_core constants as 146 words: ext rc 8x12, int rc 22, diag 12, M4 16)
        let mut circuit = core::ptr::null_mut();
        let rc = unsafe { qpgpu_circuit_load_batch(ctx, pack.as_ptr(), pack.len(), max_batch, &mut circuit) };
        if rc != QPGPU_OK {
            let msg = last_error(ctx);
            unsafe { qpgpu_ctx_destroy(ctx) };
            bail!("qpgpu_circuit_load: {msg}");
        }
        Ok(Self { ctx, circuit })
    }

    /// `ProverCircuitData::prove(pw)`: witness generation stays plonky2's (it owns the generators), everything after it
    /// runs on the GPU. The returned proof is parsed from the same bytes the CPU prover would serialise.
    pub fn prove<F: RichField + Extendable<D>, C: GenericConfig<D, F = F>, const D: usize>(
        &self, data: &CircuitData<F, C, D>, pw: PartialWitness<F>,
    ) -> Result<ProofWithPublicInputs<F, C, D>> {
        // generate_partial_witness panics on "set twice with different values" exactly as before (the reference's tests
        // rely on it: wormhole/tests/src/circuit/block_header_tests.rs:34-95)
        let partition: PartitionWitness<F> =
            plonky2::iop::generator::generate_partial_witness(pw, &data.prover_only, &data.common)?;
        let public_inputs: Vec<u64> = partition.get_targets(&data.prover_only.public_inputs).iter().map(|f| f.to_canonical_u64()).collect();
        let full = partition.full_witness();                     // MatrixWitness { wire_values: Vec<Vec<F>> }, [wire][row]
        let n = data.common.degree();
        let mut wires: Vec<u64> = Vec::with_capacity(full.wire_values.len() * n);
        for col in &full.wire_values { wires.extend(col.iter().map(|f| f.to_canonical_u64())); }
        let mut bytes = vec![0u8; unsafe { qpgpu_proof_size(self.circuit) }];
        let mut len = 0usize;
        let rc = unsafe { qpgpu_prove(self.circuit, wires.as_ptr(), public_inputs.as_ptr(), bytes.as_mut_ptr(), bytes.len(), &mut len) };
        // the witness holds the spend secret: scrub our host copy (the library scrubs its device copies)
        for w in wires.iter_mut() { unsafe { core::ptr::write_volatile(w, 0) } }
        if rc != QPGPU_OK { bail!("{}", last_error(self.ctx)) }      // callers wrap: "Failed to prove: {e}" (prover/src/lib.rs:174)
        bytes.truncate(len);
        ProofWithPublicInputs::from_bytes(bytes, &data.common)
    }

    /// Many witnesses of this circuit at once (the aggregator's leaves): `GpuCircuit::from_circuit_data(.., max_batch)` sized
    /// the workspace; every stage is launched once for the batch. Proof i is byte-identical to `prove` of witness i alone.
    pub fn prove_batch<F: RichField + Extendable<D>, C: GenericConfig<D, F = F>, const D: usize>(
        &self, data: &CircuitData<F, C, D>, witnesses: Vec<PartialWitness<F>>,
    ) -> Result<Vec<ProofWithPublicInputs<F, C, D>>> {
        let n = data.common.degree();
        let size = unsafe { qpgpu_proof_size(self.circuit) };
        let (mut d_wires, mut pis_all): (Vec<*mut c_void>, Vec<Vec<u64>>) = (Vec::new(), Vec::new());
        for pw in witnesses {
            let partition = plonky2::iop::generator::generate_partial_witness(pw, &data.prover_only, &data.common)?;
            pis_all.push(partition.get_targets(&data.prover_only.public_inputs).iter().map(|f| f.to_canonical_u64()).collect());
            let full = partition.full_witness();
            let mut wires: Vec<u64> = Vec::with_capacity(full.wire_values.len() * n);
            for col in &full.wire_values { wires.extend(col.iter().map(|f| f.to_canonical_u64())); }
            let mut d: *mut c_void = core::ptr::null_mut();
            if unsafe { qpgpu_malloc(self.ctx, wires.len() * 8, &mut d) } != QPGPU_OK { bail!("{}", last_error(self.ctx)) }
            if unsafe { qpgpu_memcpy_h2d(self.ctx, d, wires.as_ptr() as *const c_void, wires.len() * 8) } != QPGPU_OK { bail!("{}", last_error(self.ctx)) }
            for w in wires.iter_mut() { unsafe { core::ptr::write_volatile(w, 0) } }
            d_wires.push(d);
        }
        let mut bufs: Vec<Vec<u8>> = (0..d_wires.len()).map(|_| vec![0u8; size]).collect();
        let outs: Vec<*mut u8> = bufs.iter_mut().map(|b| b.as_mut_ptr()).collect();
        let wire_ptrs: Vec<*const u64> = d_wires.iter().map(|d| *d as *const u64).collect();
        let pi_ptrs: Vec<*const u64> = pis_all.iter().map(|p| p.as_ptr()).collect();
        let mut lens = vec![0usize; d_wires.len()];
        let rc = unsafe { qpgpu_prove_batch_dev(self.circuit, wire_ptrs.as_ptr(), d_wires.len() as u32, pi_ptrs.as_ptr(), outs.as_ptr(), size, lens.as_mut_ptr()) };
        unsafe { qpgpu_circuit_scrub(self.circuit); }                 // device copies of everything derived from the witnesses
        let bytes = data.common.config.num_wires * n * 8;
        for d in d_wires { unsafe { qpgpu_free_scrubbed(self.ctx, d, bytes); } }   // witness copies zeroed before release
        if rc != QPGPU_OK { bail!("{}", last_error(self.ctx)) }
        bufs.into_iter().zip(lens).map(|(mut b, l)| { b.truncate(l); ProofWithPublicInputs::from_bytes(b, &data.common) }).collect()
    }
}

impl Drop for GpuCircuit {
    fn drop(&mut self) { unsafe { qpgpu_circuit_free(self.circuit); qpgpu_ctx_destroy(self.ctx); } }
}
