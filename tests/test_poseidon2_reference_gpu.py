"""The HIP Poseidon2 kernels on vectors the REFERENCE holds: the only numeric goldens of the hot path
are the seven Poseidon2 known-answer vectors — five `secret -> H(H("wormhole" || secret))` addresses
(wormhole/tests/src/circuit/unspendable_account_tests.rs:9-24) and two block-header hashes over a 45-element preimage
(wormhole/tests/test-helpers/src/lib.rs:210-219), transcribed into tests/golden/poseidon2_kats.json. Here the expected
values come from that file, not from the oracle and not from the product's host code:

  * the device sponge qpgpu_poseidon2_hash_pad10_dev (p2_pad10_sponge_kernel) must reproduce all seven;
  * the hashing kernels of a context whose proof-system hasher is Poseidon2 with qp-poseidon-core's parameters
    (permute_kernel<Poseidon2P>, leaf_hash_kernel<Poseidon2P>, node_kernel<Poseidon2P>, pow_kernel<Poseidon2P>) are tied to
    the same vectors: a permutation chain rebuilt from permute_kernel outputs reproduces the KATs, and trees / whole proofs
    agree with the oracle running orc_p2_qp_params, the parameter path the seven vectors pin
    (tests/test_oracle_poseidon.py::test_poseidon2_pinned_by_all_seven_reference_kats)."""
import ctypes
import json
import os

import numpy as np
import pytest

from oracle_binding import OracleCircuit

P = 0xFFFFFFFF00000001
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KATS = json.load(open(os.path.join(ROOT, "tests", "golden", "poseidon2_kats.json")))


def digest_felts(b):
    """32 bytes -> 4 field elements, 8 bytes each little-endian (common/src/serialization.rs:228-247)."""
    return [int.from_bytes(b[8 * i:8 * i + 8], "little") % P for i in range(4)]


def bytes_to_felts(b):
    """bytes_to_u64s: 4 bytes per element little-endian after appending 0x01 and zero padding (serialization.rs:127-141)."""
    b = bytes(b) + b"\x01"
    b += bytes((-len(b)) % 4)
    return [int.from_bytes(b[4 * i:4 * i + 4], "little") for i in range(len(b) // 4)]


def felts_digest_bytes(f):
    return b"".join(int(x).to_bytes(8, "little") for x in f)


def address_preimage(secret_hex):
    return bytes_to_felts(KATS["salt"].encode()) + digest_felts(bytes.fromhex(secret_hex))


def header_preimage(k):
    digest = bytes.fromhex(KATS["digest_hex_head"]) + bytes(KATS["digest_zero_run"]) + bytes.fromhex(KATS["digest_hex_tail"])
    parent = bytes.fromhex(k["parent_hash"]) if "parent_hash" in k else bytes(k["parent_hash_bytes"])
    pre = digest_felts(parent) + [k["block_number"]] + digest_felts(bytes.fromhex(k["state_root"])) + \
        digest_felts(bytes.fromhex(k["extrinsics_root"])) + digest_felts(bytes.fromhex(k["zk_tree_root"])) + bytes_to_felts(digest)
    assert len(pre) == 45
    return pre


@pytest.fixture()
def gpu_qp(pkg):
    """A context whose PROOF-SYSTEM hasher is Poseidon2 with qp-poseidon-core's parameters."""
    g = pkg.QpGpu(0, hasher=pkg.poseidon2_qp_params())
    yield g
    g.close()


@pytest.fixture()
def orc_qp(pkg, orc):
    orc.select_poseidon2(*pkg.poseidon2_qp_params())
    yield orc
    orc.select_poseidon()


@pytest.mark.gpu
def test_device_sponge_reproduces_all_seven_reference_kats(pkg, gpu):
    # five addresses: H(H(felts("wormhole") || secret limbs)), both hashes on the device
    pre = np.array([address_preimage(k["secret"]) for k in KATS["address_kats"]], dtype=np.uint64)
    assert pre.shape == (5, 7)
    inner = gpu.poseidon2_hash_pad10(pre)
    outer = gpu.poseidon2_hash_pad10(inner)
    for k, h in zip(KATS["address_kats"], outer):
        assert felts_digest_bytes(h).hex() == k["address"]
    # two block headers: one hash over 45 elements = six rate blocks (what tells additive from overwriting absorption)
    hp = np.array([header_preimage(k) for k in KATS["block_header_kats"]], dtype=np.uint64)
    got = gpu.poseidon2_hash_pad10(hp)
    for k, h in zip(KATS["block_header_kats"], got):
        assert felts_digest_bytes(h) == bytes(k["expected_hash_bytes"])
    assert len(KATS["address_kats"]) == 5 and len(KATS["block_header_kats"]) == 2


@pytest.mark.gpu
def test_device_sponge_edge_lengths_match_the_host_sponge(pkg, gpu):
    """Lengths around the block boundary (0, 7, 8, 9, 15, 16, 17 ...): a full last block gets a block of padding of its own."""
    lib = pkg.load_library()
    rng = np.random.default_rng(31)
    for ln in (0, 1, 7, 8, 9, 15, 16, 17, 45, 64):
        pre = rng.integers(0, P, (3, ln), dtype=np.uint64)
        got = gpu.poseidon2_hash_pad10(pre)
        for i in range(3):
            want = np.empty(4, dtype=np.uint64)
            row = np.ascontiguousarray(pre[i])
            assert lib.qpgpu_poseidon2_hash_pad10(None, 0, row.ctypes.data if ln else None, ln, want.ctypes.data) == 0
            assert (got[i] == want).all(), ln
    # a caller-supplied parameter block takes the other route (uploaded for the call)
    blk = np.concatenate([a.ravel() for a in pkg.poseidon2_qp_params()])
    pre = rng.integers(0, P, (2, 11), dtype=np.uint64)
    assert (gpu.poseidon2_hash_pad10(pre, params=blk) == gpu.poseidon2_hash_pad10(pre)).all()


@pytest.mark.gpu
def test_permute_kernel_chain_reproduces_reference_kats(pkg, gpu_qp):
    """permute_kernel<Poseidon2P> with the context's parameter block: the sponge rebuilt on the host from device permutations
    (additive absorption of the padded blocks) gives the reference's block-header hash and first address."""
    def sponge(pre):
        pre = list(pre) + [1]
        pre += [0] * ((-len(pre)) % 8)
        st = np.zeros(12, dtype=np.uint64)
        for i in range(0, len(pre), 8):
            for j in range(8):
                st[j] = (int(st[j]) + int(pre[i + j])) % P
            st = gpu_qp.poseidon_permute(st.reshape(1, 12))[0]
        return [int(x) for x in st[:4]]
    k = KATS["block_header_kats"][1]
    assert felts_digest_bytes(sponge(header_preimage(k))) == bytes(k["expected_hash_bytes"])
    a = KATS["address_kats"][0]
    assert felts_digest_bytes(sponge(sponge(address_preimage(a["secret"])))).hex() == a["address"]


@pytest.mark.gpu
def test_matrix_build_under_the_pinned_parameters(pkg, gpu_qp, orc_qp):
    """Launches of 2^18 hashes or more under Poseidon2 with qp-poseidon-core's parameters run the matrix-pipe build
    (merkle_kernels_mx.hip: the 22 internal rounds, the external layer before them and the constants behind them as one int8 GEMM,
    pmf::permute_p2qp). A 2^18-leaf tree over 21 columns (three permutations per leaf, the last block ragged) with runs of extreme
    elements: digests, every level and the cap equal the oracle's tree over orc_p2_qp_params' permutation, the KAT-pinned path."""
    log_leaves, width, cap_h = 18, 21, 4
    n = 1 << log_leaves
    rng = np.random.default_rng(4711)
    leaves = rng.integers(0, P, (n, width), dtype=np.uint64)
    ext = np.array([0, P - 1, 2**32 - 1, 2**63, 2**32, P - 2**32], dtype=np.uint64)
    for j in rng.choice(n, 4000, replace=False):
        k = int(rng.integers(1, width))
        leaves[j, rng.choice(width, k, replace=False)] = rng.choice(ext, k)
    dig_want, cap_want = orc_qp.merkle(leaves, cap_h)
    d_dig = gpu_qp.alloc(gpu_qp.merkle_digest_count(log_leaves, cap_h) * 32)
    d_cols = gpu_qp.to_device(np.ascontiguousarray(leaves.T))
    cap = gpu_qp.merkle_build_dev(d_cols, n, width, log_leaves, cap_h, d_dig)
    assert np.array_equal(cap, cap_want)
    assert np.array_equal(d_dig.download().reshape(-1, 4), dig_want)
    d_cols.free(); d_dig.free()


@pytest.mark.gpu
def test_tree_kernels_under_the_pinned_parameters(pkg, gpu_qp, orc_qp):
    """leaf_hash_kernel<Poseidon2P>, leaf_hash_rows_kernel<Poseidon2P> and node_kernel<Poseidon2P> against the oracle's Merkle
    tree over orc_p2_qp_params' permutation (the KAT-pinned path), through the polynomial-batch and the row-major entries."""
    rng = np.random.default_rng(17)
    st = rng.integers(0, P, (64, 12), dtype=np.uint64)
    got = gpu_qp.poseidon_permute(st)
    assert all((got[i] == orc_qp.poseidon(st[i])).all() for i in range(64))
    for ncols, log_n, cap_h in ((21, 9, 3), (135, 7, 4), (3, 6, 0)):
        vals = rng.integers(0, P, (ncols, 1 << log_n), dtype=np.uint64)
        o = pkg.PolyOracle(gpu_qp, vals, rate_bits=3, cap_height=cap_h)
        _, cap = orc_qp.merkle(np.ascontiguousarray(o.read(lde=True).T), cap_h)
        assert (o.cap() == cap).all(), (ncols, log_n)
        o.close()
    rows = rng.integers(0, P, (1 << 8, 32), dtype=np.uint64)
    d_rows = gpu_qp.to_device(rows)
    d_dig = gpu_qp.alloc(gpu_qp.merkle_digest_count(8, 4) * 32)
    cap = gpu_qp.merkle_build_rows_dev(d_rows, 32, 8, 4, d_dig)
    dig, want = orc_qp.merkle(rows, 4)
    assert (cap == want).all()
    assert (d_dig.download().reshape(-1, 4) == dig).all()
    d_rows.free(); d_dig.free()


@pytest.mark.gpu
def test_whole_proofs_under_poseidon2_as_proof_hasher_with_the_pinned_parameters(pkg, gpu_qp, orc_qp):
    """Everything the proof-system hasher touches (trees, transcript, public-input hash, pow_kernel<Poseidon2P>) with
    qp-poseidon-core's parameters: proof bytes equal the oracle's, single and as a lockstep batch, plain and zero-knowledge."""
    pkg.set_hasher_poseidon2(*pkg.poseidon2_qp_params())       # synth hashes the public inputs with the process default
    try:
        for d, kw, zk in ((8, dict(seed=181, num_wires=24, num_routed=16, num_public_inputs=3), False),
                          (9, dict(seed=182, poseidon=True, base_sum=True, ext_arith=True, recursion=True), True)):
            pack, wires, pis = pkg.synth_circuit(d, **kw)
            if zk:
                pack[14] = 1
            circ = pkg.Circuit(gpu_qp, pack, max_batch=3); oc = OracleCircuit(orc_qp, pack)
            ver = pkg.Verifier(pack, circuit=circ, hasher=1)
            try:
                circ.set_blinding_seed(9)
                got = circ.prove(wires, pis)
                assert got == oc.prove(wires, pis, seed=9)
                assert oc.verify(got) == 0 and ver.verify(got)
                circ.set_blinding_seed(9)
                d_w = gpu_qp.to_device(np.stack([wires] * 3))
                batch = circ.prove_batch_dev([d_w.ptr + i * wires.nbytes for i in range(3)], [pis] * 3)
                d_w.free(scrub=True)
                assert batch[0] == got and all(oc.verify(b) == 0 for b in batch)
            finally:
                ver.close(); circ.close(); oc.close()
    finally:
        pkg.set_hasher_poseidon()
