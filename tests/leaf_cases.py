"""CircuitInputs the reference's own tests and benches use, as LeafInputs (shared by the CPU and the GPU leaf-circuit tests).
Values come from tests/golden/poseidon2_kats.json (transcribed from the reference, sources listed there) and from the library's
KAT-pinned hash helpers (include/qpgpu_leaf.h) where the reference derives them with `hash_no_pad` at run time."""
import ctypes
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KATS = json.load(open(os.path.join(ROOT, "tests", "golden", "poseidon2_kats.json")))
P = 0xFFFFFFFF00000001
# wormhole/tests/test-helpers/src/lib.rs:18-31
DEFAULT_SECRETS = [k["secret"] for k in KATS["address_kats"][:2]]
DEFAULT_TRANSFER_COUNTS = [4, 98]
DEFAULT_INPUT_AMOUNTS = [100, 300]
DEFAULT_VOLUME_FEE_BPS = 10
DEFAULT_EXIT_ACCOUNT = bytes([4] * 32)


def header_digest():
    return bytes.fromhex(KATS["digest_hex_head"]) + bytes(KATS["digest_zero_run"]) + bytes.fromhex(KATS["digest_hex_tail"])


def header_kat(i):
    """(parent_hash, block_number, state_root, extrinsics_root, zk_tree_root, digest, expected block hash) of block-header KAT i
    (DEFAULT_BLOCK_HASHES[i], wormhole/tests/test-helpers/src/lib.rs:210-273)."""
    k = KATS["block_header_kats"][i]
    parent = bytes.fromhex(k["parent_hash"]) if "parent_hash" in k else bytes(k["parent_hash_bytes"])
    return (parent, k["block_number"], bytes.fromhex(k["state_root"]), bytes.fromhex(k["extrinsics_root"]), bytes.fromhex(k["zk_tree_root"]),
            header_digest(), bytes(k["expected_hash_bytes"]))


def dummy_inputs(L):
    """build_dummy_circuit_inputs (wormhole/aggregator/src/dummy_proof.rs:58-84,125-170): the reference bench's input
    (wormhole/prover/benches/prover.rs:31-42)."""
    d = KATS["dummy_leaf_inputs"]
    x = L.LeafInputs()
    x.asset_id, x.volume_fee_bps, x.block_number = d["asset_id"], d["volume_fee_bps"], d["block_number"]
    x.output_amount_1, x.output_amount_2 = d["output_amounts"]
    x.transfer_count, x.input_amount = d["transfer_count"], d["input_amount"]
    for name in ("nullifier", "block_hash", "secret", "parent_hash", "state_root", "extrinsics_root", "zk_tree_root"):
        x.set32(name, bytes.fromhex(d[name]))
    x.set32("exit_account_1", bytes.fromhex(d["exit_accounts"][0])); x.set32("exit_account_2", bytes.fromhex(d["exit_accounts"][1]))
    assert KATS["address_kats"][0]["secret"] == d["secret"]
    x.set32("unspendable_account", bytes.fromhex(KATS["address_kats"][0]["address"]))     # UnspendableAccount::from_secret: the first address KAT
    ctypes.memmove(x.digest, header_digest(), 110)
    x.zk_merkle_depth = d["zk_merkle_depth"]
    return x


def test_inputs(L, i):
    """CircuitInputs::test_inputs_0 / _1 (wormhole/tests/test-helpers/src/lib.rs:83-195): a dummy-mode proof (zero block hash and
    outputs) whose unspendable-account and depth-0 Merkle constraints are still enforced."""
    secret = bytes.fromhex(DEFAULT_SECRETS[i])
    x = L.LeafInputs()
    x.asset_id, x.output_amount_1, x.output_amount_2, x.volume_fee_bps = 0, 0, 0, DEFAULT_VOLUME_FEE_BPS
    x.set32("nullifier", L.nullifier(secret, DEFAULT_TRANSFER_COUNTS[i]))
    x.set32("exit_account_1", DEFAULT_EXIT_ACCOUNT).set32("exit_account_2", bytes(32)).set32("block_hash", bytes(32))
    hk = header_kat(i)
    x.block_number = hk[1]
    unsp = L.unspendable_account(secret)
    assert unsp.hex() == KATS["address_kats"][i]["address"]
    x.set32("secret", secret).set32("unspendable_account", unsp)
    x.transfer_count, x.input_amount = DEFAULT_TRANSFER_COUNTS[i], DEFAULT_INPUT_AMOUNTS[i]
    x.set32("parent_hash", bytes(32)).set32("state_root", hk[2]).set32("extrinsics_root", hk[3])
    ctypes.memmove(x.digest, hk[5], 110)
    x.set32("zk_tree_root", L.zk_leaf_hash(unsp, x.transfer_count, 0, x.input_amount))
    x.zk_merkle_depth = 0
    return x


def real_inputs(L, depth=3, seed=5, secret_index=1):
    """A spend that is NOT a dummy: real outputs under the fee rule, a Merkle path of `depth` levels built with the chain's tree
    rules (ZkMerkleProof::from_unsorted), a header that commits to the path's root, and the nullifier / block hash the circuit
    recomputes. The reference builds such inputs in its aggregator tests from chain data; here they come from the library's
    KAT-pinned helpers."""
    rng = np.random.default_rng(seed)
    secret = bytes.fromhex(DEFAULT_SECRETS[secret_index])
    x = L.LeafInputs()
    x.asset_id, x.volume_fee_bps = 0, DEFAULT_VOLUME_FEE_BPS
    x.transfer_count, x.input_amount = DEFAULT_TRANSFER_COUNTS[secret_index], DEFAULT_INPUT_AMOUNTS[secret_index]
    x.output_amount_1, x.output_amount_2 = (200, 97) if secret_index == 1 else (66, 33)    # (200 + 97) * 10000 <= 300 * 9990; (66 + 33) * 10000 <= 100 * 9990
    unsp = L.unspendable_account(secret)
    x.set32("secret", secret).set32("unspendable_account", unsp)
    x.set32("nullifier", L.nullifier(secret, x.transfer_count))
    x.set32("exit_account_1", DEFAULT_EXIT_ACCOUNT).set32("exit_account_2", bytes([7] * 32))
    leaf = L.zk_leaf_hash(unsp, x.transfer_count, x.asset_id, x.input_amount)
    sibs = []
    for _ in range(depth):
        lvl = rng.integers(0, 256, (3, 32), dtype=np.uint8); lvl[:, 7::8] &= 0x7F        # canonical limbs
        sibs.append([s.tobytes() for s in lvl])
    sorted_sibs, positions, root = L.zk_proof_from_unsorted(leaf, sibs)
    x.zk_merkle_depth = depth
    ctypes.memmove(x.zk_merkle_siblings, sorted_sibs, len(sorted_sibs))
    for l, p in enumerate(positions):
        x.zk_merkle_positions[l] = p
    x.set32("zk_tree_root", root)
    hk = header_kat(1)
    x.set32("parent_hash", hk[0]).set32("state_root", hk[2]).set32("extrinsics_root", hk[3])
    x.block_number = hk[1]
    ctypes.memmove(x.digest, hk[5], 110)
    x.set32("block_hash", L.block_hash(hk[0], hk[1], hk[2], hk[3], root, hk[5]))
    return x


def header_inputs(L, i):
    """BlockHeader::test_inputs_i for the block-header fragment circuit: the KAT header and DEFAULT_BLOCK_HASHES[i] as the claimed
    block hash (every other field unused by that fragment)."""
    hk = header_kat(i)
    x = L.LeafInputs()
    x.set32("parent_hash", hk[0]).set32("state_root", hk[2]).set32("extrinsics_root", hk[3]).set32("zk_tree_root", hk[4]).set32("block_hash", hk[6])
    x.block_number = hk[1]
    ctypes.memmove(x.digest, hk[5], 110)
    return x


def digest_felts(b32):
    return [int.from_bytes(bytes(b32)[8 * i:8 * i + 8], "little") % P for i in range(4)]


def proof_public_inputs(proof, n):
    return np.frombuffer(proof[-8 * n:], dtype=np.uint64) if n else np.zeros(0, dtype=np.uint64)


def shared_tree_inputs(L, count, depth=2, seed=11, exits=None, outputs=None, asset_id=0):
    """`count` real spends of ONE block: different random secrets, their leaves in one 4-ary tree of `depth` levels (the other
    leaves random), one header committing to that tree's root — what a private batch aggregates (every real slot carries the
    same block hash, distinct nullifiers). exits: per spend (exit_account_1, exit_account_2) as 32 bytes each; outputs: per spend
    (output_amount_1, output_amount_2) under the fee rule for an input of 300. asset_id: one value or one per spend (the asset is part of
    the leaf hash, so leaves of different assets can share a tree)."""
    rng = np.random.default_rng(seed)
    assert 1 <= count <= 4 ** depth

    def canon32():
        b = rng.integers(0, 256, 32, dtype=np.uint8); b[7::8] &= 0x7F
        return b.tobytes()

    assets = [asset_id] * count if isinstance(asset_id, int) else list(asset_id)
    spends = []
    for i in range(count):
        secret = canon32()
        tc = int(rng.integers(1, 1000))
        unsp = L.unspendable_account(secret)
        spends.append((secret, tc, unsp, L.zk_leaf_hash(unsp, tc, assets[i], 300)))
    level = [s[3] for s in spends] + [canon32() for _ in range(4 ** depth - count)]
    levels = [level]
    for _ in range(depth):
        level = [L.zk_proof_from_unsorted(level[g], [level[g + 1:g + 4]])[2] for g in range(0, len(level), 4)]
        levels.append(level)
    root = levels[-1][0]
    hk = header_kat(1)
    bh = L.block_hash(hk[0], hk[1], hk[2], hk[3], root, hk[5])
    out = []
    for i, (secret, tc, unsp, leaf) in enumerate(spends):
        x = L.LeafInputs()
        x.asset_id, x.volume_fee_bps, x.transfer_count, x.input_amount = assets[i], DEFAULT_VOLUME_FEE_BPS, tc, 300
        x.output_amount_1, x.output_amount_2 = outputs[i] if outputs else (200, 97)
        x.set32("secret", secret).set32("unspendable_account", unsp).set32("nullifier", L.nullifier(secret, tc))
        e1, e2 = exits[i] if exits else (bytes([4] * 32), bytes([7] * 32))
        x.set32("exit_account_1", e1).set32("exit_account_2", e2)
        sibs, idx = [], i
        for l in range(depth):
            g = idx - idx % 4
            sibs.append([levels[l][k] for k in range(g, g + 4) if k != idx])
            idx //= 4
        sorted_sibs, positions, r = L.zk_proof_from_unsorted(leaf, sibs)
        assert r == root
        x.zk_merkle_depth = depth
        ctypes.memmove(x.zk_merkle_siblings, sorted_sibs, len(sorted_sibs))
        for l, p in enumerate(positions):
            x.zk_merkle_positions[l] = p
        x.set32("zk_tree_root", root).set32("parent_hash", hk[0]).set32("state_root", hk[2]).set32("extrinsics_root", hk[3]).set32("block_hash", bh)
        x.block_number = hk[1]
        ctypes.memmove(x.digest, hk[5], 110)
        out.append(x)
    return out
