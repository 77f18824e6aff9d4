// leaf_circuit.cpp — the Wormhole leaf circuit, restated on the native builder (builder.hpp): SURVEY.md §8 rows a6 / a2.
//
//   CircuitTargets::new                         wormhole/circuit/src/circuit.rs:44-55 (creation order = public-input order)
//     ZkMerkleProofTargets::new / ZkLeafTargets wormhole/circuit/src/zk_merkle_proof.rs:78-99,147-176
//     NullifierTargets::new                     wormhole/circuit/src/nullifier.rs:272-281
//     UnspendableAccountTargets::new            wormhole/circuit/src/unspendable_account.rs:200-207
//     DualExitAccountTargets::new               wormhole/circuit/src/substrate_account.rs:111-136
//     BlockHeaderTargets / HeaderTargets::new   wormhole/circuit/src/block_header/mod.rs:50-56, header.rs:36-66
//   WormholeCircuit::new_internal               wormhole/circuit/src/circuit.rs:115-152
//     UnspendableAccount::circuit               unspendable_account.rs:215-237
//     ZkMerkleProofData::circuit                zk_merkle_proof.rs:480-626
//     BlockHeader::circuit_without_hash_binding block_header/mod.rs:79-85
//     connect_shared_targets                    circuit.rs:233-323
//       Nullifier::conditional_hash_binding     nullifier.rs:285-325
//       BlockHeader::conditional_block_hash_binding  block_header/mod.rs:93-108
//   gadgets is_const_less_than / enforce_target_less_than_const   common/src/gadgets.rs:40-125
//
// Every statement of those functions appears below in the same order, on the builder's restatement of plonky2's gadgets;
// the logical targets of include/qpgpu_leaf.h are the fields of CircuitTargets. The result is a circuit pack plus the
// wire cell of every logical target (what qpgpu_leaf_map_targets takes), so that
//   CircuitInputs -> qpgpu_leaf_fill_witness -> cells -> stage s1 on the device -> s2..s12
// proves the reference's statement: the proof's 21 public inputs are the reference's, its block-hash, nullifier and
// unspendable-account digests are computed by Poseidon2 gate rows, its Merkle walk by selects over the position hints.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include "../../include/qpgpu.h"
#include "../../include/qpgpu_leaf.h"
#include "builder.hpp"
#include "gadgets.hpp"
#include "poseidon.hpp"

using cb::BoolTarget;
using cb::Builder;
using cb::HashOutTarget;
using cb::Target;
using gl::u64;

namespace {

constexpr unsigned MAX_DEPTH = QPGPU_LEAF_MAX_DEPTH;

struct ZkLeafTargets { Target to_account[4], transfer_count[2], asset_id, input_amount, output_amount_1, output_amount_2, volume_fee_bps; };
struct ZkMerkleProofTargets {
    HashOutTarget root_hash; Target depth;
    HashOutTarget siblings[MAX_DEPTH][3]; Target positions[MAX_DEPTH];
    ZkLeafTargets leaf; BoolTarget is_not_dummy;
};
struct NullifierTargets { HashOutTarget hash, secret; Target transfer_count[2]; };
struct UnspendableAccountTargets { HashOutTarget account_id, secret; };
struct DualExitAccountTargets { Target exit_account_1[4], exit_account_2[4]; };
struct HeaderTargets { Target parent_hash[4], block_number, state_root[4], extrinsics_root[4], zk_tree_root[4], digest[QPGPU_LEAF_DIGEST_LOGS_FELTS]; };
struct BlockHeaderTargets { HashOutTarget block_hash; HeaderTargets header; };
struct CircuitTargets {
    NullifierTargets nullifier; UnspendableAccountTargets unspendable_account; ZkMerkleProofTargets zk_merkle_proof;
    DualExitAccountTargets exit_accounts; BlockHeaderTargets block_header;
};

// string_to_felts (common/src/serialization.rs): 4 bytes per element little-endian after the terminator 0x01
std::vector<Target> salt_constants(Builder &b, const char *salt) {
    u64 f[8];
    const size_t n = qpgpu_bytes_to_felts((const uint8_t *)salt, std::strlen(salt), f, 8);
    std::vector<Target> out;
    for (size_t i = 0; i < n; i++) out.push_back(b.constant(f[i]));
    return out;
}

// ---- target creation, in CircuitTargets::new's order ----
ZkMerkleProofTargets zk_merkle_proof_targets(Builder &b) {
    ZkMerkleProofTargets t;
    // ZkLeafTargets::new: the four public inputs first ("registered first for consistent ordering")
    t.leaf.asset_id = b.add_virtual_public_input();
    t.leaf.output_amount_1 = b.add_virtual_public_input();
    t.leaf.output_amount_2 = b.add_virtual_public_input();
    t.leaf.volume_fee_bps = b.add_virtual_public_input();
    for (auto &e : t.leaf.to_account) e = b.add_virtual_target();
    for (auto &e : t.leaf.transfer_count) e = b.add_virtual_target();
    t.leaf.input_amount = b.add_virtual_target();
    t.root_hash = b.add_virtual_hash();
    t.depth = b.add_virtual_target();
    t.is_not_dummy = b.add_virtual_bool_target_safe();
    for (unsigned l = 0; l < MAX_DEPTH; l++) for (auto &s : t.siblings[l]) s = b.add_virtual_hash();
    for (auto &p : t.positions) p = b.add_virtual_target();
    return t;
}
CircuitTargets circuit_targets(Builder &b) {
    CircuitTargets t;
    t.zk_merkle_proof = zk_merkle_proof_targets(b);       // first, so that asset_id is public input 0
    t.nullifier.hash = b.add_virtual_hash_public_input();
    t.nullifier.secret = b.add_virtual_hash();
    for (auto &e : t.nullifier.transfer_count) e = b.add_virtual_target();
    t.unspendable_account.account_id = b.add_virtual_hash();
    t.unspendable_account.secret = b.add_virtual_hash();
    for (auto &e : t.exit_accounts.exit_account_1) e = b.add_virtual_public_input();
    for (auto &e : t.exit_accounts.exit_account_2) e = b.add_virtual_public_input();
    t.block_header.block_hash = b.add_virtual_hash_public_input();
    HeaderTargets &h = t.block_header.header;
    for (auto &e : h.parent_hash) e = b.add_virtual_target();
    h.block_number = b.add_virtual_public_input();
    for (auto &e : h.state_root) e = b.add_virtual_target();
    for (auto &e : h.extrinsics_root) e = b.add_virtual_target();
    for (auto &e : h.zk_tree_root) e = b.add_virtual_target();
    for (auto &e : h.digest) e = b.add_virtual_target();
    return t;
}

using gadgets::enforce_target_less_than_const;
using gadgets::is_const_less_than;

// ---- fragments ----
void unspendable_account_circuit(const UnspendableAccountTargets &t, Builder &b) {
    std::vector<Target> preimage = salt_constants(b, "wormhole");
    for (Target e : t.secret.elements) preimage.push_back(e);
    b.set_hash_tag(QPGPU_LEAF_HASH_UNSPENDABLE_INNER);
    const HashOutTarget inner = b.hash_n_to_hash_no_pad_p2(preimage);
    b.set_hash_tag(QPGPU_LEAF_HASH_UNSPENDABLE_OUTER);
    const HashOutTarget outer = b.hash_n_to_hash_no_pad_p2({inner.elements[0], inner.elements[1], inner.elements[2], inner.elements[3]});
    for (int i = 0; i < 4; i++) b.connect(outer.elements[i], t.account_id.elements[i]);
}

void zk_merkle_proof_circuit(const ZkMerkleProofTargets &t, Builder &b) {
    const Target zero = b.zero();
    // 32-bit range checks: transfer_count x2, asset_id, input_amount, output_amount_1, output_amount_2, volume_fee_bps
    for (Target x : {t.leaf.transfer_count[0], t.leaf.transfer_count[1], t.leaf.asset_id, t.leaf.input_amount, t.leaf.output_amount_1,
                     t.leaf.output_amount_2, t.leaf.volume_fee_bps})
        b.range_check(x, 32);
    // fee constraint: (output_1 + output_2) * 10000 <= input * (10000 - fee_bps)
    const Target ten_thousand = b.constant(10000);
    const Target total_output = b.add(t.leaf.output_amount_1, t.leaf.output_amount_2);
    const Target lhs = b.mul(total_output, ten_thousand);
    const Target fee_complement = b.sub(ten_thousand, t.leaf.volume_fee_bps);
    b.range_check(fee_complement, 14);
    const Target rhs = b.mul(t.leaf.input_amount, fee_complement);
    const Target diff = b.sub(rhs, lhs);
    b.range_check(diff, 48);
    // leaf hash: (to, transfer_count, asset_id, amount)
    std::vector<Target> leaf_felts(t.leaf.to_account, t.leaf.to_account + 4);
    leaf_felts.push_back(t.leaf.transfer_count[0]); leaf_felts.push_back(t.leaf.transfer_count[1]);
    leaf_felts.push_back(t.leaf.asset_id); leaf_felts.push_back(t.leaf.input_amount);
    b.set_hash_tag(QPGPU_LEAF_HASH_ZK_LEAF);
    const HashOutTarget leaf_hash = b.hash_n_to_hash_no_pad_p2(leaf_felts);
    // depth <= MAX_DEPTH
    unsigned n_log = 0;
    while ((MAX_DEPTH >> n_log) != 0) n_log++;            // usize::BITS - MAX_DEPTH.leading_zeros() = 5
    enforce_target_less_than_const(b, t.depth, MAX_DEPTH + 1, n_log);
    // the walk: the running hash is inserted among the sorted siblings at the hinted position
    HashOutTarget current_hash = leaf_hash;
    for (unsigned level = 0; level < MAX_DEPTH; level++) {
        const BoolTarget is_active_level = is_const_less_than(b, level, t.depth, n_log);
        const HashOutTarget *siblings = t.siblings[level];
        const Target position = t.positions[level];
        b.range_check(position, 2);
        const Target one = b.one(), two = b.constant(2), three = b.constant(3);
        const BoolTarget pos_is_0 = b.is_equal(position, zero);
        const BoolTarget pos_is_1 = b.is_equal(position, one);
        const BoolTarget pos_is_2 = b.is_equal(position, two);
        const BoolTarget pos_is_3 = b.is_equal(position, three);
        std::vector<Target> parent_preimage;
        for (unsigned slot = 0; slot < 4; slot++)
            for (unsigned e = 0; e < 4; e++) {
                Target child;
                switch (slot) {
                case 0: child = b.select(pos_is_0, current_hash.elements[e], siblings[0].elements[e]); break;
                case 1: {
                    const Target not_current = b.select(pos_is_0, siblings[0].elements[e], siblings[1].elements[e]);
                    child = b.select(pos_is_1, current_hash.elements[e], not_current);
                    break;
                }
                case 2: {
                    const BoolTarget pos_le_1 = b.or_(pos_is_0, pos_is_1);
                    const Target not_current = b.select(pos_le_1, siblings[1].elements[e], siblings[2].elements[e]);
                    child = b.select(pos_is_2, current_hash.elements[e], not_current);
                    break;
                }
                default: child = b.select(pos_is_3, current_hash.elements[e], siblings[2].elements[e]); break;
                }
                parent_preimage.push_back(child);
            }
        b.set_hash_tag(QPGPU_LEAF_HASH_MERKLE_LEVEL_0 + (int)level);
        const HashOutTarget parent_hash = b.hash_n_to_hash_no_pad_p2(parent_preimage);
        HashOutTarget next;
        for (int i = 0; i < 4; i++) next.elements[i] = b.select(is_active_level, parent_hash.elements[i], current_hash.elements[i]);
        current_hash = next;
        for (int i = 0; i < 4; i++) b.add_hint_target(next.elements[i]);      // the running hash after this level: the walk's serial path
    }
    // the computed root equals the expected one unless the proof is a dummy
    for (int i = 0; i < 4; i++) {
        const Target d = b.sub(current_hash.elements[i], t.root_hash.elements[i]);
        const Target result = b.mul(d, t.is_not_dummy.target);
        b.connect(result, zero);
    }
}

HashOutTarget computed_nullifier(const NullifierTargets &t, Builder &b) {
    std::vector<Target> preimage = salt_constants(b, "~nullif~");
    for (Target e : t.secret.elements) preimage.push_back(e);
    for (Target e : t.transfer_count) preimage.push_back(e);
    b.set_hash_tag(QPGPU_LEAF_HASH_NULLIFIER_INNER);
    const HashOutTarget inner = b.hash_n_to_hash_no_pad_p2(preimage);
    b.set_hash_tag(QPGPU_LEAF_HASH_NULLIFIER_OUTER);
    return b.hash_n_to_hash_no_pad_p2({inner.elements[0], inner.elements[1], inner.elements[2], inner.elements[3]});
}
void conditional_binding(Builder &b, const HashOutTarget &claimed, const HashOutTarget &computed, Target is_not_dummy) {
    const Target zero = b.zero();
    for (int i = 0; i < 4; i++) {
        const Target d = b.sub(claimed.elements[i], computed.elements[i]);
        const Target result = b.mul(d, is_not_dummy);
        b.connect(result, zero);
    }
}
HashOutTarget computed_block_hash(const BlockHeaderTargets &t, Builder &b) {
    const HeaderTargets &h = t.header;
    std::vector<Target> pre(h.parent_hash, h.parent_hash + 4);      // HeaderTargets::collect_to_vec
    pre.push_back(h.block_number);
    pre.insert(pre.end(), h.state_root, h.state_root + 4);
    pre.insert(pre.end(), h.extrinsics_root, h.extrinsics_root + 4);
    pre.insert(pre.end(), h.zk_tree_root, h.zk_tree_root + 4);
    pre.insert(pre.end(), h.digest, h.digest + QPGPU_LEAF_DIGEST_LOGS_FELTS);
    b.set_hash_tag(QPGPU_LEAF_HASH_BLOCK_HEADER);
    return b.hash_n_to_hash_no_pad_p2(pre);
}

void connect_shared_targets(const CircuitTargets &t, Builder &b) {
    b.connect_hashes(t.nullifier.secret, t.unspendable_account.secret);
    for (int i = 0; i < 2; i++) b.connect(t.nullifier.transfer_count[i], t.zk_merkle_proof.leaf.transfer_count[i]);
    for (int i = 0; i < 4; i++) b.connect(t.unspendable_account.account_id.elements[i], t.zk_merkle_proof.leaf.to_account[i]);
    // dummy detection: block_hash == 0 AND both outputs == 0
    const Target zero = b.zero(), one = b.one();
    const Target *bh = t.block_header.block_hash.elements;
    const BoolTarget bh0 = b.is_equal(bh[0], zero), bh1 = b.is_equal(bh[1], zero), bh2 = b.is_equal(bh[2], zero), bh3 = b.is_equal(bh[3], zero);
    const BoolTarget bh01 = b.and_(bh0, bh1), bh23 = b.and_(bh2, bh3);
    const BoolTarget block_hash_is_zero = b.and_(bh01, bh23);
    const ZkLeafTargets &leaf = t.zk_merkle_proof.leaf;
    const BoolTarget o1 = b.is_equal(leaf.output_amount_1, zero), o2 = b.is_equal(leaf.output_amount_2, zero);
    const BoolTarget both_outputs_zero = b.and_(o1, o2);
    const BoolTarget is_dummy = b.and_(block_hash_is_zero, both_outputs_zero);
    const Target is_not_dummy = b.sub(one, is_dummy.target);
    b.connect(t.zk_merkle_proof.is_not_dummy.target, is_not_dummy);
    conditional_binding(b, t.nullifier.hash, computed_nullifier(t.nullifier, b), is_not_dummy);
    conditional_binding(b, t.block_header.block_hash, computed_block_hash(t.block_header, b), is_not_dummy);
    // header.zk_tree_root == zk_merkle_proof.root_hash unless dummy
    for (int i = 0; i < 4; i++) {
        const Target d = b.sub(t.block_header.header.zk_tree_root[i], t.zk_merkle_proof.root_hash.elements[i]);
        const Target result = b.mul(d, is_not_dummy);
        b.connect(result, zero);
    }
}

void logical_targets(const CircuitTargets &t, Target (&lt)[QPGPU_LT_COUNT]) {
    auto set4 = [&](unsigned base, const Target *e) { for (int i = 0; i < 4; i++) lt[base + i] = e[i]; };
    set4(QPGPU_LT_NULLIFIER_HASH, t.nullifier.hash.elements);
    set4(QPGPU_LT_NULLIFIER_SECRET, t.nullifier.secret.elements);
    lt[QPGPU_LT_NULLIFIER_TRANSFER_COUNT] = t.nullifier.transfer_count[0]; lt[QPGPU_LT_NULLIFIER_TRANSFER_COUNT + 1] = t.nullifier.transfer_count[1];
    set4(QPGPU_LT_UNSPENDABLE_ACCOUNT_ID, t.unspendable_account.account_id.elements);
    set4(QPGPU_LT_UNSPENDABLE_SECRET, t.unspendable_account.secret.elements);
    const ZkMerkleProofTargets &z = t.zk_merkle_proof;
    set4(QPGPU_LT_ZK_ROOT_HASH, z.root_hash.elements);
    lt[QPGPU_LT_ZK_DEPTH] = z.depth;
    for (unsigned l = 0; l < MAX_DEPTH; l++) {
        for (unsigned s = 0; s < 3; s++) set4(QPGPU_LT_ZK_SIBLINGS + (l * 3 + s) * 4, z.siblings[l][s].elements);
        lt[QPGPU_LT_ZK_POSITIONS + l] = z.positions[l];
    }
    set4(QPGPU_LT_LEAF_TO_ACCOUNT, z.leaf.to_account);
    lt[QPGPU_LT_LEAF_TRANSFER_COUNT] = z.leaf.transfer_count[0]; lt[QPGPU_LT_LEAF_TRANSFER_COUNT + 1] = z.leaf.transfer_count[1];
    lt[QPGPU_LT_LEAF_ASSET_ID] = z.leaf.asset_id; lt[QPGPU_LT_LEAF_INPUT_AMOUNT] = z.leaf.input_amount;
    lt[QPGPU_LT_LEAF_OUTPUT_AMOUNT_1] = z.leaf.output_amount_1; lt[QPGPU_LT_LEAF_OUTPUT_AMOUNT_2] = z.leaf.output_amount_2;
    lt[QPGPU_LT_LEAF_VOLUME_FEE_BPS] = z.leaf.volume_fee_bps;
    set4(QPGPU_LT_EXIT_ACCOUNT_1, t.exit_accounts.exit_account_1);
    set4(QPGPU_LT_EXIT_ACCOUNT_2, t.exit_accounts.exit_account_2);
    set4(QPGPU_LT_BLOCK_HASH, t.block_header.block_hash.elements);
    const HeaderTargets &h = t.block_header.header;
    set4(QPGPU_LT_HEADER_PARENT_HASH, h.parent_hash);
    lt[QPGPU_LT_HEADER_BLOCK_NUMBER] = h.block_number;
    set4(QPGPU_LT_HEADER_STATE_ROOT, h.state_root);
    set4(QPGPU_LT_HEADER_EXTRINSICS_ROOT, h.extrinsics_root);
    set4(QPGPU_LT_HEADER_ZK_TREE_ROOT, h.zk_tree_root);
    for (unsigned i = 0; i < QPGPU_LEAF_DIGEST_LOGS_FELTS; i++) lt[QPGPU_LT_HEADER_DIGEST + i] = h.digest[i];
}

}  // namespace

extern "C" {

static int leaf_circuit_build_impl(unsigned fragment, unsigned min_degree_bits, int inner_hasher, const uint64_t *p2_layout, uint64_t *pack_out, size_t pack_cap_words,
                                   size_t *pack_words, uint64_t *target_map_out, uint64_t *info_out, std::vector<uint64_t> *hint_cells, char *err) {
    auto fail = [&](int code, const std::string &m) { if (err) std::snprintf(err, QPGPU_LEAF_ERR_CAP, "%s", m.c_str()); return code; };
    if (err) err[0] = 0;
    if (!pack_words) return fail(QPGPU_EINVAL, "leaf_circuit_build: null argument");
    if (inner_hasher != hasher::POSEIDON && inner_hasher != hasher::POSEIDON2) return fail(QPGPU_EINVAL, "leaf_circuit_build: unknown inner hasher");
    try {
        cb::Config cfg;                     // wormhole_leaf_circuit_config = standard_recursion_config (common/src/circuit.rs:378-380)
        cfg.min_degree_bits = min_degree_bits;
        cfg.inner_hasher = inner_hasher;
        if (p2_layout) {
            for (int i = 0; i < P2GateLayout::WORDS; i++) if (p2_layout[i] > 0xFFFFFFFFull) return fail(QPGPU_EINVAL, "leaf_circuit_build: Poseidon2 gate layout field out of range");
            P2GateLayout &l = cfg.p2_layout;
            l.w_input = (uint32_t)p2_layout[0]; l.w_output = (uint32_t)p2_layout[1]; l.w_swap = (uint32_t)p2_layout[2]; l.w_delta = (uint32_t)p2_layout[3];
            l.w_full0 = (uint32_t)p2_layout[4]; l.w_partial = (uint32_t)p2_layout[5]; l.w_full1 = (uint32_t)p2_layout[6];
            l.first_round_wires = (uint32_t)p2_layout[7]; l.constraint_order = (uint32_t)p2_layout[8]; l.end_wire = (uint32_t)p2_layout[9];
        }
        Builder b(cfg);
        Target lt[QPGPU_LT_COUNT];
        for (Target &t : lt) t = cb::NO_TARGET;
        size_t gates_after_targets = 0, g1 = 0, g2 = 0, g3 = 0, g4 = 0;
        if (fragment == QPGPU_LEAF_FRAGMENT_FULL) {
            const CircuitTargets targets = circuit_targets(b);
            gates_after_targets = b.num_gates();
            unspendable_account_circuit(targets.unspendable_account, b);
            g1 = b.num_gates();
            zk_merkle_proof_circuit(targets.zk_merkle_proof, b);
            g2 = b.num_gates();
            /* DualExitAccount::circuit: no constraints, the exit accounts are public inputs only */
            b.range_check(targets.block_header.header.block_number, 32);      // BlockHeader::circuit_without_hash_binding
            g3 = b.num_gates();
            connect_shared_targets(targets, b);
            g4 = b.num_gates();
            logical_targets(targets, lt);
        } else if (fragment == QPGPU_LEAF_FRAGMENT_BLOCK_HEADER) {
            // BlockHeaderTargets::new + BlockHeader::circuit (block_header/mod.rs:122-126): the unconditional binding, as the
            // reference's fragment tests compose it (wormhole/tests/src/circuit/block_header_tests.rs:8-19)
            CircuitTargets t{};
            t.block_header.block_hash = b.add_virtual_hash_public_input();
            HeaderTargets &h = t.block_header.header;
            for (auto &e : h.parent_hash) e = b.add_virtual_target();
            h.block_number = b.add_virtual_public_input();
            for (auto &e : h.state_root) e = b.add_virtual_target();
            for (auto &e : h.extrinsics_root) e = b.add_virtual_target();
            for (auto &e : h.zk_tree_root) e = b.add_virtual_target();
            for (auto &e : h.digest) e = b.add_virtual_target();
            b.range_check(h.block_number, 32);
            b.connect_hashes(t.block_header.block_hash, computed_block_hash(t.block_header, b));
            for (int i = 0; i < 4; i++) {
                lt[QPGPU_LT_BLOCK_HASH + i] = t.block_header.block_hash.elements[i]; lt[QPGPU_LT_HEADER_PARENT_HASH + i] = h.parent_hash[i];
                lt[QPGPU_LT_HEADER_STATE_ROOT + i] = h.state_root[i]; lt[QPGPU_LT_HEADER_EXTRINSICS_ROOT + i] = h.extrinsics_root[i];
                lt[QPGPU_LT_HEADER_ZK_TREE_ROOT + i] = h.zk_tree_root[i];
            }
            lt[QPGPU_LT_HEADER_BLOCK_NUMBER] = h.block_number;
            for (unsigned i = 0; i < QPGPU_LEAF_DIGEST_LOGS_FELTS; i++) lt[QPGPU_LT_HEADER_DIGEST + i] = h.digest[i];
        } else if (fragment == QPGPU_LEAF_FRAGMENT_UNSPENDABLE_ACCOUNT) {
            // UnspendableAccountTargets::new + UnspendableAccount::circuit (wormhole/tests/src/circuit/unspendable_account_tests.rs:26-40)
            UnspendableAccountTargets t;
            t.account_id = b.add_virtual_hash();
            t.secret = b.add_virtual_hash();
            unspendable_account_circuit(t, b);
            for (int i = 0; i < 4; i++) { lt[QPGPU_LT_UNSPENDABLE_ACCOUNT_ID + i] = t.account_id.elements[i]; lt[QPGPU_LT_UNSPENDABLE_SECRET + i] = t.secret.elements[i]; }
        } else if (fragment == QPGPU_LEAF_FRAGMENT_NULLIFIER) {
            // NullifierTargets::new + Nullifier::circuit, the unconditional binding (nullifier.rs:327-343; nullifier_tests.rs)
            NullifierTargets t;
            t.hash = b.add_virtual_hash_public_input();
            t.secret = b.add_virtual_hash();
            for (auto &e : t.transfer_count) e = b.add_virtual_target();
            b.connect_hashes(t.hash, computed_nullifier(t, b));
            for (int i = 0; i < 4; i++) { lt[QPGPU_LT_NULLIFIER_HASH + i] = t.hash.elements[i]; lt[QPGPU_LT_NULLIFIER_SECRET + i] = t.secret.elements[i]; }
            for (int i = 0; i < 2; i++) lt[QPGPU_LT_NULLIFIER_TRANSFER_COUNT + i] = t.transfer_count[i];
        } else if (fragment == QPGPU_LEAF_FRAGMENT_FAKE_LEAF) {
            // build_fake_leaf_circuit (wormhole/tests/test-helpers/src/fake_leaf.rs:20-39): 21 free public inputs in the leaf's layout,
            // the three 32-bit range checks — the reference's way of testing the aggregation layers on arbitrary leaf public inputs
            const std::vector<Target> pis = b.add_virtual_targets(21);
            b.range_check(pis[1], 32); b.range_check(pis[2], 32); b.range_check(pis[3], 32);
            for (Target t : pis) b.register_public_input(t);
        } else return fail(QPGPU_EINVAL, "leaf_circuit_build: unknown fragment");
        CircuitPack pack;
        const std::string why = b.build(pack);
        if (!why.empty()) return fail(QPGPU_EINVAL, "leaf_circuit_build: " + why);
        const std::vector<uint64_t> words = pack.serialize();
        *pack_words = words.size();
        if (pack_out) {
            if (pack_cap_words < words.size()) return fail(QPGPU_EBUFSIZE, "leaf_circuit_build: pack buffer too small");
            std::memcpy(pack_out, words.data(), words.size() * 8);
        }
        if (target_map_out) {
            for (unsigned i = 0; i < QPGPU_LT_COUNT; i++) { const u64 c = lt[i] == cb::NO_TARGET ? cb::NO_CELL : b.cell_of(lt[i]); target_map_out[i] = c == cb::NO_CELL ? UINT64_MAX : c; }
        }
        if (hint_cells) {
            // the 12 output cells of every Poseidon2 row, hash call sites in the order of their tags, rows of a site in sponge order
            std::vector<std::pair<int, uint32_t>> rows = b.poseidon2_rows();
            std::stable_sort(rows.begin(), rows.end(), [](const std::pair<int, uint32_t> &x, const std::pair<int, uint32_t> &y) { return x.first < y.first; });
            hint_cells->clear();
            for (const auto &r : rows) {
                if (r.first < 0) continue;                     // (the public-input hash of a Poseidon2-hashed circuit: not a call site of the leaf)
                for (uint32_t i = 0; i < 12; i++) hint_cells->push_back(b.poseidon2_output_cell(r.second, i));
            }
            for (Target t : b.hint_targets()) hint_cells->push_back(b.cell_of(t));      // then the running hash of the Merkle walk after every level
        }
        if (info_out) {
            // the figures the reference's GateProfiler prints (wormhole/circuit/src/profile.rs): gates per fragment, rows per gate type
            const std::map<uint64_t, size_t> gc = b.gate_counts();
            auto cnt = [&](uint64_t t) { auto it = gc.find(t); return it == gc.end() ? (uint64_t)0 : (uint64_t)it->second; };
            const uint64_t info[QPGPU_LEAF_CIRCUIT_INFO_WORDS] = {pack.degree_bits, b.rows_before_padding(), gates_after_targets, g1 - gates_after_targets, g2 - g1, g3 - g2, g4 - g3,
                                                                 cnt(GATE_ARITHMETIC), cnt(GATE_BASE_SUM), cnt(GATE_POSEIDON2), cnt(GATE_POSEIDON), cnt(GATE_CONSTANT),
                                                                 cnt(GATE_PUBLIC_INPUT), cnt(GATE_NOOP), (uint64_t)pack.hints.size(), (uint64_t)pack.num_selectors};
            std::memcpy(info_out, info, sizeof info);
        }
    } catch (const std::exception &e) {
        return fail(QPGPU_EINVAL, std::string("leaf_circuit_build: ") + e.what());
    }
    return QPGPU_OK;
}

int qpgpu_leaf_circuit_build(unsigned fragment, unsigned min_degree_bits, int inner_hasher, const uint64_t *p2_layout, uint64_t *pack_out, size_t pack_cap_words,
                             size_t *pack_words, uint64_t *target_map_out, uint64_t *info_out, char *err) {
    return leaf_circuit_build_impl(fragment, min_degree_bits, inner_hasher, p2_layout, pack_out, pack_cap_words, pack_words, target_map_out, info_out, nullptr, err);
}

int qpgpu_leaf_circuit_hash_hint_cells(unsigned min_degree_bits, int inner_hasher, const uint64_t *p2_layout, uint64_t *cells_out, size_t cap, size_t *count, char *err) {
    if (err) err[0] = 0;
    if (!count) { if (err) std::snprintf(err, QPGPU_LEAF_ERR_CAP, "leaf_circuit_hash_hint_cells: null argument"); return QPGPU_EINVAL; }
    std::vector<uint64_t> cells;
    size_t words = 0;
    const int rc = leaf_circuit_build_impl(QPGPU_LEAF_FRAGMENT_FULL, min_degree_bits, inner_hasher, p2_layout, nullptr, 0, &words, nullptr, nullptr, &cells, err);
    if (rc != QPGPU_OK) return rc;
    *count = cells.size();
    if (cells.size() != QPGPU_LEAF_HASH_HINTS) { if (err) std::snprintf(err, QPGPU_LEAF_ERR_CAP, "leaf_circuit_hash_hint_cells: %zu cells where %d are expected", cells.size(), QPGPU_LEAF_HASH_HINTS); return QPGPU_EINVAL; }
    if (cells_out) {
        if (cap < cells.size()) { if (err) std::snprintf(err, QPGPU_LEAF_ERR_CAP, "leaf_circuit_hash_hint_cells: cell buffer too small"); return QPGPU_EBUFSIZE; }
        std::memcpy(cells_out, cells.data(), cells.size() * 8);
    }
    return QPGPU_OK;
}

// WormholeProver::commit (wormhole/prover/src/lib.rs:156-163): CircuitInputs -> the PartialWitness as (cell, value) pairs of a
// circuit built by qpgpu_leaf_circuit_build, plus the 21 public inputs
int qpgpu_leaf_commit(const qpgpu_leaf_inputs *in, const uint64_t *target_map, uint64_t *cells_out, uint64_t *values_out, size_t cap,
                      size_t *count, uint64_t public_inputs_out[QPGPU_LEAF_PUBLIC_INPUTS], char *err) {
    if (err) err[0] = 0;
    if (!target_map || !cells_out || !values_out || !count || cap < QPGPU_LT_COUNT) {
        if (err) std::snprintf(err, QPGPU_LEAF_ERR_CAP, "leaf_commit: null argument or room for fewer than %d assignments", QPGPU_LT_COUNT);
        return -1;
    }
    uint32_t targets[QPGPU_LT_COUNT];
    uint64_t values[QPGPU_LT_COUNT];
    size_t n = 0;
    const int rc = qpgpu_leaf_fill_witness(in, public_inputs_out, targets, values, QPGPU_LT_COUNT, &n, err);
    if (rc == 0) *count = qpgpu_leaf_map_targets(targets, values, n, target_map, QPGPU_LT_COUNT, cells_out, values_out);
    std::memset(values, 0, sizeof values);          // the assignments carry the spend secret
    return rc;
}

}  // extern "C"
