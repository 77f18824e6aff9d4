// leaf_witness.cpp — native leaf-witness front-end (host only): CircuitInputs -> 21 public inputs + the PartialWitness
// assignments of the Wormhole leaf circuit, the byte <-> felt codecs, and the fork's Poseidon2 sponge on an injected
// parameter block. C ABI and reference citations: include/qpgpu_leaf.h.
//
// fill_witness (wormhole/prover/src/lib.rs:187-221) is pure re-encoding and does not hash; the derive-a-hash helpers at the
// end do, on qp-poseidon-core's parameter set (poseidon2::qp_params, pinned by all seven of the reference's known-answer
// vectors) or on a caller-supplied block.
#include <cstdio>
#include <algorithm>
#include <array>
#include <cstring>
#include <vector>
#include "../../include/qpgpu_leaf.h"
#include "gl64.hpp"
#include "poseidon.hpp"

using gl::u64;

namespace {

constexpr size_t MAX_SERIALIZED_BYTES = 1u << 20;          // common/src/serialization.rs:27

u64 load_le64(const uint8_t *p) { u64 v = 0; for (int k = 0; k < 8; k++) v |= (u64)p[k] << (8 * k); return v; }
void store_le64(u64 v, uint8_t *p) { for (int k = 0; k < 8; k++) p[k] = (uint8_t)(v >> (8 * k)); }

int fail(char *err, const char *fmt, unsigned long long a = 0, unsigned long long b = 0) {
    if (err) std::snprintf(err, QPGPU_LEAF_ERR_CAP, fmt, a, b);
    return -1;
}

bool parse_params(const uint64_t *params, size_t n_words, poseidon2::Params &p) {
    if (!params && n_words == 0) { p = poseidon2::qp_params(); return true; }     // the pinned qp-poseidon-core set
    if (!params || n_words != (size_t)poseidon2::PARAM_WORDS) return false;
    const uint64_t *w = params;
    for (int i = 0; i < 96; i++) p.rc_ext[i] = gl::canon(*w++);
    for (int i = 0; i < 22; i++) p.rc_int[i] = gl::canon(*w++);
    for (int i = 0; i < 12; i++) p.diag_m1[i] = gl::canon(*w++);
    for (int i = 0; i < 16; i++) p.m4[i] = gl::canon(*w++);
    return true;
}

// Poseidon2Hash::hash_no_pad of the fork: `input || 1 || 0*` to a multiple of the rate, each block ADDED into the rate
// part of the state (the block-header known-answer vectors, 45 elements = 6 blocks, tell additive from overwrite absorption;
// single-block inputs cannot)
void hash_pad10(const poseidon2::Params &p, const u64 *in, size_t n, u64 out[4]) {
    u64 st[12] = {0};
    const size_t padded = (n + 1 + 7) / 8 * 8;
    for (size_t i = 0; i < padded; i += 8) {
        for (size_t j = 0; j < 8; j++) {
            const size_t k = i + j;
            st[j] = gl::canon(gl::add(st[j], k < n ? gl::canon(in[k]) : (k == n ? 1 : 0)));
        }
        poseidon2::permute(st, p);
    }
    for (int i = 0; i < 4; i++) out[i] = gl::canon(st[i]);
    std::memset(st, 0, sizeof st);      // the state absorbed a secret in the callers below
}

}  // namespace

extern "C" {

size_t qpgpu_bytes_to_felts(const uint8_t *in, size_t len, uint64_t *out, size_t out_cap) {
    if ((!in && len) || len > MAX_SERIALIZED_BYTES) return 0;
    const size_t n = (len + 1 + 3) / 4;
    if (!out || out_cap < n) return 0;
    for (size_t i = 0; i < n; i++) {
        uint32_t v = 0;
        for (size_t k = 0; k < 4; k++) {
            const size_t idx = 4 * i + k;
            const uint8_t b = idx < len ? in[idx] : (idx == len ? 0x01 : 0x00);
            v |= (uint32_t)b << (8 * k);
        }
        out[i] = v;
    }
    return n;
}

size_t qpgpu_felts_to_bytes(const uint64_t *in, size_t n, uint8_t *out, size_t out_cap) {
    if (!in || n == 0 || n > MAX_SERIALIZED_BYTES / 4 + 1) return (size_t)-1;
    std::vector<uint8_t> buf(4 * n);
    for (size_t i = 0; i < n; i++) {
        const u64 v = gl::canon(in[i]);
        if (v > 0xFFFFFFFFull) return (size_t)-1;
        for (size_t k = 0; k < 4; k++) buf[4 * i + k] = (uint8_t)(v >> (8 * k));
    }
    // the terminator is the last non-zero byte and sits in the last element
    size_t end = buf.size();
    while (end > 0 && buf[end - 1] == 0) end--;
    if (end == 0 || buf[end - 1] != 0x01 || end - 1 < 4 * (n - 1)) return (size_t)-1;
    const size_t len = end - 1;
    if (!out || out_cap < len) return (size_t)-1;
    std::memcpy(out, buf.data(), len);
    return len;
}

void qpgpu_bytes_to_digest(const uint8_t in[32], uint64_t out[4]) {
    if (!in || !out) return;                       // (every entry point of the ABI takes NULL without faulting: tools/null_arg_sweep.py)
    for (int i = 0; i < 4; i++) out[i] = gl::canon(load_le64(in + 8 * i));   // from_noncanonical_u64, serialised canonical
}
void qpgpu_digest_to_bytes(const uint64_t in[4], uint8_t out[32]) {
    if (!in || !out) return;
    for (int i = 0; i < 4; i++) store_le64(gl::canon(in[i]), out + 8 * i);
}
int qpgpu_bytes_digest_is_canonical(const uint8_t in[32]) {
    if (!in) return 0;
    for (int i = 0; i < 4; i++) if (load_le64(in + 8 * i) >= gl::P) return 0;
    return 1;
}
void qpgpu_u64_to_felts(uint64_t v, uint64_t out[2]) { if (!out) return; out[0] = v >> 32; out[1] = v & 0xFFFFFFFFull; }
void qpgpu_u128_to_felts(uint64_t hi, uint64_t lo, uint64_t out[4]) {
    if (!out) return;
    out[0] = hi >> 32; out[1] = hi & 0xFFFFFFFFull; out[2] = lo >> 32; out[3] = lo & 0xFFFFFFFFull;
}

int qpgpu_leaf_is_not_dummy(const qpgpu_leaf_inputs *in) {
    if (!in) return 0;
    static const uint8_t zero[32] = {0};
    return !(std::memcmp(in->block_hash, zero, 32) == 0 && in->output_amount_1 == 0 && in->output_amount_2 == 0);
}

int qpgpu_leaf_fill_witness(const qpgpu_leaf_inputs *in, uint64_t public_inputs_out[QPGPU_LEAF_PUBLIC_INPUTS],
                            uint32_t *targets_out, uint64_t *values_out, size_t cap, size_t *count, char *err) {
    if (err) err[0] = 0;
    if (!in || !public_inputs_out || !targets_out || !values_out || !count) return fail(err, "fill_witness: null argument");
    if (cap < QPGPU_LT_COUNT) return fail(err, "fill_witness: room for %llu assignments needed, %llu given", QPGPU_LT_COUNT, cap);
    // fill_witness / ZkMerkleProofData::try_from: bounds before any work
    if (in->zk_merkle_depth > QPGPU_LEAF_MAX_DEPTH)
        return fail(err, "ZK Merkle proof depth %llu exceeds maximum supported depth %llu", in->zk_merkle_depth, QPGPU_LEAF_MAX_DEPTH);
    for (uint32_t l = 0; l < in->zk_merkle_depth; l++)
        if (in->zk_merkle_positions[l] > 3)
            return fail(err, "ZK Merkle proof position %llu at level %llu is invalid (must be 0-3)", in->zk_merkle_positions[l], l);
    // BytesDigest::try_from on every field of that type (wormhole/inputs/src/lib.rs:148-167)
    struct { const uint8_t *p; const char *name; } digests[] = {
        {in->nullifier, "nullifier"}, {in->exit_account_1, "exit_account_1"}, {in->exit_account_2, "exit_account_2"},
        {in->block_hash, "block_hash"}, {in->secret, "secret"}, {in->unspendable_account, "unspendable_account"},
        {in->parent_hash, "parent_hash"}, {in->state_root, "state_root"}, {in->extrinsics_root, "extrinsics_root"}};
    for (auto &d : digests)
        for (int i = 0; i < 4; i++)
            if (load_le64(d.p + 8 * i) >= gl::P) {
                if (err) std::snprintf(err, QPGPU_LEAF_ERR_CAP, "%s: chunk %d is out of the field range", d.name, i);
                return -1;
            }

    size_t n = 0;
    auto set = [&](uint32_t target, u64 v) { targets_out[n] = target; values_out[n] = gl::canon(v); n++; };
    auto set4 = [&](uint32_t base, const uint8_t bytes[32]) { u64 f[4]; qpgpu_bytes_to_digest(bytes, f); for (int i = 0; i < 4; i++) set(base + i, f[i]); };
    u64 tc[2];
    qpgpu_u64_to_felts(in->transfer_count, tc);

    // Nullifier::fill_targets: hash, secret, transfer_count
    set4(QPGPU_LT_NULLIFIER_HASH, in->nullifier);
    set4(QPGPU_LT_NULLIFIER_SECRET, in->secret);
    set(QPGPU_LT_NULLIFIER_TRANSFER_COUNT, tc[0]); set(QPGPU_LT_NULLIFIER_TRANSFER_COUNT + 1, tc[1]);
    // UnspendableAccount::fill_targets: account_id, secret
    set4(QPGPU_LT_UNSPENDABLE_ACCOUNT_ID, in->unspendable_account);
    set4(QPGPU_LT_UNSPENDABLE_SECRET, in->secret);
    // ZkMerkleProofData::fill_targets: root, depth, siblings + position per level (zero padded to MAX_DEPTH), leaf
    set4(QPGPU_LT_ZK_ROOT_HASH, in->zk_tree_root);
    set(QPGPU_LT_ZK_DEPTH, in->zk_merkle_depth);
    static const uint8_t zero32[32] = {0};
    for (uint32_t l = 0; l < QPGPU_LEAF_MAX_DEPTH; l++) {
        for (uint32_t s = 0; s < 3; s++) set4(QPGPU_LT_ZK_SIBLINGS + (l * 3 + s) * 4, l < in->zk_merkle_depth ? in->zk_merkle_siblings[l][s] : zero32);
        set(QPGPU_LT_ZK_POSITIONS + l, l < in->zk_merkle_depth ? in->zk_merkle_positions[l] : 0);
    }
    set4(QPGPU_LT_LEAF_TO_ACCOUNT, in->unspendable_account);
    set(QPGPU_LT_LEAF_TRANSFER_COUNT, tc[0]); set(QPGPU_LT_LEAF_TRANSFER_COUNT + 1, tc[1]);
    set(QPGPU_LT_LEAF_ASSET_ID, in->asset_id);
    set(QPGPU_LT_LEAF_INPUT_AMOUNT, in->input_amount);
    set(QPGPU_LT_LEAF_OUTPUT_AMOUNT_1, in->output_amount_1);
    set(QPGPU_LT_LEAF_OUTPUT_AMOUNT_2, in->output_amount_2);
    set(QPGPU_LT_LEAF_VOLUME_FEE_BPS, in->volume_fee_bps);
    // DualExitAccount::fill_targets
    set4(QPGPU_LT_EXIT_ACCOUNT_1, in->exit_account_1);
    set4(QPGPU_LT_EXIT_ACCOUNT_2, in->exit_account_2);
    // BlockHeader::fill_targets: block_hash, then the header preimage fields
    set4(QPGPU_LT_BLOCK_HASH, in->block_hash);
    set4(QPGPU_LT_HEADER_PARENT_HASH, in->parent_hash);
    set(QPGPU_LT_HEADER_BLOCK_NUMBER, in->block_number);
    set4(QPGPU_LT_HEADER_STATE_ROOT, in->state_root);
    set4(QPGPU_LT_HEADER_EXTRINSICS_ROOT, in->extrinsics_root);
    set4(QPGPU_LT_HEADER_ZK_TREE_ROOT, in->zk_tree_root);
    {
        u64 dg[QPGPU_LEAF_DIGEST_LOGS_FELTS];
        if (qpgpu_bytes_to_felts(in->digest, QPGPU_LEAF_DIGEST_LOGS_SIZE, dg, QPGPU_LEAF_DIGEST_LOGS_FELTS) != QPGPU_LEAF_DIGEST_LOGS_FELTS)
            return fail(err, "failed to encode digest logs");
        for (uint32_t i = 0; i < QPGPU_LEAF_DIGEST_LOGS_FELTS; i++) set(QPGPU_LT_HEADER_DIGEST + i, dg[i]);
    }
    *count = n;

    // the 21 public inputs in registration order (wormhole/inputs/src/lib.rs:68-80)
    u64 *pi = public_inputs_out;
    pi[0] = in->asset_id; pi[1] = in->output_amount_1; pi[2] = in->output_amount_2; pi[3] = in->volume_fee_bps;
    qpgpu_bytes_to_digest(in->nullifier, pi + 4);
    qpgpu_bytes_to_digest(in->exit_account_1, pi + 8);
    qpgpu_bytes_to_digest(in->exit_account_2, pi + 12);
    qpgpu_bytes_to_digest(in->block_hash, pi + 16);
    pi[20] = in->block_number;
    return 0;
}

size_t qpgpu_leaf_map_targets(const uint32_t *targets, const uint64_t *values, size_t count, const uint64_t *target_map,
                              size_t map_len, uint64_t *cells_out, uint64_t *cell_values_out) {
    if (!targets || !values || !target_map || !cells_out || !cell_values_out) return 0;
    size_t n = 0;
    for (size_t i = 0; i < count; i++) {
        if (targets[i] >= map_len || target_map[targets[i]] == UINT64_MAX) continue;   // a target the builder dropped
        cells_out[n] = target_map[targets[i]]; cell_values_out[n] = values[i]; n++;
    }
    return n;
}

// ---- Poseidon2 of the fork: params == NULL && n_words == 0 selects the pinned qp-poseidon-core set ----
size_t qpgpu_poseidon2_qp_params(uint64_t *out, size_t cap) {
    if (out && cap >= (size_t)poseidon2::PARAM_WORDS) {
        const poseidon2::Params &p = poseidon2::qp_params();
        uint64_t *w = out;
        for (int i = 0; i < 96; i++) *w++ = p.rc_ext[i];
        for (int i = 0; i < 22; i++) *w++ = p.rc_int[i];
        for (int i = 0; i < 12; i++) *w++ = p.diag_m1[i];
        for (int i = 0; i < 16; i++) *w++ = p.m4[i];
    }
    return poseidon2::PARAM_WORDS;
}
int qpgpu_poseidon2_permute(const uint64_t *params, size_t n_words, uint64_t state[12]) {
    poseidon2::Params p;
    if (!state || !parse_params(params, n_words, p)) return -1;
    u64 s[12];
    for (int i = 0; i < 12; i++) s[i] = gl::canon(state[i]);
    poseidon2::permute(s, p);
    for (int i = 0; i < 12; i++) state[i] = s[i];
    return 0;
}
int qpgpu_poseidon2_hash_pad10(const uint64_t *params, size_t n_words, const uint64_t *in, size_t n, uint64_t out[4]) {
    poseidon2::Params p;
    if ((!in && n) || !out || !parse_params(params, n_words, p)) return -1;
    hash_pad10(p, in, n, out);
    return 0;
}
int qpgpu_poseidon2_hash_bytes(const uint64_t *params, size_t n_words, const uint64_t *in, size_t n, uint8_t out[32]) {
    u64 h[4];
    if (!out || qpgpu_poseidon2_hash_pad10(params, n_words, in, n, h)) return -1;
    qpgpu_digest_to_bytes(h, out);
    return 0;
}

static int double_hash(const uint64_t *params, size_t n_words, const char *salt, const u64 *tail, size_t n_tail, uint8_t out[32]) {
    poseidon2::Params p;
    if (!out || !parse_params(params, n_words, p)) return -1;
    u64 pre[3 + 6];
    const size_t ns = qpgpu_bytes_to_felts((const uint8_t *)salt, std::strlen(salt), pre, 3);   // 8 bytes -> 3 elements
    if (ns != 3 || n_tail > 6) return -1;
    for (size_t i = 0; i < n_tail; i++) pre[3 + i] = tail[i];
    u64 inner[4], outer[4];
    hash_pad10(p, pre, 3 + n_tail, inner);
    hash_pad10(p, inner, 4, outer);
    qpgpu_digest_to_bytes(outer, out);
    std::memset(pre, 0, sizeof pre); std::memset(inner, 0, sizeof inner);   // the preimage holds the spend secret
    return 0;
}
int qpgpu_leaf_unspendable_account(const uint64_t *params, size_t n_words, const uint8_t secret[32], uint8_t out[32]) {
    if (!secret) return -1;
    u64 s[4];
    qpgpu_bytes_to_digest(secret, s);
    const int rc = double_hash(params, n_words, "wormhole", s, 4, out);
    std::memset(s, 0, sizeof s);
    return rc;
}
int qpgpu_leaf_nullifier(const uint64_t *params, size_t n_words, const uint8_t secret[32], uint64_t transfer_count, uint8_t out[32]) {
    if (!secret) return -1;
    u64 t[6];
    qpgpu_bytes_to_digest(secret, t);
    qpgpu_u64_to_felts(transfer_count, t + 4);
    const int rc = double_hash(params, n_words, "~nullif~", t, 6, out);
    std::memset(t, 0, sizeof t);
    return rc;
}
int qpgpu_leaf_block_hash(const uint64_t *params, size_t n_words, const uint8_t parent_hash[32], uint32_t block_number,
                          const uint8_t state_root[32], const uint8_t extrinsics_root[32], const uint8_t zk_tree_root[32],
                          const uint8_t digest[QPGPU_LEAF_DIGEST_LOGS_SIZE], uint8_t out[32]) {
    poseidon2::Params p;
    if (!parent_hash || !state_root || !extrinsics_root || !zk_tree_root || !digest || !out || !parse_params(params, n_words, p)) return -1;
    u64 pre[45], h[4];
    qpgpu_bytes_to_digest(parent_hash, pre);
    pre[4] = block_number;
    qpgpu_bytes_to_digest(state_root, pre + 5);
    qpgpu_bytes_to_digest(extrinsics_root, pre + 9);
    qpgpu_bytes_to_digest(zk_tree_root, pre + 13);
    if (qpgpu_bytes_to_felts(digest, QPGPU_LEAF_DIGEST_LOGS_SIZE, pre + 17, 28) != 28) return -1;
    hash_pad10(p, pre, 45, h);
    qpgpu_digest_to_bytes(h, out);
    return 0;
}


// ---- hash hints (include/qpgpu_leaf.h): every sponge state of the leaf circuit's 8 hash call sites, in tag order ----
namespace {
// hash_pad10 that also writes the 12-element state after every permutation to *out (advanced)
void sponge_states(const poseidon2::Params &p, const u64 *in, size_t n, u64 *&out, u64 digest[4]) {
    u64 st[12] = {0};
    const size_t padded = (n + 1 + 7) / 8 * 8;
    for (size_t i = 0; i < padded; i += 8) {
        for (size_t j = 0; j < 8; j++) {
            const size_t k = i + j;
            st[j] = gl::canon(gl::add(st[j], k < n ? gl::canon(in[k]) : (k == n ? 1 : 0)));
        }
        poseidon2::permute_qp(st, p);      // (p is the qp set here: its external layer needs no multiplications)
        for (int j = 0; j < 12; j++) *out++ = gl::canon(st[j]);
    }
    for (int i = 0; i < 4; i++) digest[i] = gl::canon(st[i]);
    std::memset(st, 0, sizeof st);
}
}  // namespace

int qpgpu_leaf_hash_hints(const qpgpu_leaf_inputs *in, uint64_t *values_out, size_t cap, size_t *count, char *err) {
    if (err) err[0] = 0;
    if (!in || !values_out || !count) return fail(err, "leaf_hash_hints: null argument");
    if (cap < QPGPU_LEAF_HASH_HINTS) return fail(err, "leaf_hash_hints: room for %llu values needed, %llu given", (unsigned long long)QPGPU_LEAF_HASH_HINTS, (unsigned long long)cap);
    {   // the same refusals as fill_witness (depth, positions, field range of every digest), before any hashing
        uint32_t t[QPGPU_LT_COUNT]; uint64_t v[QPGPU_LT_COUNT], pis[QPGPU_LEAF_PUBLIC_INPUTS]; size_t n = 0;
        const int rc = qpgpu_leaf_fill_witness(in, pis, t, v, QPGPU_LT_COUNT, &n, err);
        std::memset(v, 0, sizeof v);
        if (rc) return rc;
    }
    const poseidon2::Params &p = poseidon2::qp_params();
    u64 *out = values_out;
    u64 secret[4], account[4], tc[2], pre[48], inner[4], digest[4];
    qpgpu_bytes_to_digest(in->secret, secret);
    qpgpu_bytes_to_digest(in->unspendable_account, account);
    qpgpu_u64_to_felts(in->transfer_count, tc);
    // QPGPU_LEAF_HASH_UNSPENDABLE_INNER / _OUTER
    if (qpgpu_bytes_to_felts((const uint8_t *)"wormhole", 8, pre, 3) != 3) return fail(err, "leaf_hash_hints: salt encoding");
    std::memcpy(pre + 3, secret, 32);
    sponge_states(p, pre, 7, out, inner);
    sponge_states(p, inner, 4, out, digest);
    // QPGPU_LEAF_HASH_ZK_LEAF: the leaf's to_account is the unspendable account's target class, i.e. the assigned account id
    std::memcpy(pre, account, 32); pre[4] = tc[0]; pre[5] = tc[1]; pre[6] = in->asset_id; pre[7] = in->input_amount;
    u64 cur[4], walk[4 * QPGPU_LEAF_MAX_DEPTH];
    sponge_states(p, pre, 8, out, cur);
    // QPGPU_LEAF_HASH_MERKLE_LEVEL_l: the running hash inserted among the level's sorted siblings at the hinted position; levels past
    // the depth hash what fill_witness pads them with (zero siblings, position 0) and leave the running hash as it is
    for (uint32_t l = 0; l < QPGPU_LEAF_MAX_DEPTH; l++) {
        const bool active = l < in->zk_merkle_depth;
        const unsigned pos = active ? in->zk_merkle_positions[l] : 0;
        u64 sib[3][4] = {{0}};
        if (active) for (int k = 0; k < 3; k++) qpgpu_bytes_to_digest(in->zk_merkle_siblings[l][k], sib[k]);
        for (unsigned slot = 0, k = 0; slot < 4; slot++) {
            const u64 *child = slot == pos ? cur : sib[k++];
            std::memcpy(pre + 4 * slot, child, 32);
        }
        u64 parent[4];
        sponge_states(p, pre, 16, out, parent);
        if (active) std::memcpy(cur, parent, 32);
        std::memcpy(walk + 4 * l, cur, 32);
    }
    // QPGPU_LEAF_HASH_NULLIFIER_INNER / _OUTER
    if (qpgpu_bytes_to_felts((const uint8_t *)"~nullif~", 8, pre, 3) != 3) return fail(err, "leaf_hash_hints: salt encoding");
    std::memcpy(pre + 3, secret, 32); pre[7] = tc[0]; pre[8] = tc[1];
    sponge_states(p, pre, 9, out, inner);
    sponge_states(p, inner, 4, out, digest);
    // QPGPU_LEAF_HASH_BLOCK_HEADER
    qpgpu_bytes_to_digest(in->parent_hash, pre);
    pre[4] = in->block_number;
    qpgpu_bytes_to_digest(in->state_root, pre + 5);
    qpgpu_bytes_to_digest(in->extrinsics_root, pre + 9);
    qpgpu_bytes_to_digest(in->zk_tree_root, pre + 13);
    if (qpgpu_bytes_to_felts(in->digest, QPGPU_LEAF_DIGEST_LOGS_SIZE, pre + 17, 28) != 28) return fail(err, "failed to encode digest logs");
    sponge_states(p, pre, 45, out, digest);
    for (unsigned i = 0; i < 4 * QPGPU_LEAF_MAX_DEPTH; i++) *out++ = walk[i];      // the running hash after every level of the walk
    std::memset(secret, 0, sizeof secret); std::memset(pre, 0, sizeof pre); std::memset(inner, 0, sizeof inner); std::memset(digest, 0, sizeof digest);   // the spend secret and what is derived from it alone
    *count = (size_t)(out - values_out);
    if (*count != QPGPU_LEAF_HASH_HINTS) return fail(err, "leaf_hash_hints: %llu values where %llu are expected", (unsigned long long)*count, (unsigned long long)QPGPU_LEAF_HASH_HINTS);
    return 0;
}

// ---- common/src/zk_merkle.rs ----
static bool canonical32(const uint8_t *h) { return qpgpu_bytes_digest_is_canonical(h) == 1; }

int qpgpu_zk_leaf_hash(const uint8_t to_account[32], uint64_t transfer_count, uint32_t asset_id, uint32_t input_amount, uint8_t out[32]) {
    if (!to_account || !out) return -1;
    u64 pre[8];
    qpgpu_bytes_to_digest(to_account, pre);
    qpgpu_u64_to_felts(transfer_count, pre + 4);
    pre[6] = asset_id; pre[7] = input_amount;
    const int rc = qpgpu_poseidon2_hash_bytes(nullptr, 0, pre, 8, out);
    std::memset(pre, 0, sizeof pre);       // the deposit's account and amount
    return rc;
}

int qpgpu_zk_hash_node_presorted(const uint8_t *children, uint8_t out[32]) {
    if (!children || !out) return -1;
    u64 limbs[16];
    for (int c = 0; c < 4; c++) {
        if (!canonical32(children + 32 * c)) return -1;         // hash_bytes_compact rejects a limb >= p
        qpgpu_bytes_to_digest(children + 32 * c, limbs + 4 * c);
    }
    return qpgpu_poseidon2_hash_bytes(nullptr, 0, limbs, 16, out);
}

int qpgpu_zk_hash_node(const uint8_t *children, uint8_t out[32]) {
    if (!children || !out) return -1;
    std::array<std::array<uint8_t, 32>, 4> sorted;      // [u8; 32]'s Ord: byte-lexicographic
    std::memcpy(sorted.data(), children, 128);
    std::sort(sorted.begin(), sorted.end());
    return qpgpu_zk_hash_node_presorted(sorted[0].data(), out);
}

int qpgpu_zk_insert_at_position(const uint8_t current[32], const uint8_t *sibs, unsigned position, uint8_t *out) {
    if (!current || !sibs || !out || position > 3) return -1;
    for (unsigned slot = 0, k = 0; slot < 4; slot++) {
        if (slot == position) std::memcpy(out + 32 * slot, current, 32);
        else std::memcpy(out + 32 * slot, sibs + 32 * k++, 32);
    }
    return 0;
}

int qpgpu_zk_proof_verify(const uint8_t leaf_hash[32], const uint8_t *siblings, const uint8_t *positions, size_t depth, const uint8_t root[32]) {
    if (!leaf_hash || !root || depth > QPGPU_LEAF_MAX_DEPTH || (depth && (!siblings || !positions))) return 0;
    if (!canonical32(leaf_hash)) return 0;
    for (size_t i = 0; i < depth * 3; i++) if (!canonical32(siblings + 32 * i)) return 0;
    uint8_t cur[32], four[128];
    std::memcpy(cur, leaf_hash, 32);
    for (size_t l = 0; l < depth; l++) {
        if (qpgpu_zk_insert_at_position(cur, siblings + 96 * l, positions[l], four)) return 0;
        if (qpgpu_zk_hash_node_presorted(four, cur)) return 0;
    }
    return std::memcmp(cur, root, 32) == 0;
}

int qpgpu_zk_proof_from_unsorted(const uint8_t leaf_hash[32], const uint8_t *unsorted, size_t depth, uint8_t *sorted_out, uint8_t *positions_out,
                                 uint8_t root_out[32], char *err) {
    auto fail = [&](const char *m) { if (err) std::snprintf(err, QPGPU_LEAF_ERR_CAP, "%s", m); return -1; };
    if (!leaf_hash || !root_out || (depth && (!unsorted || !sorted_out || !positions_out))) return fail("null argument");
    if (depth > QPGPU_LEAF_MAX_DEPTH) return fail("from_unsorted: proof depth exceeds MAX_DEPTH");
    if (!canonical32(leaf_hash)) return fail("from_unsorted: leaf hash bytes are noncanonical");
    for (size_t i = 0; i < depth * 3; i++) if (!canonical32(unsorted + 32 * i)) return fail("from_unsorted: sibling hash bytes are noncanonical");
    uint8_t cur[32];
    std::memcpy(cur, leaf_hash, 32);
    for (size_t l = 0; l < depth; l++) {
        std::array<std::array<uint8_t, 32>, 4> four;
        std::memcpy(four[0].data(), cur, 32);
        std::memcpy(four[1].data(), unsorted + 96 * l, 96);
        std::sort(four.begin(), four.end());
        unsigned pos = 0;
        while (std::memcmp(four[pos].data(), cur, 32)) pos++;   // the first slot holding the running hash
        positions_out[l] = (uint8_t)pos;
        for (unsigned slot = 0, k = 0; slot < 4; slot++) if (slot != pos) std::memcpy(sorted_out + 96 * l + 32 * k++, four[slot].data(), 32);
        if (qpgpu_zk_hash_node_presorted(four[0].data(), cur)) return fail("from_unsorted: node hash failed");
    }
    std::memcpy(root_out, cur, 32);
    return 0;
}

// ---- the leaf circuit's constraints on CircuitInputs ----
int qpgpu_leaf_check_constraints(const qpgpu_leaf_inputs *in, char *err) {
    auto fail = [&](int code, const char *m) { if (err) std::snprintf(err, QPGPU_LEAF_ERR_CAP, "%s", m); return code; };
    if (!in) return fail(-1, "null argument");
    {   // the same input validation as fill_witness (depth, positions, canonical digests)
        uint64_t pis[QPGPU_LEAF_PUBLIC_INPUTS];
        std::vector<uint32_t> t(QPGPU_LT_COUNT);
        std::vector<uint64_t> v(QPGPU_LT_COUNT);
        size_t cnt = 0;
        const int rc = qpgpu_leaf_fill_witness(in, pis, t.data(), v.data(), QPGPU_LT_COUNT, &cnt, err);
        for (auto &x : v) x = 0;
        if (rc) return -1;
    }
    uint8_t h[32];
    // UnspendableAccount::circuit: account_id == H(H(salt || secret)), unconditional
    if (qpgpu_leaf_unspendable_account(nullptr, 0, in->secret, h)) return fail(-1, "hashing failed");
    if (std::memcmp(h, in->unspendable_account, 32)) return fail(-4, "unspendable_account is not H(H(\"wormhole\" || secret))");
    // ZkMerkleProofData::circuit: fee relation with its range checks (all amounts are 32-bit by type)
    if (in->volume_fee_bps > 10000) return fail(-4, "volume_fee_bps exceeds 10000 (range check of 10000 - fee_bps)");
    const uint64_t lhs = ((uint64_t)in->output_amount_1 + in->output_amount_2) * 10000ull;
    const uint64_t rhs = (uint64_t)in->input_amount * (10000ull - in->volume_fee_bps);
    if (lhs > rhs) return fail(-4, "fee constraint violated: (output_1 + output_2) * 10000 > input * (10000 - fee_bps)");
    static const uint8_t zero32[32] = {0};
    const bool dummy = !std::memcmp(in->block_hash, zero32, 32) && in->output_amount_1 == 0 && in->output_amount_2 == 0;
    if (dummy) return 0;                    // the remaining bindings are multiplied by is_not_dummy
    if (qpgpu_leaf_nullifier(nullptr, 0, in->secret, in->transfer_count, h)) return fail(-1, "hashing failed");
    if (std::memcmp(h, in->nullifier, 32)) return fail(-4, "nullifier is not H(H(\"~nullif~\" || secret || transfer_count))");
    if (qpgpu_leaf_block_hash(nullptr, 0, in->parent_hash, in->block_number, in->state_root, in->extrinsics_root, in->zk_tree_root, in->digest, h))
        return fail(-1, "hashing failed");
    if (std::memcmp(h, in->block_hash, 32)) return fail(-4, "block_hash is not the hash of the header contents");
    // the Merkle path from the leaf (to_account = unspendable account) to the header's ZK tree root
    if (qpgpu_zk_leaf_hash(in->unspendable_account, in->transfer_count, in->asset_id, in->input_amount, h)) return fail(-1, "hashing failed");
    uint8_t four[128];
    for (uint32_t l = 0; l < in->zk_merkle_depth; l++) {
        if (qpgpu_zk_insert_at_position(h, &in->zk_merkle_siblings[l][0][0], in->zk_merkle_positions[l], four) || qpgpu_zk_hash_node_presorted(four, h))
            return fail(-1, "malformed Merkle level");
    }
    if (std::memcmp(h, in->zk_tree_root, 32)) return fail(-4, "ZK Merkle path does not lead to the header's zk_tree_root");
    return 0;
}

}  // extern "C"
