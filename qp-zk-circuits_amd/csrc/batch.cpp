// batch.cpp — host side of the two aggregation levels (include/qpgpu_batch.h): public-input parsers, the admission checks
// of PrivateBatchProver::commit / PublicBatchProver::commit, padding + shuffle + dummy-nullifier preimages, and the native
// mirror of what the two wrapper circuits write into their public inputs. References are cited per function in the header.
#include "../../include/qpgpu_batch.h"
#include "../../include/qpgpu_leaf.h"
#include <sys/random.h>
#include <algorithm>
#include <array>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace {

constexpr uint64_t P = 0xFFFFFFFF00000001ull;
constexpr int ERR_INVALID = -1, ERR_UNSAT = -4;

int fail(char *err, int code, const char *fmt, ...) {
    if (err) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(err, QPGPU_BATCH_ERR_CAP, fmt, ap);
        va_end(ap);
    }
    return code;
}

using Digest = std::array<uint64_t, 4>;
Digest digest_at(const uint64_t *p) { return {p[0], p[1], p[2], p[3]}; }
bool is_zero(const Digest &d) { return (d[0] | d[1] | d[2] | d[3]) == 0; }

// hash_u64s_to_bytes_digest: 8 little-endian bytes per felt, each chunk below the field order
bool digest_bytes(const uint64_t *v, uint8_t out[32], std::string &why) {
    for (int i = 0; i < 4; i++) {
        if (v[i] >= P) {
            why = "Chunk out of field range at index " + std::to_string(i) + ": " + std::to_string(v[i]);
            return false;
        }
        for (int k = 0; k < 8; k++) out[8 * i + k] = (uint8_t)(v[i] >> (8 * k));
    }
    return true;
}
std::string digest_debug(const uint8_t b[32]) {   // BytesDigest's Debug form
    char buf[80];
    size_t n = (size_t)snprintf(buf, sizeof buf, "BytesDigest(0x");
    for (int i = 0; i < 32; i++) n += (size_t)snprintf(buf + n, sizeof buf - n, "%02x", b[i]);
    snprintf(buf + n, sizeof buf - n, ")");
    return buf;
}
bool to_u32(uint64_t v, uint32_t &out) { out = (uint32_t)v; return v <= 0xFFFFFFFFull; }
bool all_canonical(const uint64_t *p, size_t n) {
    for (size_t i = 0; i < n; i++) if (p[i] >= P) return false;
    return true;
}
uint64_t add_mod(uint64_t a, uint64_t b) { unsigned __int128 s = (unsigned __int128)a + b; return (uint64_t)(s >= P ? s - P : s); }

// leaf public-input offsets (private_batch/circuit/constants.rs:19-27)
enum { L_ASSET = 0, L_OUT1 = 1, L_OUT2 = 2, L_FEE = 3, L_NULL = 4, L_EXIT1 = 8, L_EXIT2 = 12, L_BLOCK = 16, L_NUMBER = 20 };
// private-batch output offsets (aggregated_output)
enum { A_SLOTS = 0, A_ASSET = 1, A_FEE = 2, A_BLOCK = 3, A_NUMBER = 7 };

struct ChaCha20 {
    uint32_t key[8];
    uint64_t counter = 0;
    uint32_t block[16];
    int pos = 16;
    static uint32_t rotl(uint32_t v, int n) { return (v << n) | (v >> (32 - n)); }
    static void quarter(uint32_t *x, int a, int b, int c, int d) {
        x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 16);
        x[c] += x[d]; x[b] = rotl(x[b] ^ x[c], 12);
        x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 8);
        x[c] += x[d]; x[b] = rotl(x[b] ^ x[c], 7);
    }
    explicit ChaCha20(const uint8_t seed[32]) { std::memcpy(key, seed, 32); }
    ~ChaCha20() { volatile uint32_t *k = key; for (int i = 0; i < 8; i++) k[i] = 0; }
    void refill() {
        uint32_t in[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u};
        std::memcpy(in + 4, key, sizeof key);
        in[12] = (uint32_t)counter; in[13] = (uint32_t)(counter >> 32); in[14] = 0x42415443u; in[15] = 0x48u;   // stream "BATCH"
        uint32_t x[16];
        std::memcpy(x, in, sizeof x);
        for (int dr = 0; dr < 10; dr++) {
            quarter(x, 0, 4, 8, 12); quarter(x, 1, 5, 9, 13); quarter(x, 2, 6, 10, 14); quarter(x, 3, 7, 11, 15);
            quarter(x, 0, 5, 10, 15); quarter(x, 1, 6, 11, 12); quarter(x, 2, 7, 8, 13); quarter(x, 3, 4, 9, 14);
        }
        for (int i = 0; i < 16; i++) block[i] = x[i] + in[i];
        counter++; pos = 0;
    }
    uint32_t next32() { if (pos == 16) refill(); return block[pos++]; }
    uint64_t next64() { uint64_t lo = next32(); return ((uint64_t)next32() << 32) | lo; }
    uint64_t below(uint64_t range) {   // uniform in [0, range): widening multiply with rejection of the biased zone
        const uint64_t limit = (uint64_t)(-range) % range;
        for (;;) {
            unsigned __int128 m = (unsigned __int128)next64() * range;
            if ((uint64_t)m >= limit) return (uint64_t)(m >> 64);
        }
    }
};

}  // namespace

extern "C" {

int qpgpu_validate_proof_count(uint64_t count, const char *label, char *err) {
    if (count == 0) return fail(err, ERR_INVALID, "%s must be > 0", label);
    if (count > QPGPU_BATCH_MAX_PROOFS)
        return fail(err, ERR_INVALID, "%s (%llu) exceeds maximum allowed (%d)", label, (unsigned long long)count, QPGPU_BATCH_MAX_PROOFS);
    return 0;
}

size_t qpgpu_private_batch_pi_len(size_t n_leaf) { return QPGPU_LEAF_PI_LEN * n_leaf + 8; }

size_t qpgpu_public_batch_pi_len(size_t m, size_t n) {
    if (m == 0 || n == 0 || m > QPGPU_BATCH_MAX_PROOFS || n > QPGPU_BATCH_MAX_PROOFS) return 0;
    return QPGPU_PUBLIC_BATCH_HEADER_LEN + m * 2 * n * QPGPU_EXIT_SLOT_LEN + m * n * 4;
}

int qpgpu_leaf_public_inputs_parse(const uint64_t *pis, size_t n, qpgpu_leaf_public_inputs *out, char *err) {
    if (!pis || !out) return fail(err, ERR_INVALID, "null argument");
    if (n != QPGPU_LEAF_PI_LEN) return fail(err, ERR_INVALID, "public inputs should contain %d field elements, got %zu", QPGPU_LEAF_PI_LEN, n);
    if (!to_u32(pis[L_ASSET], out->asset_id)) return fail(err, ERR_INVALID, "failed to convert asset_id to u32");
    if (!to_u32(pis[L_OUT1], out->output_amount_1)) return fail(err, ERR_INVALID, "failed to convert output_amount_1 to u32");
    if (!to_u32(pis[L_OUT2], out->output_amount_2)) return fail(err, ERR_INVALID, "failed to convert output_amount_2 to u32");
    if (!to_u32(pis[L_FEE], out->volume_fee_bps)) return fail(err, ERR_INVALID, "failed to convert volume_fee_bps to u32");
    std::string why;
    if (!digest_bytes(pis + L_NULL, out->nullifier, why)) return fail(err, ERR_INVALID, "failed to parse nullifier: %s", why.c_str());
    if (!digest_bytes(pis + L_EXIT1, out->exit_account_1, why)) return fail(err, ERR_INVALID, "failed to parse exit_account_1: %s", why.c_str());
    if (!digest_bytes(pis + L_EXIT2, out->exit_account_2, why)) return fail(err, ERR_INVALID, "failed to parse exit_account_2: %s", why.c_str());
    if (!digest_bytes(pis + L_BLOCK, out->block_hash, why)) return fail(err, ERR_INVALID, "failed to parse block_hash: %s", why.c_str());
    if (!to_u32(pis[L_NUMBER], out->block_number)) return fail(err, ERR_INVALID, "failed to convert block_number to u32");
    return 0;
}

int qpgpu_private_batch_public_inputs_parse(const uint64_t *pis, size_t n, qpgpu_private_batch_public_inputs *out,
                                            qpgpu_exit_slot *slots, uint8_t *nullifiers, char *err) {
    if (!pis || !out) return fail(err, ERR_INVALID, "null argument");
    if (n < 8) return fail(err, ERR_INVALID, "AggregatedPI: too few elements, need at least 8 for header, got %zu", n);
    const size_t payload = n - 8;
    if (payload % QPGPU_LEAF_PI_LEN)
        return fail(err, ERR_INVALID, "AggregatedPI: malformed length %zu - expected 8 + N*%d felts for the padded aggregated layout", n, QPGPU_LEAF_PI_LEN);
    if (!to_u32(pis[0], out->num_exit_slots)) return fail(err, ERR_INVALID, "AggregatedPI: num_exit_slots at index 0 exceeds u32 range");
    if (!to_u32(pis[1], out->asset_id)) return fail(err, ERR_INVALID, "AggregatedPI: asset_id at index 1 exceeds u32 range");
    if (!to_u32(pis[2], out->volume_fee_bps)) return fail(err, ERR_INVALID, "AggregatedPI: volume_fee_bps at index 2 exceeds u32 range");
    const size_t n_leaf = payload / QPGPU_LEAF_PI_LEN;
    if (int rc = qpgpu_validate_proof_count(n_leaf, "AggregatedPI: n_leaf", err)) return rc;
    if (out->num_exit_slots != n_leaf * 2)
        return fail(err, ERR_INVALID,
                    "AggregatedPI: num_exit_slots at index 0 is %u, but the layout implies %zu exit slots (%zu leaves); these are not "
                    "private-batch aggregation PIs", out->num_exit_slots, n_leaf * 2, n_leaf);
    std::string why;
    if (!digest_bytes(pis + 3, out->block_hash, why)) return fail(err, ERR_INVALID, "AggregatedPI: parsing block_hash from indices 3..7: %s", why.c_str());
    if (!to_u32(pis[7], out->block_number)) return fail(err, ERR_INVALID, "AggregatedPI: parsing block_number from index 7");
    out->n_leaf = (uint32_t)n_leaf;
    size_t cursor = 8;
    for (size_t i = 0; i < 2 * n_leaf; i++) {
        uint32_t sum;
        if (!to_u32(pis[cursor], sum)) return fail(err, ERR_INVALID, "AggregatedPI: summed_output_amount at cursor %zu exceeds u32 range", cursor);
        cursor++;
        uint8_t acct[32];
        if (!digest_bytes(pis + cursor, acct, why))
            return fail(err, ERR_INVALID, "AggregatedPI: parsing exit_account[%zu] at cursor %zu: %s", i, cursor, why.c_str());
        cursor += 4;
        if (slots) { slots[i].summed_output_amount = sum; std::memcpy(slots[i].exit_account, acct, 32); }
    }
    for (size_t i = 0; i < n_leaf; i++) {
        uint8_t nf[32];
        if (!digest_bytes(pis + cursor, nf, why)) return fail(err, ERR_INVALID, "AggregatedPI: parsing nullifier[%zu] at cursor %zu: %s", i, cursor, why.c_str());
        cursor += 4;
        if (nullifiers) std::memcpy(nullifiers + 32 * i, nf, 32);
    }
    return 0;
}

int qpgpu_public_batch_public_inputs_parse(const uint64_t *pis, size_t n, uint64_t m, uint64_t nl, qpgpu_public_batch_public_inputs *out,
                                           qpgpu_exit_slot *slots, uint8_t *nullifiers, char *err) {
    if (!pis || !out) return fail(err, ERR_INVALID, "null argument");
    if (int rc = qpgpu_validate_proof_count(m, "num_private_batch_proofs", err)) return rc;
    if (int rc = qpgpu_validate_proof_count(nl, "num_leaf_proofs", err)) return rc;
    const size_t expected = qpgpu_public_batch_pi_len((size_t)m, (size_t)nl);
    if (n != expected)
        return fail(err, ERR_INVALID, "PublicBatchPI: expected %zu felts (n_inner=%llu, n_leaves=%llu), got %zu", expected,
                    (unsigned long long)m, (unsigned long long)nl, n);
    const size_t total_slots = (size_t)m * 2 * (size_t)nl, total_nulls = (size_t)m * (size_t)nl;
    std::string why;
    if (!digest_bytes(pis, out->aggregator_address, why)) return fail(err, ERR_INVALID, "PublicBatchPI: parsing aggregator_address: %s", why.c_str());
    if (!to_u32(pis[4], out->asset_id)) return fail(err, ERR_INVALID, "PublicBatchPI: asset_id exceeds u32 range");
    if (!to_u32(pis[5], out->volume_fee_bps)) return fail(err, ERR_INVALID, "PublicBatchPI: volume_fee_bps exceeds u32 range");
    if (!digest_bytes(pis + 6, out->block_hash, why)) return fail(err, ERR_INVALID, "PublicBatchPI: parsing block_hash: %s", why.c_str());
    if (!to_u32(pis[10], out->block_number)) return fail(err, ERR_INVALID, "PublicBatchPI: block_number exceeds u32 range");
    if (!to_u32(pis[11], out->total_exit_slots)) return fail(err, ERR_INVALID, "PublicBatchPI: total_exit_slots exceeds u32 range");
    if (out->total_exit_slots != total_slots)
        return fail(err, ERR_INVALID, "PublicBatchPI: total_exit_slots %u != expected %zu", out->total_exit_slots, total_slots);
    size_t cursor = QPGPU_PUBLIC_BATCH_HEADER_LEN;
    for (size_t i = 0; i < total_slots; i++) {
        uint32_t sum;
        if (!to_u32(pis[cursor], sum)) return fail(err, ERR_INVALID, "PublicBatchPI: exit slot %zu sum exceeds u32", i);
        cursor++;
        uint8_t acct[32];
        if (!digest_bytes(pis + cursor, acct, why)) return fail(err, ERR_INVALID, "PublicBatchPI: parsing exit slot %zu account: %s", i, why.c_str());
        cursor += 4;
        if (slots) { slots[i].summed_output_amount = sum; std::memcpy(slots[i].exit_account, acct, 32); }
    }
    for (size_t i = 0; i < total_nulls; i++) {
        uint8_t nf[32];
        if (!digest_bytes(pis + cursor, nf, why)) return fail(err, ERR_INVALID, "PublicBatchPI: parsing nullifier %zu: %s", i, why.c_str());
        cursor += 4;
        if (nullifiers) std::memcpy(nullifiers + 32 * i, nf, 32);
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------ private batch

int qpgpu_private_batch_preflight(const uint64_t *leaf_pis, size_t count, size_t num_leaf_proofs, char *err) {
    if (count == 0) return fail(err, ERR_INVALID, "no leaf proofs to aggregate");
    if (count > num_leaf_proofs) return fail(err, ERR_INVALID, "too many proofs: got %zu, expected at most %zu", count, num_leaf_proofs);
    if (!leaf_pis) return fail(err, ERR_INVALID, "null argument");
    if (!all_canonical(leaf_pis, count * QPGPU_LEAF_PI_LEN)) return fail(err, ERR_INVALID, "leaf public inputs must be canonical field elements");
    const bool padding = count < num_leaf_proofs;
    auto row = [&](size_t i) { return leaf_pis + i * QPGPU_LEAF_PI_LEN; };
    for (size_t i = 0; i < count; i++) {
        if (!padding) break;
        const uint64_t asset = row(i)[L_ASSET];
        if (asset > 0xFFFFFFFFull) return fail(err, ERR_INVALID, "leaf proof %zu: leaf proof asset_id exceeds u32 range", i);
        if (asset != 0)
            return fail(err, ERR_INVALID,
                        "real proof %zu has asset_id=%llu, but dummy proofs use asset_id=0. All proofs must have the same asset_id for "
                        "aggregation when padding is required.", i, (unsigned long long)asset);
    }
    // ensure_leaf_batch_compatible
    for (size_t i = 1; i < count; i++)
        if (row(i)[L_ASSET] != row(0)[L_ASSET])
            return fail(err, ERR_INVALID,
                        "leaf proof %zu has asset_id=%llu, but proof 0 has asset_id=%llu; the private-batch circuit enforces asset "
                        "consistency across all slots", i, (unsigned long long)row(i)[L_ASSET], (unsigned long long)row(0)[L_ASSET]);
    long ref = -1;
    std::map<Digest, size_t> seen;
    for (size_t i = 0; i < count; i++) {
        const Digest block = digest_at(row(i) + L_BLOCK);
        if (is_zero(block)) continue;   // dummy sentinel: exempt from block / fee / nullifier consistency
        if (ref < 0) ref = (long)i;
        else {
            if (block != digest_at(row((size_t)ref) + L_BLOCK))
                return fail(err, ERR_INVALID,
                            "leaf proof %zu is for a different block than proof %ld; all non-dummy proofs in a private batch must share one "
                            "block hash", i, ref);
            if (row(i)[L_FEE] != row((size_t)ref)[L_FEE])
                return fail(err, ERR_INVALID,
                            "leaf proof %zu has volume_fee_bps=%llu, but proof %ld has volume_fee_bps=%llu; all non-dummy proofs in a private "
                            "batch must share one fee rate", i, (unsigned long long)row(i)[L_FEE], ref, (unsigned long long)row((size_t)ref)[L_FEE]);
        }
        auto ins = seen.emplace(digest_at(row(i) + L_NULL), i);
        if (!ins.second)
            return fail(err, ERR_INVALID,
                        "leaf proof %zu carries the same nullifier as proof %zu; the private-batch circuit enforces pairwise-distinct real "
                        "nullifiers, so this batch (e.g. the same leaf proof supplied twice) would only fail after the expensive recursive "
                        "proving run", i, ins.first->second);
    }
    if (ref < 0)
        return fail(err, ERR_INVALID,
                    "every supplied leaf proof is all-dummy (block_hash == 0): such a batch settles nothing; supply at least one real leaf proof");
    return 0;
}

int qpgpu_dummy_leaf_template_check(const uint64_t *pis, size_t n, char *err) {
    qpgpu_leaf_public_inputs p;
    char inner[QPGPU_BATCH_ERR_CAP];
    if (qpgpu_leaf_public_inputs_parse(pis, n, &p, inner))
        return fail(err, ERR_INVALID, "failed to parse dummy leaf proof template public inputs: %.300s", inner);
    static const uint8_t zero[32] = {0};
    if (std::memcmp(p.block_hash, zero, 32))
        return fail(err, ERR_INVALID, "dummy leaf proof template has non-zero block_hash %s; padding templates must carry the all-zero block-hash sentinel",
                    digest_debug(p.block_hash).c_str());
    if (p.output_amount_1 || p.output_amount_2)
        return fail(err, ERR_INVALID, "dummy leaf proof template has non-zero output amounts (%u, %u); padding templates must contribute zero to every exit slot",
                    p.output_amount_1, p.output_amount_2);
    if (p.asset_id)
        return fail(err, ERR_INVALID,
                    "dummy leaf proof template has non-zero asset_id %u; padding templates must use the native asset (asset_id = 0) because the "
                    "circuit enforces asset_id equality across all slots, dummies included", p.asset_id);
    if (std::memcmp(p.exit_account_1, zero, 32) || std::memcmp(p.exit_account_2, zero, 32))
        return fail(err, ERR_INVALID,
                    "dummy leaf proof template has non-zero exit account(s); padding templates must use the canonical all-zero exit account");
    return 0;
}

int qpgpu_private_batch_arrange(size_t count, size_t num_leaf_proofs, const uint8_t *seed32, uint32_t *slot_source, uint64_t *preimages, char *err) {
    if (!slot_source || !preimages) return fail(err, ERR_INVALID, "null argument");
    if (int rc = qpgpu_validate_proof_count(num_leaf_proofs, "num_leaf_proofs", err)) return rc;
    if (count == 0) return fail(err, ERR_INVALID, "no leaf proofs to aggregate");
    if (count > num_leaf_proofs) return fail(err, ERR_INVALID, "too many proofs: got %zu, expected at most %zu", count, num_leaf_proofs);
    uint8_t key[32];
    if (seed32) std::memcpy(key, seed32, 32);
    else {
        size_t got = 0;
        while (got < 32) {
            const ssize_t r = getrandom(key + got, 32 - got, 0);
            if (r <= 0) return fail(err, ERR_INVALID, "operating-system entropy source unavailable");
            got += (size_t)r;
        }
    }
    ChaCha20 rng(key);
    volatile uint8_t *k = key;
    for (int i = 0; i < 32; i++) k[i] = 0;
    for (size_t s = 0; s < num_leaf_proofs; s++) slot_source[s] = s < count ? (uint32_t)s : UINT32_MAX;
    // uniform shuffle (Fisher-Yates, as rand's SliceRandom::shuffle): hides which slots are padding
    for (size_t i = num_leaf_proofs; i-- > 1;) std::swap(slot_source[i], slot_source[rng.below(i + 1)]);
    // one dummy-nullifier preimage per slot: 32 random bytes whose four 8-byte chunks are below the field order
    for (size_t s = 0; s < num_leaf_proofs * 4; s++) {
        uint64_t v;
        do v = rng.next64(); while (v >= P);
        preimages[s] = v;
    }
    return 0;
}

int qpgpu_random_field_elements(const uint8_t *seed32, uint64_t *out, size_t n, char *err) {
    if (!out && n) return fail(err, ERR_INVALID, "null argument");
    uint8_t key[32];
    if (seed32) std::memcpy(key, seed32, 32);
    else {
        size_t got = 0;
        while (got < 32) {
            const ssize_t r = getrandom(key + got, 32 - got, 0);
            if (r <= 0) return fail(err, ERR_INVALID, "operating-system entropy source unavailable");
            got += (size_t)r;
        }
    }
    ChaCha20 rng(key);
    volatile uint8_t *k = key;
    for (int i = 0; i < 32; i++) k[i] = 0;
    for (size_t i = 0; i < n; i++) {
        uint64_t v;
        do v = rng.next64(); while (v >= P);
        out[i] = v;
    }
    return 0;
}

int qpgpu_private_batch_outputs(const uint64_t *leaf_pis, size_t n_leaf, const uint64_t *dummy_preimages, uint64_t *out, char *err) {
    if (!leaf_pis || !dummy_preimages || !out) return fail(err, ERR_INVALID, "null argument");
    if (int rc = qpgpu_validate_proof_count(n_leaf, "n_leaf", err)) return rc;
    if (!all_canonical(leaf_pis, n_leaf * QPGPU_LEAF_PI_LEN) || !all_canonical(dummy_preimages, n_leaf * 4))
        return fail(err, ERR_INVALID, "inputs must be canonical field elements");
    auto row = [&](size_t i) { return leaf_pis + i * QPGPU_LEAF_PI_LEN; };
    std::vector<bool> dummy(n_leaf);
    long ref = -1;
    for (size_t i = 0; i < n_leaf; i++) {
        dummy[i] = is_zero(digest_at(row(i) + L_BLOCK));
        if (!dummy[i] && ref < 0) ref = (long)i;   // first non-dummy slot: block hash, block number and fee reference
    }
    const uint64_t asset_ref = row(0)[L_ASSET];
    Digest block_ref = {0, 0, 0, 0};
    uint64_t number_ref = 0, fee_ref = 0;
    if (ref >= 0) { block_ref = digest_at(row((size_t)ref) + L_BLOCK); number_ref = row((size_t)ref)[L_NUMBER]; fee_ref = row((size_t)ref)[L_FEE]; }
    for (size_t i = 0; i < n_leaf; i++) {
        if (!dummy[i] && digest_at(row(i) + L_BLOCK) != block_ref)
            return fail(err, ERR_UNSAT, "slot %zu: block hash differs from the reference slot %ld (is_dummy OR block == block_ref)", i, ref);
        if (row(i)[L_ASSET] != asset_ref) return fail(err, ERR_UNSAT, "slot %zu: asset_id differs from slot 0 (asset_id equality holds for every slot, dummies included)", i);
        if (!dummy[i] && row(i)[L_FEE] != fee_ref) return fail(err, ERR_UNSAT, "slot %zu: volume_fee_bps differs from the reference slot %ld", i, ref);
    }
    size_t w = 0;
    out[w++] = 2 * n_leaf;
    out[w++] = asset_ref;
    out[w++] = fee_ref;
    for (int j = 0; j < 4; j++) out[w++] = block_ref[(size_t)j];
    out[w++] = number_ref;
    // exit slots: dummy slots masked to (zero account, 0) before the grouping; a slot carries the sum over every slot with
    // the same account unless the account appeared in an earlier slot, in which case it is zeroed
    const size_t ns = 2 * n_leaf;
    std::vector<Digest> exits(ns);
    std::vector<uint64_t> amounts(ns);
    for (size_t s = 0; s < ns; s++) {
        const size_t i = s / 2;
        exits[s] = dummy[i] ? Digest{0, 0, 0, 0} : digest_at(row(i) + (s % 2 ? L_EXIT2 : L_EXIT1));
        amounts[s] = dummy[i] ? 0 : row(i)[s % 2 ? L_OUT2 : L_OUT1];
    }
    for (size_t s = 0; s < ns; s++) {
        bool dup = false;
        for (size_t e = 0; e < s; e++) dup = dup || exits[e] == exits[s];
        uint64_t acc = 0;
        for (size_t t = 0; t < ns; t++) if (exits[t] == exits[s]) acc = add_mod(acc, amounts[t]);
        const uint64_t sum = dup ? 0 : acc;
        if (sum > 0xFFFFFFFFull) return fail(err, ERR_UNSAT, "exit slot %zu: summed output amount %llu fails the 32-bit range check", s, (unsigned long long)sum);
        out[w++] = sum;
        for (int j = 0; j < 4; j++) out[w++] = dup ? 0 : exits[s][(size_t)j];
    }
    // real nullifiers pairwise distinct
    for (size_t i = 0; i < n_leaf; i++)
        for (size_t j = i + 1; j < n_leaf; j++)
            if (!dummy[i] && !dummy[j] && digest_at(row(i) + L_NULL) == digest_at(row(j) + L_NULL))
                return fail(err, ERR_UNSAT, "slots %zu and %zu carry the same real nullifier", i, j);
    // dummy slots: H(H(preimage)) under Poseidon2 (hash_n_to_hash_no_pad_p2 twice); the region is emitted sorted
    std::vector<Digest> selected(n_leaf);
    for (size_t i = 0; i < n_leaf; i++) {
        if (!dummy[i]) { selected[i] = digest_at(row(i) + L_NULL); continue; }
        uint64_t inner[4], outer[4];
        if (qpgpu_poseidon2_hash_pad10(nullptr, 0, dummy_preimages + 4 * i, 4, inner) || qpgpu_poseidon2_hash_pad10(nullptr, 0, inner, 4, outer))
            return fail(err, ERR_INVALID, "Poseidon2 hash of a dummy nullifier preimage failed");
        selected[i] = digest_at(outer);
    }
    std::sort(selected.begin(), selected.end());   // ascending lexicographic over canonical limbs, limb 0 most significant
    for (const Digest &d : selected) for (int j = 0; j < 4; j++) out[w++] = d[(size_t)j];
    const size_t total = qpgpu_private_batch_pi_len(n_leaf);
    while (w < total) out[w++] = 0;
    return 0;
}

// ------------------------------------------------------------------------------------------------- public batch

int qpgpu_public_batch_preflight(const uint64_t *inner_pis, size_t count, size_t pi_len, size_t num_private_batch_proofs, char *err) {
    if (count == 0) return fail(err, ERR_INVALID, "no private-batch proofs to aggregate");
    if (count > num_private_batch_proofs)
        return fail(err, ERR_INVALID, "Expected at most %zu private-batch proofs, but got %zu", num_private_batch_proofs, count);
    if (!inner_pis) return fail(err, ERR_INVALID, "null argument");
    if (pi_len < 8 || (pi_len - 8) % QPGPU_LEAF_PI_LEN || (pi_len - 8) / QPGPU_LEAF_PI_LEN == 0 || (pi_len - 8) / QPGPU_LEAF_PI_LEN > QPGPU_BATCH_MAX_PROOFS)
        return fail(err, ERR_INVALID, "private-batch proof 0 is malformed: private-batch proof public input length mismatch: expected 8 + N*%d, got %zu",
                    QPGPU_LEAF_PI_LEN, pi_len);
    if (!all_canonical(inner_pis, count * pi_len)) return fail(err, ERR_INVALID, "private-batch public inputs must be canonical field elements");
    auto row = [&](size_t i) { return inner_pis + i * pi_len; };
    long ref = -1;
    for (size_t i = 0; i < count; i++) {
        const Digest block = digest_at(row(i) + A_BLOCK);
        if (is_zero(block)) continue;
        if (ref < 0) { ref = (long)i; continue; }
        const uint64_t *r = row((size_t)ref);
        if (block != digest_at(r + A_BLOCK))
            return fail(err, ERR_INVALID,
                        "private-batch proof %zu is for a different block than proof %ld; all non-dummy proofs in a public batch must share one "
                        "block hash", i, ref);
        if (row(i)[A_ASSET] != r[A_ASSET])
            return fail(err, ERR_INVALID,
                        "private-batch proof %zu has asset_id=%llu, but proof %ld has asset_id=%llu; all non-dummy proofs in a public batch must "
                        "share one asset", i, (unsigned long long)row(i)[A_ASSET], ref, (unsigned long long)r[A_ASSET]);
        if (row(i)[A_FEE] != r[A_FEE])
            return fail(err, ERR_INVALID,
                        "private-batch proof %zu has volume_fee_bps=%llu, but proof %ld has volume_fee_bps=%llu; all non-dummy proofs in a public "
                        "batch must share one fee rate", i, (unsigned long long)row(i)[A_FEE], ref, (unsigned long long)r[A_FEE]);
    }
    if (ref < 0)
        return fail(err, ERR_INVALID,
                    "every supplied private-batch proof is all-dummy (block_hash == 0): such a batch settles nothing on-chain; supply at least one "
                    "real private-batch proof");
    return 0;
}

int qpgpu_dummy_private_batch_template_check(const uint64_t *pis, size_t n, char *err) {
    qpgpu_private_batch_public_inputs h;
    std::vector<qpgpu_exit_slot> slots(2 * QPGPU_BATCH_MAX_PROOFS);
    char inner[QPGPU_BATCH_ERR_CAP];
    if (qpgpu_private_batch_public_inputs_parse(pis, n, &h, slots.data(), nullptr, inner))
        return fail(err, ERR_INVALID, "failed to parse dummy private-batch proof public inputs: %.300s", inner);
    static const uint8_t zero[32] = {0};
    if (std::memcmp(h.block_hash, zero, 32))
        return fail(err, ERR_INVALID, "dummy private-batch proof template has non-zero block_hash %s; padding templates must carry the all-zero block-hash sentinel",
                    digest_debug(h.block_hash).c_str());
    for (size_t i = 0; i < 2 * (size_t)h.n_leaf; i++) {
        if (slots[i].summed_output_amount)
            return fail(err, ERR_INVALID,
                        "dummy private-batch proof template forwards non-zero payout at slot %zu (%u); padding templates must contribute zero to "
                        "every exit slot", i, slots[i].summed_output_amount);
        if (std::memcmp(slots[i].exit_account, zero, 32))
            return fail(err, ERR_INVALID,
                        "dummy private-batch proof template has non-zero exit account at slot %zu; padding templates must carry the canonical "
                        "all-zero exit account so padded slots stay indistinguishable from unused ones", i);
    }
    return 0;
}

int qpgpu_public_batch_outputs(const uint64_t *inner_pis, size_t m, size_t n_leaf, const uint8_t aggregator_address[32], uint64_t *out, char *err) {
    if (!inner_pis || !aggregator_address || !out) return fail(err, ERR_INVALID, "null argument");
    if (int rc = qpgpu_validate_proof_count(m, "num_private_batch_proofs", err)) return rc;
    if (int rc = qpgpu_validate_proof_count(n_leaf, "num_leaf_proofs", err)) return rc;
    const size_t pi_len = qpgpu_private_batch_pi_len(n_leaf);
    if (!all_canonical(inner_pis, m * pi_len)) return fail(err, ERR_INVALID, "inputs must be canonical field elements");
    uint64_t addr[4];
    for (int i = 0; i < 4; i++) {
        addr[i] = 0;
        for (int k = 0; k < 8; k++) addr[i] |= (uint64_t)aggregator_address[8 * i + k] << (8 * k);
        if (addr[i] >= P) return fail(err, ERR_INVALID, "aggregator address: Chunk out of field range at index %d: %llu", i, (unsigned long long)addr[i]);
    }
    auto row = [&](size_t i) { return inner_pis + i * pi_len; };
    std::vector<bool> dummy(m);
    long ref = -1;
    for (size_t i = 0; i < m; i++) {
        dummy[i] = is_zero(digest_at(row(i) + A_BLOCK));
        if (!dummy[i] && ref < 0) ref = (long)i;
    }
    Digest block_ref = {0, 0, 0, 0};
    uint64_t number_ref = 0, asset_ref = 0, fee_ref = 0;
    if (ref >= 0) {
        const uint64_t *r = row((size_t)ref);
        block_ref = digest_at(r + A_BLOCK); number_ref = r[A_NUMBER]; asset_ref = r[A_ASSET]; fee_ref = r[A_FEE];
    }
    for (size_t i = 0; i < m; i++) {
        if (dummy[i]) continue;
        if (row(i)[A_ASSET] != asset_ref) return fail(err, ERR_UNSAT, "inner proof %zu: asset_id differs from the reference proof %ld", i, ref);
        if (row(i)[A_FEE] != fee_ref) return fail(err, ERR_UNSAT, "inner proof %zu: volume_fee_bps differs from the reference proof %ld", i, ref);
        if (digest_at(row(i) + A_BLOCK) != block_ref) return fail(err, ERR_UNSAT, "inner proof %zu: block hash differs from the reference proof %ld", i, ref);
    }
    size_t w = 0;
    for (int j = 0; j < 4; j++) out[w++] = addr[j];
    out[w++] = asset_ref;
    out[w++] = fee_ref;
    for (int j = 0; j < 4; j++) out[w++] = block_ref[(size_t)j];
    out[w++] = number_ref;
    out[w++] = m * 2 * n_leaf;
    const size_t slots_start = QPGPU_BATCH_HEADER_LEN, nulls_start = QPGPU_BATCH_HEADER_LEN + 2 * n_leaf * QPGPU_EXIT_SLOT_LEN;
    for (size_t i = 0; i < m; i++)      // exit slots forwarded in order, a dummy inner's zeroed
        for (size_t k = 0; k < 2 * n_leaf * QPGPU_EXIT_SLOT_LEN; k++) out[w++] = dummy[i] ? 0 : row(i)[slots_start + k];
    for (size_t i = 0; i < m; i++)      // nullifiers likewise
        for (size_t k = 0; k < n_leaf * 4; k++) out[w++] = dummy[i] ? 0 : row(i)[nulls_start + k];
    return 0;
}

}  // extern "C"
