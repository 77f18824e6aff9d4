// Sanitizer driver for the circuit builder and the circuits restated on it (built by tests/soak/sanitize_host.py with
// g++ -fsanitize=address,undefined): builds every leaf fragment and the fake leaf, commits inputs, builds wrapper circuits with
// every flag combination over a valid inner pack, fills their proof targets from a valid proof and from mutated ones, and hands
// the wrapper builder mutated / truncated / random inner packs and caps. The functions must only ever return error codes.
// usage: builder_driver <fake_leaf_proof.bin> [iterations]
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#include "../../include/qpgpu.h"
#include "../../include/qpgpu_batch.h"
#include "../../include/qpgpu_leaf.h"
#include "../../include/qpgpu_verify.h"
#include "../../include/qpgpu_wire.h"

static std::vector<uint8_t> slurp(const char *path) {
    FILE *f = fopen(path, "rb");
    if (!f) { perror(path); exit(2); }
    std::vector<uint8_t> b;
    uint8_t buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) b.insert(b.end(), buf, buf + n);
    fclose(f);
    return b;
}

static bool build_leaf(unsigned fragment, unsigned min_bits, int hasher, std::vector<uint64_t> &pack, std::vector<uint64_t> &map, char *err) {
    size_t words = 0;
    if (qpgpu_leaf_circuit_build(fragment, min_bits, hasher, nullptr, nullptr, 0, &words, nullptr, nullptr, err)) return false;
    pack.assign(words, 0); map.assign(QPGPU_LT_COUNT, 0);
    uint64_t info[QPGPU_LEAF_CIRCUIT_INFO_WORDS];
    return qpgpu_leaf_circuit_build(fragment, min_bits, hasher, nullptr, pack.data(), words, &words, map.data(), info, err) == 0;
}

static int build_wrapper(const std::vector<uint64_t> &inner, const std::vector<uint64_t> &cap, unsigned n, unsigned routed, unsigned flags, std::vector<uint64_t> &pack,
                         std::vector<uint64_t> &map, char *err) {
    size_t words = 0, mc = 0;
    int rc = qpgpu_wrapper_circuit_build(inner.data(), inner.size(), cap.data(), cap.size(), n, routed, 0, 0, flags, nullptr, 0, &words, nullptr, 0, &mc, nullptr, err);
    if (rc) return rc;
    pack.assign(words, 0); map.assign(mc, 0);
    uint64_t info[QPGPU_WRAPPER_CIRCUIT_INFO_WORDS];
    return qpgpu_wrapper_circuit_build(inner.data(), inner.size(), cap.data(), cap.size(), n, routed, 0, 0, flags, pack.data(), words, &words, map.data(), mc, &mc, info, err);
}

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: builder_driver fake_leaf_proof.bin [iterations]\n"); return 2; }
    const std::vector<uint8_t> proof = slurp(argv[1]);
    const int iters = argc > 2 ? atoi(argv[2]) : 60;
    std::mt19937_64 rng(4242);
    char err[QPGPU_BATCH_ERR_CAP];
    long built = 0, refused = 0, filled = 0, fill_refused = 0;

    // ---- every leaf fragment, both inner hashers, padded and not; commit of the dummy inputs against each target map ----
    std::vector<uint64_t> pack, map, fake, fake_map;
    for (unsigned fragment = 0; fragment <= 5; fragment++)
        for (int hasher = 0; hasher < 2; hasher++)
            for (unsigned min_bits : {0u, 9u}) {
                if (!build_leaf(fragment, min_bits, hasher, pack, map, err)) { if (fragment <= QPGPU_LEAF_FRAGMENT_FAKE_LEAF) { fprintf(stderr, "fragment %u: %s\n", fragment, err); return 1; } refused++; continue; }
                built++;
                if (qpgpu_pack_validate(pack.data(), pack.size(), err)) { fprintf(stderr, "built pack does not validate: %s\n", err); return 1; }
                qpgpu_leaf_inputs in;
                memset(&in, 0, sizeof in);
                in.volume_fee_bps = 10; in.transfer_count = 4; in.input_amount = 100;
                for (int i = 0; i < 32; i++) in.secret[i] = (uint8_t)(i % 8 == 7 ? 0 : 3 * i + 1);
                qpgpu_leaf_unspendable_account(nullptr, 0, in.secret, in.unspendable_account);
                uint64_t cells[QPGPU_LT_COUNT], values[QPGPU_LT_COUNT], pis[QPGPU_LEAF_PUBLIC_INPUTS];
                size_t count = 0;
                if (qpgpu_leaf_commit(&in, map.data(), cells, values, QPGPU_LT_COUNT, &count, pis, err)) { fprintf(stderr, "commit: %s\n", err); return 1; }
                // inputs the front-end refuses: a depth above 16, a position above 3, a non-canonical limb
                qpgpu_leaf_inputs bad = in; bad.zk_merkle_depth = 17;
                if (qpgpu_leaf_commit(&bad, map.data(), cells, values, QPGPU_LT_COUNT, &count, pis, err) == 0) { fprintf(stderr, "depth 17 accepted\n"); return 1; }
                bad = in; memset(bad.nullifier, 0xFF, 8);
                if (qpgpu_leaf_commit(&bad, map.data(), cells, values, QPGPU_LT_COUNT, &count, pis, err) == 0) { fprintf(stderr, "non-canonical nullifier accepted\n"); return 1; }
                if (qpgpu_leaf_commit(&in, map.data(), cells, values, 10, &count, pis, err) == 0) { fprintf(stderr, "short buffer accepted\n"); return 1; }
                if (fragment == QPGPU_LEAF_FRAGMENT_FULL) {     // the hash hints: cells of the circuit, values of the inputs, short buffers, refused inputs
                    std::vector<uint64_t> hc(QPGPU_LEAF_HASH_HINTS), hv(QPGPU_LEAF_HASH_HINTS);
                    size_t hn = 0;
                    if (qpgpu_leaf_circuit_hash_hint_cells(min_bits, hasher, nullptr, hc.data(), hc.size(), &hn, err) || hn != QPGPU_LEAF_HASH_HINTS) { fprintf(stderr, "hint cells: %s\n", err); return 1; }
                    if (qpgpu_leaf_hash_hints(&in, hv.data(), hv.size(), &hn, err) || hn != QPGPU_LEAF_HASH_HINTS) { fprintf(stderr, "hints: %s\n", err); return 1; }
                    if (qpgpu_leaf_circuit_hash_hint_cells(min_bits, hasher, nullptr, hc.data(), 10, &hn, err) == 0 || qpgpu_leaf_hash_hints(&in, hv.data(), 10, &hn, err) == 0) { fprintf(stderr, "short hint buffer accepted\n"); return 1; }
                    qpgpu_leaf_inputs deep = in; deep.zk_merkle_depth = 16;
                    for (int l = 0; l < 16; l++) { deep.zk_merkle_positions[l] = (uint8_t)(l % 4); for (int k = 0; k < 3; k++) for (int i = 0; i < 32; i++) deep.zk_merkle_siblings[l][k][i] = (uint8_t)(i % 8 == 7 ? 0 : l + k + i); }
                    if (qpgpu_leaf_hash_hints(&deep, hv.data(), hv.size(), &hn, err)) { fprintf(stderr, "hints (depth 16): %s\n", err); return 1; }
                    if (qpgpu_leaf_hash_hints(&bad, hv.data(), hv.size(), &hn, err) == 0) { fprintf(stderr, "hints of refused inputs accepted\n"); return 1; }
                }
            }
    if (!build_leaf(QPGPU_LEAF_FRAGMENT_FAKE_LEAF, 0, 0, fake, fake_map, err)) { fprintf(stderr, "fake leaf: %s\n", err); return 1; }
    std::vector<uint64_t> cap((size_t)4 << 4);
    for (auto &c : cap) c = rng() % 0xFFFFFFFF00000001ull;

    // ---- wrapper circuits over the fake leaf: every flag combination, several proof counts and wire budgets ----
    std::vector<uint64_t> w, wmap, keep, keep_map;
    for (unsigned flags = 0; flags < 32; flags++)
        for (unsigned n : {1u, 3u})
            for (unsigned routed : {0u, 60u}) {
                const int rc = build_wrapper(fake, cap, n, routed, flags, w, wmap, err);
                const bool bad_combo = ((flags & 2) && (flags & 4)) || ((flags & 8) && !(flags & 1)) || (flags & 4);     // public batch over a leaf-shaped circuit is refused
                if ((rc == 0) == bad_combo) { fprintf(stderr, "flags %u n %u routed %u: rc %d (%s)\n", flags, n, routed, rc, err); return 1; }
                if (rc == 0) { built++; if (qpgpu_pack_validate(w.data(), w.size(), err)) { fprintf(stderr, "wrapper pack does not validate: %s\n", err); return 1; } if (flags == 11 && n == 3 && routed == 0) { keep = w; keep_map = wmap; } }
                else refused++;
            }
    // a public-batch circuit over the private-batch one
    {
        std::vector<uint64_t> pub, pub_map;
        if (build_wrapper(keep, cap, 2, 0, 1 | 4 | 8, pub, pub_map, err)) { fprintf(stderr, "public batch: %s\n", err); return 1; }
        built++;
    }
    // ---- proof targets: the valid proof, then mutated / truncated / random ones ----
    const size_t T = qpgpu_proof_target_count(fake.data(), fake.size());
    std::vector<uint32_t> ids(3 * (T + 4));
    std::vector<uint64_t> vals(3 * (T + 4)), cells(keep_map.size()), cvals(keep_map.size()), pre(12, 7);
    for (int i = 0; i < iters * 4; i++) {
        std::vector<uint8_t> p = proof;
        const int kind = i % 5;
        if (kind == 1) p[rng() % p.size()] ^= (uint8_t)(1u << (rng() % 8));
        else if (kind == 2) { const size_t at = rng() % p.size(); for (int k = 0; k < 8 && at + k < p.size(); k++) p[at + k] = 0xFF; }
        else if (kind == 3) p.resize(rng() % p.size());
        else if (kind == 4) for (auto &b : p) b = (uint8_t)rng();
        const uint8_t *ps[3] = {proof.data(), p.data(), proof.data()};
        const size_t lens[3] = {proof.size(), p.size(), proof.size()};
        size_t cnt = 0;
        const int rc = qpgpu_batch_fill_proof_targets(fake.data(), fake.size(), ps, lens, 3, 3, pre.data(), 3, 3, "leaf proof", ids.data(), vals.data(), ids.size(), &cnt, err);
        if (rc == 0) { filled++; (void)qpgpu_leaf_map_targets(ids.data(), vals.data(), cnt, keep_map.data(), keep_map.size(), cells.data(), cvals.data()); }
        else fill_refused++;
        if (kind == 0 && rc) { fprintf(stderr, "valid proof refused: %s\n", err); return 1; }
    }
    // ---- the wrapper builder on mutated / truncated / random inner packs and caps of the wrong size ----
    for (int i = 0; i < iters; i++) {
        std::vector<uint64_t> q = fake;
        const int kind = i % 4;
        if (kind == 0) q[rng() % std::min<size_t>(q.size(), 48)] = rng() >> (rng() % 64);
        else if (kind == 1) q.resize(rng() % q.size());
        else if (kind == 2) q[rng() % q.size()] ^= 1ull << (rng() % 64);
        else for (auto &x : q) x = rng();
        std::vector<uint64_t> c2 = cap;
        if (i % 7 == 0) c2.resize(rng() % 100);
        if (build_wrapper(q, c2, 1 + (unsigned)(rng() % 3), 0, 1 | 8 | (i % 2 ? 2 : 0), w, wmap, err) == 0) built++; else refused++;
    }
    // ---- the gadget circuits and random programs over the builder's gadgets (csrc/gadget_circuits.cpp), incl. kinds that do not exist ----
    for (unsigned kind : {0u, 1u, 2u, 3u, 4u, 5u, 6u, 7u, 8u, 9u, 999u, 3000u, 3050u, 7096u, 7097u}) {
        size_t nw = 0, ni = 0, no = 0;
        if (qpgpu_builder_gadget_circuit(kind, nullptr, 0, &nw, nullptr, 0, &ni, &no, err) != 0) { refused++; continue; }
        std::vector<uint64_t> gp(nw), cells(ni + no);
        if (qpgpu_builder_gadget_circuit(kind, gp.data(), gp.size(), &nw, cells.data(), cells.size(), &ni, &no, err) != 0) { fprintf(stderr, "gadget circuit %u: %s\n", kind, err); return 1; }
        if (qpgpu_pack_validate(gp.data(), gp.size(), err)) { fprintf(stderr, "gadget circuit %u does not validate: %s\n", kind, err); return 1; }
        if (qpgpu_builder_gadget_circuit(kind, gp.data(), gp.size() / 2, &nw, cells.data(), cells.size(), &ni, &no, err) == 0) { fprintf(stderr, "short pack buffer accepted\n"); return 1; }
        built++;
    }
    for (int i = 0; i < std::max(iters / 2, 8); i++) {
        size_t nw = 0, ni = 0, no = 0;
        const unsigned kind = 1000 + (unsigned)(rng() % 1900);
        if (qpgpu_builder_gadget_circuit(kind, nullptr, 0, &nw, nullptr, 0, &ni, &no, err) != 0) { fprintf(stderr, "random program %u: %s\n", kind, err); return 1; }
        std::vector<uint64_t> gp(nw), cells(ni + no);
        if (qpgpu_builder_gadget_circuit(kind, gp.data(), gp.size(), &nw, cells.data(), cells.size(), &ni, &no, err) != 0 || qpgpu_pack_validate(gp.data(), gp.size(), err)) { fprintf(stderr, "random program %u: %s\n", kind, err); return 1; }
        built++;
    }
    { size_t g = 0; if (qpgpu_builder_sort_gate_cost(8, 60, &g, err) != 0 || g == 0 || qpgpu_builder_sort_gate_cost(5000, 60, &g, err) == 0) { fprintf(stderr, "sort gate cost\n"); return 1; } }
    printf("builder driver: %ld circuits built, %ld refused, %ld proof-target fills, %ld refused fills\n", built, refused, filled, fill_refused);
    return 0;
}
