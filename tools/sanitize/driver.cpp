// Sanitizer driver for the host-only part of the library (built by tests/soak/sanitize_host.py with g++ -fsanitize=address,undefined):
// feeds the parsers and the verifier valid, mutated, truncated and random inputs. Any out-of-bounds access, overflow or
// leak aborts the run; the functions themselves must only ever return error codes.
// usage: driver <pack.bin> <proof.bin> [iterations]
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>
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

int main(int argc, char **argv) {
    if (argc < 3) { fprintf(stderr, "usage: driver pack.bin proof.bin [iterations]\n"); return 2; }
    const std::vector<uint8_t> pack_bytes = slurp(argv[1]), proof = slurp(argv[2]);
    const int iters = argc > 3 ? atoi(argv[3]) : 300;
    std::vector<uint64_t> pack(pack_bytes.size() / 8);
    memcpy(pack.data(), pack_bytes.data(), pack.size() * 8);
    std::mt19937_64 rng(12345);
    char err[512];
    long accepted = 0, rejected = 0, refused_packs = 0;

    // ---- verifier: valid proof, then mutations, truncations, random bytes ----
    qpgpu_verifier *v = nullptr;
    if (qpgpu_verifier_create(pack.data(), pack.size(), nullptr, 0, 0, nullptr, 0, &v, err)) { fprintf(stderr, "verifier_create: %s\n", err); return 1; }
    if (qpgpu_verifier_verify(v, proof.data(), proof.size(), err)) { fprintf(stderr, "valid proof rejected: %s\n", err); return 1; }
    accepted++;
    for (int i = 0; i < iters; i++) {
        std::vector<uint8_t> p = proof;
        const int kind = i % 4;
        if (kind == 0) p[rng() % p.size()] ^= (uint8_t)(1u << (rng() % 8));
        else if (kind == 1) { const size_t at = rng() % p.size(); for (int k = 0; k < 8 && at + k < p.size(); k++) p[at + k] = 0xFF; }   // non-canonical words, huge path lengths
        else if (kind == 2) p.resize(rng() % p.size());
        else for (auto &b : p) b = (uint8_t)rng();
        if (qpgpu_verifier_verify(v, p.data(), p.size(), err) == 0) { fprintf(stderr, "mutation %d accepted\n", i); return 1; }
        rejected++;
    }
    qpgpu_verifier_free(v);
    // ---- pack parser / validator / verifier construction on mutated packs ----
    for (int i = 0; i < iters; i++) {
        std::vector<uint64_t> q = pack;
        const int kind = i % 3;
        if (kind == 0) q[rng() % std::min<size_t>(q.size(), 64)] = rng() >> (rng() % 64);
        else if (kind == 1) q.resize(rng() % q.size());
        else q[rng() % q.size()] ^= 1ull << (rng() % 64);
        (void)qpgpu_pack_validate(q.data(), q.size(), err);
        qpgpu_verifier *w = nullptr;
        if (qpgpu_verifier_create(q.data(), q.size(), nullptr, 0, 0, nullptr, 0, &w, err) == 0) {
            (void)qpgpu_verifier_verify(w, proof.data(), proof.size(), err);
            qpgpu_verifier_free(w);
        } else refused_packs++;
    }
    // ---- inner-proof targets: shapes and assignments from valid, mutated, truncated and random proofs and packs ----
    {
        size_t nt = 0, np_ = 0, nv = 0;
        std::vector<uint32_t> tshape(8192), pshape(8192);
        if (qpgpu_proof_target_shape(pack.data(), pack.size(), tshape.data(), tshape.size(), &nt, err)) { fprintf(stderr, "target_shape: %s\n", err); return 1; }
        const size_t T = qpgpu_proof_target_count(pack.data(), pack.size());
        std::vector<uint64_t> vals(2 * (T + 4) + 8);
        std::vector<uint32_t> ids(2 * (T + 4) + 8);
        if (qpgpu_proof_target_values(pack.data(), pack.size(), proof.data(), proof.size(), 0, "leaf proof", vals.data(), vals.size(), &nv, err) || nv != T) { fprintf(stderr, "target_values on the valid proof: %s\n", err); return 1; }
        for (int i = 0; i < iters; i++) {
            std::vector<uint8_t> p = proof;
            const int kind = i % 5;
            if (kind == 0) p[rng() % p.size()] ^= (uint8_t)(1u << (rng() % 8));
            else if (kind == 1) { const size_t at = rng() % p.size(); for (int k = 0; k < 8 && at + k < p.size(); k++) p[at + k] = 0xFF; }
            else if (kind == 2) p.resize(rng() % p.size());
            else if (kind == 3) p.resize(p.size() + 8 * (rng() % 5));
            else for (auto &b : p) b = (uint8_t)rng();
            if (qpgpu_proof_shape_of_bytes(pack.data(), pack.size(), p.data(), p.size(), pshape.data(), pshape.size(), &np_, err) == 0)
                (void)qpgpu_ensure_proof_shape_matches_targets(tshape.data(), nt, pshape.data(), np_, (size_t)i, "leaf proof", err);
            (void)qpgpu_proof_target_values(pack.data(), pack.size(), p.data(), p.size(), 1, nullptr, vals.data(), vals.size(), &nv, err);
            const uint8_t *two[2] = {proof.data(), p.data()};
            const size_t lens[2] = {proof.size(), p.size()};
            uint64_t pre[8];
            for (auto &x : pre) x = (rng() % 7 == 0) ? rng() : rng() % 0xFFFFFFFF00000001ull;
            size_t cnt = 0;
            (void)qpgpu_batch_fill_proof_targets(pack.data(), pack.size(), two, lens, 2, 1 + rng() % 3, pre, 1 + rng() % 3, 1 + rng() % 3, "leaf proof", ids.data(), vals.data(), ids.size(), &cnt, err);
            (void)qpgpu_batch_fill_proof_targets(pack.data(), pack.size(), two, lens, 2, 2, pre, 2, 2, "leaf proof", ids.data(), vals.data(), ids.size(), &cnt, err);
            // shape descriptors: mutated and truncated words must be answered, never indexed past
            std::vector<uint32_t> q(tshape.begin(), tshape.begin() + nt);
            if (i % 2) q[rng() % q.size()] = (uint32_t)(rng() >> (rng() % 32)); else q.resize(rng() % q.size());
            (void)qpgpu_ensure_proof_shape_matches_targets(tshape.data(), nt, q.data(), q.size(), 0, "x", err);
            (void)qpgpu_ensure_proof_shape_matches_targets(q.data(), q.size(), tshape.data(), nt, 0, "x", err);
            // and a mutated pack as the inner circuit
            std::vector<uint64_t> qp = pack;
            if (i % 3 == 0) qp[rng() % std::min<size_t>(qp.size(), 40)] = rng() >> (rng() % 64); else if (i % 3 == 1) qp.resize(rng() % qp.size());
            (void)qpgpu_proof_target_count(qp.data(), qp.size());
            (void)qpgpu_proof_target_values(qp.data(), qp.size(), proof.data(), proof.size(), 0, "leaf proof", vals.data(), vals.size(), &nv, err);
        }
    }
    // ---- public-input parsers, admission checks and wrapper outputs on random rows ----
    for (int i = 0; i < iters; i++) {
        const size_t n_leaf = 1 + rng() % 64, count = 1 + rng() % n_leaf;
        std::vector<uint64_t> rows(n_leaf * 21), pre(n_leaf * 4), out(21 * n_leaf + 8);
        for (auto &x : rows) x = (rng() % 5 == 0) ? rng() : rng() % 7;        // mostly small values so that some batches are consistent
        for (auto &x : pre) x = rng() % 0xFFFFFFFF00000001ull;
        (void)qpgpu_private_batch_preflight(rows.data(), count, n_leaf, err);
        (void)qpgpu_dummy_leaf_template_check(rows.data(), 21, err);
        if (qpgpu_private_batch_outputs(rows.data(), n_leaf, pre.data(), out.data(), err) == 0) {
            qpgpu_private_batch_public_inputs hdr;
            std::vector<qpgpu_exit_slot> slots(128);
            std::vector<uint8_t> nulls(64 * 32);
            // (a leaf row with a field element above u32 where the layout has a u32 is provable and unparseable, as in the reference)
            (void)qpgpu_private_batch_public_inputs_parse(out.data(), out.size(), &hdr, slots.data(), nulls.data(), err);
            (void)qpgpu_dummy_private_batch_template_check(out.data(), out.size(), err);
            const size_t m = 1 + rng() % 4;
            if (n_leaf <= 8) {
                std::vector<uint64_t> inner(m * out.size()), pub(qpgpu_public_batch_pi_len(m, n_leaf));
                for (size_t k = 0; k < m; k++) memcpy(inner.data() + k * out.size(), out.data(), out.size() * 8);
                uint8_t addr[32] = {1, 2, 3};
                (void)qpgpu_public_batch_preflight(inner.data(), m, out.size(), m, err);
                if (qpgpu_public_batch_outputs(inner.data(), m, n_leaf, addr, pub.data(), err) == 0) {
                    qpgpu_public_batch_public_inputs ph;
                    (void)qpgpu_public_batch_public_inputs_parse(pub.data(), pub.size(), m, n_leaf, &ph, nullptr, nullptr, err);
                }
            }
        }
        std::vector<uint64_t> junk(rng() % 300);
        for (auto &x : junk) x = rng();
        qpgpu_private_batch_public_inputs hdr; qpgpu_public_batch_public_inputs ph; qpgpu_leaf_public_inputs lp;
        (void)qpgpu_private_batch_public_inputs_parse(junk.data(), junk.size(), &hdr, nullptr, nullptr, err);
        (void)qpgpu_public_batch_public_inputs_parse(junk.data(), junk.size(), rng() % 70, rng() % 70, &ph, nullptr, nullptr, err);
        (void)qpgpu_leaf_public_inputs_parse(junk.data(), junk.size(), &lp, err);
        std::vector<uint32_t> src(n_leaf);
        uint8_t seed[32]; for (auto &b : seed) b = (uint8_t)rng();
        if (qpgpu_private_batch_arrange(count, n_leaf, seed, src.data(), pre.data(), err)) { fprintf(stderr, "arrange: %s\n", err); return 1; }
    }
    // ---- wire formats on random text ----
    for (int i = 0; i < iters; i++) {
        std::string s(rng() % 200, ' ');
        static const char alphabet[] = "{}[]\":,0123456789abcdefnul_ \n-+.eEtrueflsnum_leaf_proofsnum_private_batch_proofsnum_layer0_proofs";
        for (auto &c : s) c = alphabet[rng() % (sizeof alphabet - 1)];
        qpgpu_bins_config cfg;
        (void)qpgpu_bins_config_parse(s.data(), s.size(), &cfg, err);
        std::vector<uint8_t> bytes(s.size());
        (void)qpgpu_hex_decode(s.data(), s.size(), bytes.data(), bytes.size());
        std::vector<char> hex(2 * bytes.size() + 1);
        (void)qpgpu_hex_encode(bytes.data(), bytes.size(), hex.data(), hex.size());
        std::vector<uint64_t> felts(bytes.size() / 4 + 2);
        (void)qpgpu_bytes_to_felts(bytes.data(), bytes.size(), felts.data(), felts.size());
    }
    // ---- native ZK Merkle tree and leaf constraint check on random bytes ----
    for (int i = 0; i < iters; i++) {
        const size_t depth = rng() % 20;
        std::vector<uint8_t> sib(96 * depth + 128), pos(depth + 1), sorted(96 * depth + 128), posout(depth + 1);
        uint8_t leaf[32], root[32], out[128];
        for (auto &b : sib) b = (uint8_t)rng();
        for (auto &b : pos) b = (uint8_t)(rng() % 6);
        for (auto &b : leaf) b = (uint8_t)rng();
        if (i % 2) for (size_t k = 7; k < sib.size(); k += 8) sib[k] &= 0x7F;      // canonical limbs half of the time
        if (i % 2) for (size_t k = 7; k < 32; k += 8) leaf[k] &= 0x7F;
        (void)qpgpu_zk_proof_verify(leaf, sib.data(), pos.data(), depth, root);
        if (qpgpu_zk_proof_from_unsorted(leaf, sib.data(), depth, sorted.data(), posout.data(), root, err) == 0 &&
            qpgpu_zk_proof_verify(leaf, sorted.data(), posout.data(), depth, root) != 1) { fprintf(stderr, "from_unsorted proof does not verify\n"); return 1; }
        (void)qpgpu_zk_hash_node(sib.data(), out);
        (void)qpgpu_zk_insert_at_position(leaf, sib.data(), (unsigned)(rng() % 6), out);
        qpgpu_leaf_inputs in;
        uint8_t *raw = reinterpret_cast<uint8_t *>(&in);
        for (size_t k = 0; k < sizeof in; k++) raw[k] = (uint8_t)rng();
        if (i % 2) in.zk_merkle_depth %= 17;
        (void)qpgpu_leaf_check_constraints(&in, err);
    }
    printf("ok: %ld accepted, %ld rejected, %ld mutated packs refused\n", accepted, rejected, refused_packs);
    return 0;
}
