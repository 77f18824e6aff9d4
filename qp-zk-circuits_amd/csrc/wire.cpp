// wire.cpp — proof hex, config.json, artifact names and the circuit-pack validator (host only). C ABI and reference
// citations: include/qpgpu_wire.h.
#include <cctype>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>
#include "../../include/qpgpu_wire.h"
#include "circuit.hpp"
#include "gl64.hpp"

namespace {

int fail(char *err, const std::string &msg) {
    if (err) { std::snprintf(err, QPGPU_WIRE_ERR_CAP, "%s", msg.c_str()); }
    return -1;
}

int hexval(char c) {
    if (c >= '0' && c <= '9') return c - '0';
    if (c >= 'a' && c <= 'f') return c - 'a' + 10;
    if (c >= 'A' && c <= 'F') return c - 'A' + 10;
    return -1;
}

// ---- the JSON subset config.json needs: one object of string keys with unsigned-integer or null values ----
struct Json {
    const char *p, *end;
    void ws() { while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) p++; }
    bool lit(char c) { ws(); if (p < end && *p == c) { p++; return true; } return false; }
    bool str(std::string &out) {
        ws();
        if (p >= end || *p != '"') return false;
        p++;
        out.clear();
        while (p < end && *p != '"') {
            if (*p == '\\') { if (p + 1 >= end) return false; out.push_back(p[1]); p += 2; }   // escapes do not occur in the keys we care about
            else out.push_back(*p++);
        }
        if (p >= end) return false;
        p++;
        return true;
    }
    // value: unsigned integer, null, or anything else (skipped for unknown keys)
    enum Kind { UINT, NUL, OTHER, BAD };
    Kind value(uint64_t &v) {
        ws();
        if (p >= end) return BAD;
        if (end - p >= 4 && std::memcmp(p, "null", 4) == 0) { p += 4; return NUL; }
        if (*p >= '0' && *p <= '9') {
            v = 0;
            const char *s = p;
            while (p < end && *p >= '0' && *p <= '9') { if (v > (UINT64_MAX - 9) / 10) return BAD; v = v * 10 + (uint64_t)(*p - '0'); p++; }
            if (p < end && (*p == '.' || *p == 'e' || *p == 'E')) return BAD;      // usize: no fraction / exponent
            if (p - s > 1 && *s == '0') return BAD;                                 // no leading zeros in JSON
            return UINT;
        }
        // a value of an unknown key: skipped (serde ignores unknown fields)
        if (*p == '"') { std::string tmp; return str(tmp) ? OTHER : BAD; }
        if (*p == '{' || *p == '[') {
            int depth = 0;
            while (p < end) {
                if (*p == '"') { std::string tmp; if (!str(tmp)) return BAD; continue; }
                if (*p == '{' || *p == '[') depth++;
                if (*p == '}' || *p == ']') { depth--; if (depth == 0) { p++; return OTHER; } }
                p++;
            }
            return BAD;
        }
        while (p < end && *p != ',' && *p != '}' && *p != ' ' && *p != '\n' && *p != '\r' && *p != '\t') p++;   // true / false / a number
        return OTHER;
    }
};

int validate_count(uint64_t n, const char *label, char *err) {
    if (n == 0) return fail(err, std::string(label) + " must be > 0");
    if (n > QPGPU_MAX_PROOF_COUNT) return fail(err, std::string(label) + " (" + std::to_string(n) + ") exceeds maximum allowed (" + std::to_string(QPGPU_MAX_PROOF_COUNT) + ")");
    return 0;
}

}  // namespace

extern "C" {

size_t qpgpu_hex_encode(const uint8_t *in, size_t len, char *out, size_t out_cap) {
    if ((!in && len) || !out || out_cap < 2 * len + 1) return 0;
    static const char digits[] = "0123456789abcdef";
    for (size_t i = 0; i < len; i++) { out[2 * i] = digits[in[i] >> 4]; out[2 * i + 1] = digits[in[i] & 15]; }
    out[2 * len] = 0;
    return 2 * len;
}

size_t qpgpu_hex_decode(const char *in, size_t len, uint8_t *out, size_t out_cap) {
    if ((!in && len) || (len & 1) || (!out && len) || out_cap < len / 2) return (size_t)-1;
    for (size_t i = 0; i < len / 2; i++) {
        const int h = hexval(in[2 * i]), l = hexval(in[2 * i + 1]);
        if (h < 0 || l < 0) return (size_t)-1;
        out[i] = (uint8_t)(h << 4 | l);
    }
    return len / 2;
}

int qpgpu_bins_config_validate(const qpgpu_bins_config *cfg, char *err) {
    if (err) err[0] = 0;
    if (!cfg) return fail(err, "null config");
    if (validate_count(cfg->num_leaf_proofs, "num_leaf_proofs", err)) return -1;
    if (cfg->has_num_private_batch_proofs && validate_count(cfg->num_private_batch_proofs, "num_private_batch_proofs", err)) return -1;
    return 0;
}

int qpgpu_bins_config_parse(const char *json, size_t len, qpgpu_bins_config *out, char *err) {
    if (err) err[0] = 0;
    if (!json || !out) return fail(err, "null argument");
    Json j{json, json + len};
    if (!j.lit('{')) return fail(err, "failed to parse config.json: expected an object");
    bool have_leaf = false, have_priv = false;
    qpgpu_bins_config c{};
    if (!j.lit('}')) {
        for (;;) {
            std::string key;
            if (!j.str(key) || !j.lit(':')) return fail(err, "failed to parse config.json: expected a key");
            uint64_t v = 0;
            const Json::Kind k = j.value(v);
            if (k == Json::BAD) return fail(err, "failed to parse config.json: bad value for `" + key + "`");
            if (key == "num_leaf_proofs") {
                if (have_leaf) return fail(err, "failed to parse config.json: duplicate field `num_leaf_proofs`");
                if (k != Json::UINT) return fail(err, "failed to parse config.json: invalid type for `num_leaf_proofs`, expected usize");
                have_leaf = true; c.num_leaf_proofs = v;
            } else if (key == "num_private_batch_proofs" || key == "num_layer0_proofs") {
                if (have_priv) return fail(err, "failed to parse config.json: duplicate field `num_private_batch_proofs`");
                if (k != Json::UINT && k != Json::NUL) return fail(err, "failed to parse config.json: invalid type for `num_private_batch_proofs`, expected usize or null");
                have_priv = true; c.has_num_private_batch_proofs = k == Json::UINT; c.num_private_batch_proofs = k == Json::UINT ? v : 0;
            }
            if (j.lit(',')) continue;
            if (j.lit('}')) break;
            return fail(err, "failed to parse config.json: expected `,` or `}`");
        }
    }
    j.ws();
    if (j.p != j.end) return fail(err, "failed to parse config.json: trailing characters");
    if (!have_leaf) return fail(err, "failed to parse config.json: missing field `num_leaf_proofs`");
    // a missing Option field deserialises as None
    if (qpgpu_bins_config_validate(&c, err)) return -1;
    *out = c;
    return 0;
}

size_t qpgpu_bins_config_write(const qpgpu_bins_config *cfg, char *out, size_t out_cap) {
    if (!cfg || !out || qpgpu_bins_config_validate(cfg, nullptr)) return 0;
    char buf[160];
    int n;
    if (cfg->has_num_private_batch_proofs)
        n = std::snprintf(buf, sizeof buf, "{\n  \"num_leaf_proofs\": %llu,\n  \"num_private_batch_proofs\": %llu\n}",
                          (unsigned long long)cfg->num_leaf_proofs, (unsigned long long)cfg->num_private_batch_proofs);
    else
        n = std::snprintf(buf, sizeof buf, "{\n  \"num_leaf_proofs\": %llu,\n  \"num_private_batch_proofs\": null\n}", (unsigned long long)cfg->num_leaf_proofs);
    if (n <= 0 || (size_t)n + 1 > out_cap) return 0;
    std::memcpy(out, buf, (size_t)n + 1);
    return (size_t)n;
}

const char *qpgpu_artifact_name(int level, int kind) {
    static const char *names[3][4] = {
        {"common.bin", "verifier.bin", "dummy_proof.bin", "prover_pack.qpcp"},
        {"private_batch_common.bin", "private_batch_verifier.bin", "dummy_private_batch_proof.bin", "private_batch_prover_pack.qpcp"},
        {"public_batch_common.bin", "public_batch_verifier.bin", nullptr, "public_batch_prover_pack.qpcp"}};
    if (kind == QPGPU_ARTIFACT_CONFIG) return "config.json";
    if (level < 0 || level > 2 || kind < 0 || kind > 3) return nullptr;
    return names[level][kind];
}

int qpgpu_pack_validate(const uint64_t *pack_words, size_t n_words, char *err) {
    if (err) err[0] = 0;
    if (!pack_words) return fail(err, "null pack");
    CircuitPack p;
    const std::string perr = p.parse(pack_words, n_words);     // sections, header ranges, gate parameters, hint / public-input cells
    if (!perr.empty()) return fail(err, perr);
    const uint64_t n = p.n(), R = p.num_routed_wires, ng = p.gates.size();
    // FRI schedule against the tree heights
    { uint64_t L = p.degree_bits + p.rate_bits;
      for (size_t r = 0; r < p.arity_bits.size(); r++) {
          if (L < p.arity_bits[r] + p.cap_height) return fail(err, "FRI reduction round " + std::to_string(r) + " leaves a tree of 2^" + std::to_string(L - p.arity_bits[r]) + " leaves, below the cap height " + std::to_string(p.cap_height));
          L -= p.arity_bits[r];
      } }
    // selector groups: every gate in exactly one contiguous group, one selector column per group
    { std::vector<int> seen(ng, 0);
      for (uint64_t i = 0; i < ng; i++) {
          const GateInfo &g = p.gates[i];
          if (!(g.group_start <= i && i < g.group_end)) return fail(err, "gate " + std::to_string(i) + " is outside its own selector group [" + std::to_string(g.group_start) + ", " + std::to_string(g.group_end) + ")");
          for (uint64_t j = g.group_start; j < g.group_end; j++)
              if (p.gates[j].group_start != g.group_start || p.gates[j].group_end != g.group_end || p.gates[j].selector_index != g.selector_index)
                  return fail(err, "selector groups overlap: gates " + std::to_string(i) + " and " + std::to_string(j) + " disagree about their group or selector column");
          seen[i] = 1;
      }
      for (uint64_t i = 0; i < ng; i++) for (uint64_t j = 0; j < ng; j++)
          if (p.gates[i].selector_index == p.gates[j].selector_index && p.gates[i].group_start != p.gates[j].group_start)
              return fail(err, "two selector groups share selector column " + std::to_string(p.gates[i].selector_index)); }
    // selector columns: every row selects exactly one gate, the value is the gate's index inside the column of its group
    const uint64_t UNUSED = 0xFFFFFFFFull;
    for (uint64_t r = 0; r < n; r++) {
        int hits = 0;
        for (uint64_t s = 0; s < p.num_selectors; s++) {
            const uint64_t v = p.constants_sigmas[s * n + r];
            if (v == UNUSED && p.num_selectors > 1) continue;
            if (v >= ng || p.gates[v].selector_index != s)
                return fail(err, "row " + std::to_string(r) + ": selector column " + std::to_string(s) + " holds " + std::to_string(v) + ", which names no gate of that column's group");
            hits++;
        }
        if (hits != 1) return fail(err, "row " + std::to_string(r) + " selects " + std::to_string(hits) + " gates (exactly one expected)");
    }
    // k_is: distinct cosets of the subgroup; sigma: a permutation of the n * R routed cells
    std::unordered_map<uint64_t, uint32_t> col_of;
    std::vector<uint64_t> k_inv(R);
    for (uint64_t c = 0; c < R; c++) {
        uint64_t t = p.k_is[c];
        if (gl::canon(t) == 0) return fail(err, "k_is[" + std::to_string(c) + "] is zero");
        for (uint64_t i = 0; i < p.degree_bits; i++) t = gl::mul(t, t);
        if (!col_of.emplace(gl::canon(t), (uint32_t)c).second) return fail(err, "k_is[" + std::to_string(c) + "] lies in the same coset of the subgroup as an earlier one");
        k_inv[c] = gl::inv(p.k_is[c]);
    }
    std::unordered_map<uint64_t, uint32_t> row_of;
    row_of.reserve(n * 2);
    { const uint64_t w = gl::root_of_unity((unsigned)p.degree_bits); uint64_t a = 1; for (uint64_t r = 0; r < n; r++) { row_of[gl::canon(a)] = (uint32_t)r; a = gl::mul(a, w); } }
    const uint64_t sig0 = p.num_selectors + p.num_constants;
    std::vector<uint8_t> hit(n * R, 0);
    for (uint64_t c = 0; c < R; c++)
        for (uint64_t r = 0; r < n; r++) {
            const uint64_t s = p.constants_sigmas[(sig0 + c) * n + r];
            if (s >= gl::P) return fail(err, "sigma[" + std::to_string(c) + "][" + std::to_string(r) + "] is not a canonical field element");
            uint64_t t = s;
            for (uint64_t i = 0; i < p.degree_bits; i++) t = gl::mul(t, t);
            auto ci = col_of.find(gl::canon(t));
            if (ci == col_of.end()) return fail(err, "sigma is not a permutation: sigma[" + std::to_string(c) + "][" + std::to_string(r) + "] lies outside every wire coset k_is[j] * H");
            auto ri = row_of.find(gl::canon(gl::mul(s, k_inv[ci->second])));
            if (ri == row_of.end()) return fail(err, "sigma is not a permutation: sigma[" + std::to_string(c) + "][" + std::to_string(r) + "] is not k_is[j] * w^i");
            uint8_t &h = hit[(uint64_t)ri->second * R + ci->second];
            if (h) return fail(err, "sigma is not a permutation: two cells map to (row " + std::to_string(ri->second) + ", wire " + std::to_string(ci->second) + ")");
            h = 1;
        }
    return 0;
}


// ---- CircuitConfig policy ----
int qpgpu_wormhole_circuit_config(int level, qpgpu_circuit_config *out) {
    if (!out || level < QPGPU_LEVEL_LEAF || level > QPGPU_LEVEL_PUBLIC_BATCH) return -1;
    // CircuitConfig::standard_recursion_config (SURVEY.md section 8 "Parameters common to all rows")
    qpgpu_circuit_config c{};
    c.num_wires = 135; c.num_routed_wires = 80; c.num_constants = 2; c.use_base_arithmetic_gate = 1; c.security_bits = 100;
    c.num_challenges = 2; c.zero_knowledge = 0; c.max_quotient_degree_factor = 8;
    c.rate_bits = 3; c.cap_height = 4; c.proof_of_work_bits = 16; c.num_query_rounds = 28;
    c.reduction_arity_bits = 4; c.reduction_final_poly_bits = 5;
    if (level == QPGPU_LEVEL_PRIVATE_BATCH) { c.zero_knowledge = 1; c.num_routed_wires = 60; }   // wormhole_private_batch_circuit_config
    *out = c;
    return 0;
}

static int cfg_fail(char *err, const char *fmt, ...) {
    if (err) { va_list ap; va_start(ap, fmt); vsnprintf(err, QPGPU_CONFIG_ERR_CAP, fmt, ap); va_end(ap); }
    return -1;
}

int qpgpu_validate_circuit_config(const qpgpu_circuit_config *c, char *err) {
    if (!c) return cfg_fail(err, "null argument");
    const struct { const char *name; uint64_t v; } zero[3] = {{"num_challenges", c->num_challenges}, {"security_bits", c->security_bits},
                                                              {"fri_config.num_query_rounds", c->num_query_rounds}};
    for (const auto &z : zero) if (z.v == 0) return cfg_fail(err, "circuit config %s must be greater than 0", z.name);
    if (c->num_wires < 135) return cfg_fail(err, "circuit config num_wires (%llu) must be >= 135 (Poseidon gate floor)", (unsigned long long)c->num_wires);
    if (c->num_routed_wires < 37)
        return cfg_fail(err, "circuit config num_routed_wires (%llu) must be >= 37 (recursion gate floor: the FRI coset-interpolation gate routes 37 wires, "
                             "and narrower widths leave slot-packed gates with zero operation slots)", (unsigned long long)c->num_routed_wires);
    if (c->num_routed_wires > c->num_wires)
        return cfg_fail(err, "circuit config num_routed_wires (%llu) must be <= num_wires (%llu); routed wires are a prefix of the wire columns",
                        (unsigned long long)c->num_routed_wires, (unsigned long long)c->num_wires);
    if (c->max_quotient_degree_factor < 7)
        return cfg_fail(err, "circuit config max_quotient_degree_factor (%llu) must be >= 7 (Poseidon constraint degree)", (unsigned long long)c->max_quotient_degree_factor);
    if (c->rate_bits > 8)
        return cfg_fail(err, "circuit config fri_config.rate_bits (%llu) must be <= 8 (LDE memory doubles per bit: lde_size = 2^(degree_bits + rate_bits) per "
                             "committed polynomial)", (unsigned long long)c->rate_bits);
    if (c->cap_height > 8)
        return cfg_fail(err, "circuit config fri_config.cap_height (%llu) must be <= 8 (Merkle caps and recursive verifier-data allocations scale as "
                             "2^cap_height)", (unsigned long long)c->cap_height);
    uint64_t bits = 0;
    while ((1ull << bits) < c->max_quotient_degree_factor && bits < 63) bits++;
    if (c->rate_bits < bits)
        return cfg_fail(err, "circuit config fri_config.rate_bits (%llu) must be >= ceil(log2(max_quotient_degree_factor = %llu)) = %llu; plonky2's prover cannot "
                             "compute quotient chunks of degree higher than the FRI rate and asserts this only at proving time, after the full circuit build",
                        (unsigned long long)c->rate_bits, (unsigned long long)c->max_quotient_degree_factor, (unsigned long long)bits);
    return 0;
}

int qpgpu_pack_config_is_canonical(const uint64_t *w, size_t n, int level, char *err) {
    static const char *labels[3] = {"leaf", "private-batch", "public-batch"};
    qpgpu_circuit_config c;
    if (!w || qpgpu_wormhole_circuit_config(level, &c)) return cfg_fail(err, "bad argument");
    if (n < 18 || w[0] != 0x0000003150435051ull) return cfg_fail(err, "not a circuit pack");
    // header words (csrc/circuit.hpp): 2 num_wires, 3 routed, 6 challenges, 7 quotient degree factor, 10 rate, 11 cap, 12 pow, 13 queries, 14 zk
    const struct { const char *name; uint64_t got, want; } f[9] = {
        {"num_wires", w[2], c.num_wires}, {"num_routed_wires", w[3], c.num_routed_wires}, {"num_challenges", w[6], c.num_challenges},
        {"quotient_degree_factor", w[7], c.max_quotient_degree_factor}, {"fri_config.rate_bits", w[10], c.rate_bits},
        {"fri_config.cap_height", w[11], c.cap_height}, {"fri_config.proof_of_work_bits", w[12], c.proof_of_work_bits},
        {"fri_config.num_query_rounds", w[13], c.num_query_rounds}, {"zero_knowledge", w[14], (uint64_t)c.zero_knowledge}};
    for (const auto &x : f)
        if (x.got != x.want)
            return cfg_fail(err, "loaded %s circuit config does not match the canonical Wormhole config (%s loaded=%llu, expected=%llu)", labels[level], x.name,
                            (unsigned long long)x.got, (unsigned long long)x.want);
    return 0;
}

}  // extern "C"
