// circuit.cpp — circuit pack (de)serialisation and validation. See circuit.hpp for the format.
#include "circuit.hpp"

std::vector<uint64_t> CircuitPack::serialize() const {
    std::vector<uint64_t> w;
    w.push_back(QPCP_MAGIC);
    const uint64_t hdr[] = {degree_bits, num_wires, num_routed_wires, num_constants, num_selectors, num_challenges,
                            quotient_degree_factor, num_partial_products, num_public_inputs, rate_bits, cap_height,
                            proof_of_work_bits, num_query_rounds, zero_knowledge, num_gate_constraints,
                            (uint64_t)gates.size(), (uint64_t)arity_bits.size()};
    w.insert(w.end(), hdr, hdr + 17);
    w.insert(w.end(), arity_bits.begin(), arity_bits.end());
    for (const auto &g : gates) {
        const uint64_t gw[] = {g.type, g.param0, g.param1, g.selector_index, g.group_start, g.group_end, g.num_constraints, g.param2};
        w.insert(w.end(), gw, gw + 8);
    }
    w.insert(w.end(), k_is.begin(), k_is.end());
    w.insert(w.end(), circuit_digest, circuit_digest + 4);
    w.insert(w.end(), constants_sigmas.begin(), constants_sigmas.end());
    if (!hints.empty()) {
        w.push_back(QPCP_HINT_MAGIC); w.push_back(hints.size());
        for (const auto &h : hints) w.insert(w.end(), h.w, h.w + 8);
    }
    if (!pi_cells.empty()) {
        w.push_back(QPCP_PUBI_MAGIC); w.push_back(pi_cells.size());
        w.insert(w.end(), pi_cells.begin(), pi_cells.end());
    }
    if (has_p2_layout) {
        const P2GateLayout &l = p2_layout;
        const uint64_t lw[P2GateLayout::WORDS] = {l.w_input, l.w_output, l.w_swap, l.w_delta, l.w_full0, l.w_partial, l.w_full1,
                                                  l.first_round_wires, l.constraint_order, l.end_wire};
        w.push_back(QPCP_P2GL_MAGIC); w.push_back(P2GateLayout::WORDS);
        w.insert(w.end(), lw, lw + P2GateLayout::WORDS);
    }
    return w;
}

std::string CircuitPack::parse(const uint64_t *words, size_t n_words) {
    size_t pos = 0;
    // `k` words left from `pos`? written so that neither side can wrap
    auto need = [&](uint64_t k) { return k <= (uint64_t)(n_words - pos); };
    if (n_words < 18 || words[0] != QPCP_MAGIC) return "bad magic or truncated header";
    pos = 1;
    uint64_t *fields[] = {&degree_bits, &num_wires, &num_routed_wires, &num_constants, &num_selectors, &num_challenges,
                          &quotient_degree_factor, &num_partial_products, &num_public_inputs, &rate_bits, &cap_height,
                          &proof_of_work_bits, &num_query_rounds, &zero_knowledge, &num_gate_constraints};
    for (auto f : fields) *f = words[pos++];
    uint64_t ng = words[pos++], na = words[pos++];
    // every header word is bounded before it is used in a size, a shift or an index (a corrupt pack must fail here, not in
    // a host read past the buffer or a kernel indexing outside its allocation)
    if (degree_bits > 24 || ng == 0 || ng > 4096 || na > 16 || num_routed_wires == 0 || num_routed_wires > 4096 || num_wires == 0 || num_wires > 4096) return "header field out of range";
    if (num_selectors == 0 || num_selectors > 4096 || num_constants > 4096) return "header field out of range: num_selectors / num_constants";
    if (num_public_inputs > (1ull << 20)) return "header field out of range: num_public_inputs";
    if (rate_bits == 0 || rate_bits > 8) return "header field out of range: rate_bits";
    if (cap_height > 16) return "header field out of range: cap_height";
    if (proof_of_work_bits > 40) return "header field out of range: proof_of_work_bits";
    if (num_query_rounds == 0 || num_query_rounds > 4096) return "header field out of range: num_query_rounds";
    if (num_challenges == 0 || num_challenges > 4) return "unsupported num_challenges";
    if (quotient_degree_factor == 0 || quotient_degree_factor > 256) return "header field out of range: quotient_degree_factor";
    if (num_partial_products > 4096 || num_gate_constraints > (1ull << 20)) return "header field out of range";
    if (!need(na)) return "truncated arity list";
    arity_bits.assign(words + pos, words + pos + na); pos += na;
    if (!need(ng * 8)) return "truncated gate list";
    gates.resize(ng);
    for (auto &g : gates) {
        g.type = words[pos]; g.param0 = words[pos + 1]; g.param1 = words[pos + 2]; g.selector_index = words[pos + 3];
        g.group_start = words[pos + 4]; g.group_end = words[pos + 5]; g.num_constraints = words[pos + 6]; g.param2 = words[pos + 7];
        pos += 8;
    }
    if (!need(num_routed_wires + 4)) return "truncated k_is";
    k_is.assign(words + pos, words + pos + num_routed_wires); pos += num_routed_wires;
    for (int i = 0; i < 4; i++) circuit_digest[i] = words[pos++];
    const uint64_t cs = num_cs_cols() << degree_bits;          // <= 3 * 4096 * 2^24: no overflow
    if (!need(cs)) return "truncated constants_sigmas";
    constants_sigmas.assign(words + pos, words + pos + cs); pos += cs;
    hints.clear(); pi_cells.clear();
    p2_layout = P2GateLayout(); has_p2_layout = false;
    bool seen_hints = false, seen_pubi = false;
    while (pos != n_words) {   // optional trailers, each at most once
        if (!need(2)) return "trailing data";
        const uint64_t magic = words[pos], cnt = words[pos + 1];
        pos += 2;
        if (magic == QPCP_HINT_MAGIC && !seen_hints && !seen_pubi) {
            seen_hints = true;
            if (cnt > (1ull << 28) || !need(cnt * 8)) return "truncated hint list";
            hints.resize(cnt);
            for (auto &h : hints) { for (int i = 0; i < 8; i++) h.w[i] = words[pos + i]; pos += 8; }
        } else if (magic == QPCP_PUBI_MAGIC && !seen_pubi) {
            seen_pubi = true;
            if (cnt != num_public_inputs || !need(cnt)) return "public-input cell list does not match num_public_inputs";
            pi_cells.assign(words + pos, words + pos + cnt); pos += cnt;
        } else if (magic == QPCP_P2GL_MAGIC && !has_p2_layout) {
            if (cnt != P2GateLayout::WORDS || !need(cnt)) return "Poseidon2 gate layout trailer has the wrong size";
            for (int i = 0; i < P2GateLayout::WORDS; i++) if (words[pos + i] > 0xFFFFFFFFull) return "Poseidon2 gate layout field out of range";
            P2GateLayout &l = p2_layout;
            l.w_input = (uint32_t)words[pos]; l.w_output = (uint32_t)words[pos + 1]; l.w_swap = (uint32_t)words[pos + 2]; l.w_delta = (uint32_t)words[pos + 3];
            l.w_full0 = (uint32_t)words[pos + 4]; l.w_partial = (uint32_t)words[pos + 5]; l.w_full1 = (uint32_t)words[pos + 6];
            l.first_round_wires = (uint32_t)words[pos + 7]; l.constraint_order = (uint32_t)words[pos + 8]; l.end_wire = (uint32_t)words[pos + 9];
            has_p2_layout = true;
            pos += cnt;
        } else return "trailing data";
    }
    return validate();
}

std::string CircuitPack::validate() const {
    if (num_routed_wires > num_wires) return "num_routed_wires > num_wires";
    if (num_challenges == 0 || num_challenges > 4) return "unsupported num_challenges";
    if (quotient_degree_factor == 0 || (quotient_degree_factor & (quotient_degree_factor - 1))) return "quotient_degree_factor must be a power of two";
    if (rate_bits == 0 || rate_bits > 8) return "rate_bits outside 1..8";
    if (quotient_degree_factor > (1ull << rate_bits)) return "quotient_degree_factor exceeds the blowup 2^rate_bits";
    if (degree_bits > 24 || proof_of_work_bits > 40 || cap_height > 16) return "header field out of range";
    if (num_partial_products + 1 != (num_routed_wires + quotient_degree_factor - 1) / quotient_degree_factor) return "num_partial_products inconsistent";
    if (cap_height > degree_bits + rate_bits) return "cap_height above tree height";
    if (zero_knowledge > 1) return "zero_knowledge must be 0 or 1";
    uint64_t sum = 0;
    for (auto a : arity_bits) { if (a == 0 || a > 4) return "unsupported FRI arity"; sum += a; }
    if (sum > degree_bits) return "FRI reductions exceed degree";
    for (const auto &g : gates) {
        if (g.type > GATE_POSEIDON2) return "unknown gate type";
        if (g.type == GATE_POSEIDON2) {
            // fail closed: the gate's wire layout lives in un-vendored qp-plonky2 and is NOT known offline, so a pack that selects
            // the gate must say where its wires are ("P2GL1"); an assumed default could prove what the fork's verifier rejects
            if (!has_p2_layout) return "Poseidon2 gate (type 14) without a wire-layout trailer (P2GL1): the layout is not assumed";
            const std::string why = p2_layout.validate(num_wires, num_routed_wires);
            if (!why.empty()) return why;
            if (g.num_constraints != p2_layout.num_constraints()) return "bad poseidon2 gate: constraint count does not match its wire layout";
        }
        if (g.selector_index >= num_selectors) return "gate selector index out of range";
        if (g.group_end > gates.size() || g.group_start >= g.group_end) return "gate group out of range";
        if (g.num_constraints > num_gate_constraints) return "gate constraint count exceeds num_gate_constraints";
        if (g.type == GATE_ARITHMETIC && (g.param0 * 4 > num_routed_wires || g.num_constraints != g.param0 || num_constants < 2)) return "bad arithmetic gate";
        if (g.type == GATE_CONSTANT && (g.param0 > num_constants || g.param0 > num_wires || g.num_constraints != g.param0)) return "bad constant gate";
        if (g.type == GATE_PUBLIC_INPUT && (num_wires < 4 || g.num_constraints != 4)) return "bad public input gate";
        if (g.type == GATE_ARITHMETIC_EXT && (g.param0 * 8 > num_routed_wires || g.num_constraints != 2 * g.param0 || num_constants < 2)) return "bad arithmetic-extension gate";
        if (g.type == GATE_MUL_EXT && (g.param0 * 6 > num_routed_wires || g.num_constraints != 2 * g.param0 || num_constants < 1)) return "bad mul-extension gate";
        if (g.type == GATE_BASE_SUM && (g.param0 == 0 || g.param0 > 63 || g.param0 + 1 > num_routed_wires || g.num_constraints != g.param0 + 1)) return "bad base-sum gate";
        if (g.type == GATE_REDUCING && (g.param0 == 0 || 6 + g.param0 > num_routed_wires || 6 + 3 * g.param0 - 2 > num_wires || g.num_constraints != 2 * g.param0)) return "bad reducing gate";
        if (g.type == GATE_REDUCING_EXT && (g.param0 == 0 || 6 + 2 * g.param0 > num_routed_wires || 6 + 4 * g.param0 - 2 > num_wires || g.num_constraints != 2 * g.param0)) return "bad reducing-extension gate";
        if (g.type == GATE_RANDOM_ACCESS) {
            const uint64_t bits = g.param0, copies = g.param1, extra = g.param2;
            if (bits == 0 || bits > 5 || copies == 0) return "bad random-access gate";
            const uint64_t routed = (2 + (1ull << bits)) * copies + extra;
            if (routed > num_routed_wires || routed + copies * bits > num_wires || extra > num_constants || g.num_constraints != copies * (bits + 2) + extra) return "bad random-access gate";
        }
        if (g.type == GATE_EXPONENTIATION && (g.param0 == 0 || g.param0 + 2 > num_routed_wires || 2 * g.param0 + 2 > num_wires || g.num_constraints != g.param0 + 1)) return "bad exponentiation gate";
        if (g.type == GATE_POSEIDON_MDS && (num_routed_wires < 48 || g.num_constraints != 24)) return "bad poseidon-mds gate";
        if (g.type == GATE_COSET_INTERPOLATION) {
            const uint64_t bits = g.param0, deg = g.param1;
            if (bits < 2 || bits > 5 || deg < 2) return "bad coset-interpolation gate";
            const uint64_t np = 1ull << bits, ni = (np - 2) / (deg - 1), start_int = 1 + 2 * np + 4;
            if (start_int > num_routed_wires || start_int + 2 * (2 * ni + 1) > num_wires || g.num_constraints != 2 * (2 + 2 * ni)) return "bad coset-interpolation gate";
        }
        if (g.type == GATE_POSEIDON && (num_wires < 135 || num_routed_wires < 25 || g.num_constraints != 123)) return "bad poseidon gate";
    }
    if (constants_sigmas.size() != (num_cs_cols() << degree_bits)) return "constants_sigmas has the wrong size";
    if (k_is.size() != num_routed_wires) return "k_is has the wrong size";
    const uint64_t n_cells = num_wires << degree_bits;
    if (!pi_cells.empty() && pi_cells.size() != num_public_inputs) return "public-input cell list does not match num_public_inputs";
    for (uint64_t c : pi_cells) if (c >= n_cells || c % num_wires >= num_routed_wires) return "public-input cell is not a routed wire of the trace";
    auto routed_cell = [&](uint64_t c) { return c < n_cells && c % num_wires < num_routed_wires; };
    for (const auto &h : hints) {
        static const int n_cells_of[8] = {0, 2, 4, 2, 6, 1, 2, 3};   // leading arguments that are cells, per opcode
        if (h.w[0] < HINT_COPY || h.w[0] > HINT_LOW_HIGH) return "unknown hint opcode";
        for (int i = 0; i < n_cells_of[h.w[0]]; i++) if (!routed_cell(h.w[1 + i])) return "hint cell is not a routed wire of the trace";
        if (h.w[0] == HINT_WIRE_SPLIT && (h.w[3] > 63 || h.w[4] == 0 || h.w[4] > 63)) return "bad wire-split hint";
        if (h.w[0] == HINT_LOW_HIGH && (h.w[4] == 0 || h.w[4] > 63)) return "bad low-high hint";
    }
    return "";
}

// every wire block inside the trace, inputs / outputs / swap routed, no two blocks overlapping
std::string P2GateLayout::validate(uint64_t num_wires, uint64_t num_routed) const {
    if (first_round_wires > 1) return "poseidon2 gate layout: first_round_wires must be 0 or 1";
    if (constraint_order != 0) return "poseidon2 gate layout: unknown constraint order";
    struct Blk { uint64_t lo, len; bool routed; };
    std::vector<Blk> b = {{w_input, 12, true}, {w_output, 12, true}, {w_full0, 12ull * full0_rounds(), false}, {w_partial, 22, false}, {w_full1, 48, false}};
    if (has_swap()) { b.push_back({w_swap, 1, true}); b.push_back({w_delta, 4, false}); }
    for (const Blk &x : b) {
        if (x.lo + x.len > num_wires || x.lo + x.len > end_wire) return "poseidon2 gate layout: a wire block lies outside the gate's wires";
        if (x.routed && x.lo + x.len > num_routed) return "poseidon2 gate layout: inputs, outputs and the swap wire must be routed wires";
    }
    if (end_wire > num_wires) return "poseidon2 gate layout: end_wire exceeds num_wires";
    for (size_t i = 0; i < b.size(); i++)
        for (size_t j = i + 1; j < b.size(); j++)
            if (b[i].lo < b[j].lo + b[j].len && b[j].lo < b[i].lo + b[i].len) return "poseidon2 gate layout: wire blocks overlap";
    return "";
}

std::vector<uint64_t> fri_reduction_arity_bits(uint64_t degree_bits, uint64_t rate_bits, uint64_t cap_height,
                                               uint64_t arity_bits, uint64_t final_poly_bits) {
    // FriReductionStrategy::ConstantArityBits: reduce while the polynomial is longer than 2^final_poly_bits and the
    // next tree still has at least 2^cap_height leaves.
    std::vector<uint64_t> out;
    uint64_t d = degree_bits;
    while (d > final_poly_bits && d + rate_bits >= cap_height + arity_bits) {
        out.push_back(arity_bits);
        if (d < arity_bits) { out.pop_back(); break; }
        d -= arity_bits;
    }
    return out;
}
