#!/usr/bin/env python3
"""Field-by-field text dump of a circuit pack ("QPCP1", qp-zk-circuits_amd/csrc/circuit.hpp), made for diffing.

Why it exists: the Rust exporter (integration/qpgpu_backend.rs) has never been compiled — this image has no Rust toolchain —
so its first run on a maintainer's machine needs something to converge to. Procedure:

  1. build the golden circuit on the Rust side (integration/README.md: "golden pack": the builder calls that reproduce
     tests/golden/pack_small.bin's gate rows) and export it with the Rust writer: `pack_small.rust.bin`;
  2. python tools/pack_dump.py tests/golden/pack_small.bin > a.txt; python tools/pack_dump.py pack_small.rust.bin > b.txt;
  3. `diff a.txt b.txt` must be empty. The dump names every header word, every gate (type, parameters, selector group,
     constraint count), which gate each row selects, the constants of every row, the copy classes decoded from the sigma
     polynomials, the hint and public-input trailers and the Poseidon2 gate's wire-layout table — so a difference points at the
     field the writer got wrong, not at a byte offset.

Only what depends on plonky2's builder heuristics may legitimately differ for a hand-written circuit (the order of cells inside
a copy class is normalised here: classes are printed sorted, with their members sorted).

usage: pack_dump.py <pack.bin> [--no-sigmas] [--rows N]     (pack file = little-endian u64 words)"""
import sys

import numpy as np

P = 0xFFFFFFFF00000001
ROOT_2_32 = 7277203076849721926
GATE_NAMES = ["Noop", "Constant", "PublicInput", "Arithmetic", "Poseidon", "BaseSum", "ArithmeticExtension", "MulExtension", "Reducing",
              "ReducingExtension", "RandomAccess", "Exponentiation", "PoseidonMds", "CosetInterpolation", "Poseidon2"]
HEADER = ["degree_bits", "num_wires", "num_routed_wires", "num_constants", "num_selectors", "num_challenges", "quotient_degree_factor",
          "num_partial_products", "num_public_inputs", "rate_bits", "cap_height", "proof_of_work_bits", "num_query_rounds", "zero_knowledge",
          "num_gate_constraints", "num_gates", "num_arity_rounds"]
HINTS = {1: "Copy", 2: "Equality", 3: "WireSplit", 4: "QuotientExtension", 5: "Constant", 6: "NonzeroTest", 7: "LowHigh"}
P2_FIELDS = ("w_input", "w_output", "w_swap", "w_delta", "w_full0", "w_partial", "w_full1", "first_round_wires", "constraint_order", "end_wire")


def dump(words, sigmas=True, max_rows=None):
    w = [int(x) for x in words]
    out = []
    if not w or w[0] != 0x3150435051:
        raise SystemExit("not a circuit pack: bad magic")
    h = dict(zip(HEADER, w[1:18]))
    for k in HEADER:
        out.append(f"header.{k} = {h[k]}")
    pos = 18
    out.append("fri.reduction_arity_bits = " + str(w[pos:pos + h["num_arity_rounds"]]))
    pos += h["num_arity_rounds"]
    gates = []
    for i in range(h["num_gates"]):
        t, p0, p1, sel, g0, g1, nc, p2 = w[pos:pos + 8]
        pos += 8
        gates.append((t, p0, p1, sel, g0, g1, nc, p2))
        name = GATE_NAMES[t] if t < len(GATE_NAMES) else f"type{t}"
        out.append(f"gate[{i}] = {name}(param0={p0}, param1={p1}, param2={p2}) selector_polynomial={sel} group=[{g0},{g1}) constraints={nc}")
    R, n = h["num_routed_wires"], 1 << h["degree_bits"]
    k_is = w[pos:pos + R]
    pos += R
    out.append("k_is[0..4] = " + str(k_is[:4]) + f" ... ({R} cosets; k_is[j] = g^j with g = 14293326489335486720: {all(k_is[j] == pow(14293326489335486720, j, P) for j in range(R))})")
    out.append("circuit_digest = " + str(w[pos:pos + 4]))
    pos += 4
    ncs = h["num_selectors"] + h["num_constants"] + R
    cs = np.array(w[pos:pos + ncs * n], dtype=np.uint64).reshape(ncs, n)
    pos += ncs * n
    rows = n if max_rows is None else min(n, max_rows)
    for r in range(rows):
        sel = [int(cs[s, r]) for s in range(h["num_selectors"])]
        gi = next((x for x in sel if x != 0xFFFFFFFF), None)
        gname = GATE_NAMES[gates[gi][0]] if gi is not None and gi < len(gates) and gates[gi][0] < len(GATE_NAMES) else "?"
        consts = [int(cs[h["num_selectors"] + c, r]) for c in range(h["num_constants"])]
        out.append(f"row[{r}] selectors={sel} gate={gname} constants={consts}")
    if sigmas:
        # sigma(row, col) = k_is[col'] * w^row' : decode every value, then print the copy classes (cycles), normalised
        wn = pow(ROOT_2_32, 1 << (32 - h["degree_bits"]), P)
        where = {}
        for c in range(R):
            a = k_is[c]
            for r in range(n):
                where[a] = (r, c)
                a = a * wn % P
        nxt = {}
        sig0 = h["num_selectors"] + h["num_constants"]
        for c in range(R):
            for r in range(n):
                v = int(cs[sig0 + c, r])
                if v not in where:
                    raise SystemExit(f"sigma value at row {r}, wire {c} is not k_is[j] * w^i")
                nxt[(r, c)] = where[v]
        seen, classes = set(), []
        for cell in sorted(nxt):
            if cell in seen:
                continue
            cyc, x = [], cell
            while x not in seen:
                seen.add(x); cyc.append(x); x = nxt[x]
            if len(cyc) > 1:
                classes.append(sorted(cyc))
        out.append(f"copy_classes = {len(classes)} (cells as row.wire; singletons omitted)")
        for cyc in sorted(classes):
            if max_rows is None or cyc[0][0] < max_rows:
                out.append("  class " + " ".join(f"{r}.{c}" for r, c in cyc))
    while pos + 2 <= len(w):
        magic, cnt = w[pos], w[pos + 1]
        pos += 2
        if magic == 0x31544E4948:
            out.append(f"trailer HINT1 count={cnt}")
            for i in range(cnt):
                hw = w[pos:pos + 8]
                pos += 8
                out.append(f"  hint[{i}] = {HINTS.get(hw[0], hw[0])} args={hw[1:7]}")
        elif magic == 0x3149425550:
            cells = w[pos:pos + cnt]
            pos += cnt
            out.append(f"trailer PUBI1 count={cnt} cells(row.wire) = " + " ".join(f"{c // h['num_wires']}.{c % h['num_wires']}" for c in cells))
        elif magic == 0x314C473250:
            vals = w[pos:pos + cnt]
            pos += cnt
            out.append("trailer P2GL1 (Poseidon2 gate wire layout): " + ", ".join(f"{k}={'none' if k == 'w_swap' and v == 0xFFFFFFFF else v}" for k, v in zip(P2_FIELDS, vals)))
        else:
            out.append(f"UNKNOWN TRAILER magic=0x{magic:x} at word {pos - 2}")
            break
    if pos != len(w):
        out.append(f"TRAILING WORDS: {len(w) - pos}")
    return "\n".join(out) + "\n"


if __name__ == "__main__":
    if len(sys.argv) < 2:
        raise SystemExit(__doc__)
    rows = None
    if "--rows" in sys.argv:
        rows = int(sys.argv[sys.argv.index("--rows") + 1])
    sys.stdout.write(dump(np.fromfile(sys.argv[1], dtype="<u8"), sigmas="--no-sigmas" not in sys.argv, max_rows=rows))
