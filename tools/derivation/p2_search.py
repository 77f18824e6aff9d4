#!/usr/bin/env python3
"""Timeboxed search for qp-poseidon-core 3.1.0's Poseidon2 parameters (not available offline).

A candidate = (round-constant family) x (how the family is laid out over external / internal rounds) x (external 4x4 block)
x (internal diagonal convention). Every candidate is scored against the reference's seven known-answer vectors
(tests/golden/poseidon2_kats.json: 5 addresses H(H(felts("wormhole") || secret)), 2 block hashes over 45 elements) through
the product's own sponge (libqpgpu: qpgpu_leaf_unspendable_account / qpgpu_leaf_block_hash), so a hit is a hit for the code
that ships. The first address KAT is the filter; a candidate that passes it is scored on all seven.

Families enumerated (published ways of producing Poseidon2 Goldilocks t=12, R_F=8, R_P=22, x^7 constants):
  grain      HorizenLabs reference generator (Grain LFSR; first constant 0x13dcf33aba214f46), full vectors for all 30 rounds,
             the partial rounds keeping element 0 (the zkhash / poseidon2 reference instance `poseidon2_goldilocks_12`)
  xoroshiro  Plonky3 `Poseidon2::new_from_rng_128` with rand_xoshiro Xoroshiro128Plus::seed_from_u64(s) (SplitMix64 seeding),
             s in the documented test seeds and a range of small integers; elements drawn by rejection (`rng.gen::<Goldilocks>()`)
  chacha     the same constructor on rand_chacha ChaCha{8,12,20}Rng::seed_from_u64(s) (rand_core PCG32 seeding)
Layout options: p3 (all external constants first: 8 x 12, then the 22 internal) / seq (in round order: 4 x 12, 22, 4 x 12).
External block: HorizenLabs M4 (5 7 1 3 / 4 6 1 1 / 1 3 5 7 / 1 1 4 6) or Plonky3 MDSMat4 circ(2 3 1 1).
Internal matrix: J + diag(d) with d = Plonky3's MATRIX_DIAG_12_GOLDILOCKS, or d - 1 (the "M - 1" reading of the same table).

usage: p2_search.py [--seeds N] [--minutes M]    -> appends to tools/derivation/p2_search_results.md.

OUTCOME (round 2): chacha20 / seed 0x3141592653589793 / layout p3 / M4 P3 / diagonal d reproduces the five address vectors;
the two block-header vectors (45 elements, six blocks) then showed that the sponge ADDS each padded block into the state
(one-block inputs cannot tell that from overwriting). With additive absorption all seven vectors pass; the set is built into
the product (csrc/poseidon_constants.cpp: poseidon2::qp_params) and re-derived independently by the oracle
(oracle/poseidon2.c: orc_p2_qp_params).
"""
import argparse, ctypes, itertools, json, os, sys, time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
import numpy as np
import __graft_entry__ as ge
from chacha import ChaChaRng
from grain import consts as grain_consts

P = 0xFFFFFFFF00000001
M64 = (1 << 64) - 1
KATS = json.load(open(os.path.join(ROOT, "tests", "golden", "poseidon2_kats.json")))
M4_HL = [5, 7, 1, 3, 4, 6, 1, 1, 1, 3, 5, 7, 1, 1, 4, 6]
M4_P3 = [2, 3, 1, 1, 1, 2, 3, 1, 1, 1, 2, 3, 3, 1, 1, 2]
DIAG_P3 = [0xc3b6c08e23ba9300, 0xd84b5de94a324fb6, 0x0d0c371c5b35b84f, 0x7964f570e7188037, 0x5daf18bbd996604b, 0x6743bc47b9595257,
           0x5528b9362c59bb70, 0xac45e25b7127b68b, 0xa2077d7dfbb606b5, 0xf3faac6faee378ae, 0x0c6388b51545e883, 0xd27dbb6944917b60]


class Xoroshiro128Plus:
    """rand_xoshiro::Xoroshiro128Plus; seed_from_u64 fills the state with SplitMix64 (rand_core's override in rand_xoshiro)."""
    def __init__(self, seed):
        def splitmix():
            nonlocal seed
            seed = (seed + 0x9E3779B97F4A7C15) & M64
            z = seed
            z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
            z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
            return z ^ (z >> 31)
        self.s0, self.s1 = splitmix(), splitmix()
    def next_u64(self):
        r = (self.s0 + self.s1) & M64
        s1 = self.s1 ^ self.s0
        self.s0 = (((self.s0 << 24) | (self.s0 >> 40)) & M64) ^ s1 ^ ((s1 << 16) & M64)
        self.s1 = ((s1 << 37) | (s1 >> 27)) & M64
        return r


def draw(rng, count):
    out = []
    while len(out) < count:
        v = rng.next_u64()
        if v < P:
            out.append(v)
    return out


def families(seeds):
    g = grain_consts(12, 8, 22, 360)
    yield "grain", g[:48] + g[48 + 22 * 12:], [g[48 + 12 * r] for r in range(22)], "fixed"
    for s in seeds:
        yield f"xoroshiro128+ seed {s}", None, None, ("rng", lambda s=s: Xoroshiro128Plus(s))
    for rounds in (8, 12, 20):
        for s in seeds:
            yield f"chacha{rounds} seed {s}", None, None, ("rng", lambda s=s, r=rounds: ChaChaRng(s, r))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=512)
    ap.add_argument("--minutes", type=float, default=15.0)
    args = ap.parse_args()
    pkg = ge.load_package()
    L = pkg.load_library()
    L.qpgpu_leaf_unspendable_account.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_void_p]
    L.qpgpu_leaf_block_hash.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_uint32, ctypes.c_char_p, ctypes.c_char_p,
                                        ctypes.c_char_p, ctypes.c_char_p, ctypes.c_void_p]
    digest = bytes.fromhex(KATS["digest_hex_head"]) + bytes(KATS["digest_zero_run"]) + bytes.fromhex(KATS["digest_hex_tail"])
    out = ctypes.create_string_buffer(32)

    def score(block, full):
        ok = 0
        for k in KATS["address_kats"]:
            L.qpgpu_leaf_unspendable_account(block.ctypes.data, block.size, bytes.fromhex(k["secret"]), out)
            hit = out.raw.hex() == k["address"]
            ok += hit
            if not full and not hit:
                return ok
        for k in KATS["block_header_kats"]:
            parent = bytes.fromhex(k["parent_hash"]) if "parent_hash" in k else bytes(k["parent_hash_bytes"])
            L.qpgpu_leaf_block_hash(block.ctypes.data, block.size, parent, k["block_number"], bytes.fromhex(k["state_root"]),
                                    bytes.fromhex(k["extrinsics_root"]), bytes.fromhex(k["zk_tree_root"]), digest, out)
            ok += out.raw == bytes(k["expected_hash_bytes"])
        return ok

    notable = [0, 1, 2, 3, 7, 42, 1337, 12345, 2023, 2024, 2025, 0xdeadbeef, 0x189189189189189, 0x3141592653589793, 31415926535, 271828182845,
               0x5eed, 0x5EED5EED, 0xC0FFEE, 0x517cc1b727220a95, 0x9E3779B97F4A7C15]
    seeds = sorted(set(list(range(args.seeds)) + notable))
    t0 = time.time()
    tried, best, hits = 0, 0, []
    per_family = {}
    for name, ext, internal, how in families(seeds):
        if time.time() - t0 > args.minutes * 60:
            break
        layouts = [("fixed", ext, internal)] if how == "fixed" else []
        if how != "fixed":
            v = draw(how[1](), 96 + 22)
            layouts = [("p3", v[:96], v[96:]), ("seq", v[:48] + v[70:], v[48:70])]
        fam = name.split(" seed")[0]
        for (lay, e, i), m4, dconv in itertools.product(layouts, (("HL", M4_HL), ("P3", M4_P3)), ("d", "d-1")):
            diag = DIAG_P3 if dconv == "d" else [(x - 1) % P for x in DIAG_P3]
            block = np.array(list(e) + list(i) + diag + m4[1], dtype=np.uint64)
            s = score(block, False)
            if s >= 1:
                s = score(block, True)
                hits.append((name, lay, m4[0], dconv, s))
            best = max(best, s)
            tried += 1
            per_family[fam] = per_family.get(fam, 0) + 1
    dt = time.time() - t0
    lines = [f"\n## run {time.strftime('%Y-%m-%d %H:%M:%S')} ({dt / 60:.1f} min, {tried} candidates, seeds 0..{args.seeds - 1} + {len(notable)} notable)\n",
             "| family | candidates | layouts x M4 x diagonal | best score /7 |", "|---|---|---|---|"]
    for fam, cnt in per_family.items():
        fam_best = max([h[4] for h in hits if h[0].startswith(fam)] + [0])
        lines.append(f"| {fam} | {cnt} | {'fixed' if fam == 'grain' else 'p3, seq'} x HL, P3 x d, d-1 | {fam_best} |")
    top = [h for h in hits if h[4] == best and best > 0]
    lines.append(f"\nResult: best {best}/7" + (f" by {top}" if top else " (no candidate reproduces any vector)") + ("; PINNED." if best == 7 else "."))
    open(os.path.join(HERE, "p2_search_results.md"), "a").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
