#!/usr/bin/env python3
"""Regenerates tests/golden/pack_small.bin and pack_small.dump.txt: the golden circuit pack the Rust exporter's first compile
converges to (tools/pack_dump.py, integration/README.md). A 2^6-row circuit with every gate type the backend knows — the
fourteen of plonky2's standard recursion circuits and the Poseidon2 gate —, three selector polynomials, free-standing
generators (hint trailer), the public-input cells and the Poseidon2 wire-layout table.
usage: python tests/golden/gen_pack_small.py   (needs the built library; no GPU)"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import __graft_entry__ as ge  # noqa: E402
from pack_dump import dump  # noqa: E402

PARAMS = dict(degree_bits=6, num_wires=135, num_routed=80, num_public_inputs=21, seed=2026, poseidon=True, base_sum=True, ext_arith=True, recursion=True,
              hints=True, poseidon2=True)

if __name__ == "__main__":
    pkg = ge.load_package()
    pack, wires, pis = pkg.synth_circuit(**PARAMS)
    pack.astype("<u8").tofile(os.path.join(HERE, "pack_small.bin"))
    open(os.path.join(HERE, "pack_small.dump.txt"), "w").write(dump(pack))
    print("pack_small.bin:", pack.size, "words")
