"""tests/golden/pack_small.bin — the byte-level target of the Rust circuit-pack exporter (integration/qpgpu_backend.rs, never
compiled here: no Rust toolchain) — and tools/pack_dump.py, the field-by-field dump a maintainer diffs the exporter's output
against (reference hook: wormhole/circuit-builder/src/lib.rs:37-110 writes the other artifacts of a circuit the same way).
The golden file must stay what the generator produces, load through every parser of the repo, and prove."""
import os
import sys

import numpy as np
import pytest

from oracle_binding import OracleCircuit

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, GOLD)


def golden():
    return np.fromfile(os.path.join(GOLD, "pack_small.bin"), dtype="<u8")


def test_golden_pack_is_what_the_generator_makes_and_dumps_identically(pkg):
    from gen_pack_small import PARAMS
    from pack_dump import dump
    pack, wires, pis = pkg.synth_circuit(**PARAMS)
    g = golden()
    assert g.size == pack.size and np.array_equal(g, pack), "tests/golden/pack_small.bin is stale: python tests/golden/gen_pack_small.py"
    text = dump(g)
    assert text == open(os.path.join(GOLD, "pack_small.dump.txt")).read()
    # what a maintainer should see named in the dump
    for needle in ("header.num_selectors = 4", "gate[13] = Poseidon2(", "gate[12] = CosetInterpolation(param0=4, param1=6", "trailer HINT1 count=",
                   "trailer PUBI1 count=21", "trailer P2GL1 (Poseidon2 gate wire layout): w_input=0, w_output=12, w_swap=24", "copy_classes = "):
        assert needle in text, needle
    # a one-word change shows up as a one-line diff that names the field
    bad = g.copy(); bad[11] = 5                      # cap_height
    d = [a for a, b in zip(dump(bad).split("\n"), text.split("\n")) if a != b]
    assert d == ["header.cap_height = 5"]


def test_golden_pack_loads_and_proves_on_the_cpu_side(pkg, orc):
    from gen_pack_small import PARAMS
    g = golden()
    _, wires, pis = pkg.synth_circuit(**PARAMS)
    oc = OracleCircuit(orc, g); ver = pkg.Verifier(g)
    try:
        proof = oc.prove(wires, pis)
        assert oc.verify(proof) == 0 and ver.verify(proof)
    finally:
        oc.close(); ver.close()


@pytest.mark.gpu
def test_golden_pack_on_the_gpu(pkg, gpu, orc):
    from gen_pack_small import PARAMS
    g = golden()
    _, wires, pis = pkg.synth_circuit(**PARAMS)
    circ = pkg.Circuit(gpu, g); oc = OracleCircuit(orc, g)
    try:
        mask = circ.witness_free_mask(*wires.shape)
        assert np.array_equal(circ.generate_witness(np.where(mask == 1, wires, 0).astype(np.uint64), pis), wires)
        proof = circ.prove(wires, pis)
        assert proof == oc.prove(wires, pis) and oc.verify(proof) == 0
    finally:
        circ.close(); oc.close()
