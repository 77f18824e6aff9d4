"""Stage s1 on the device (qpgpu_generate_witness*): from the caller-supplied cells alone (the PartialWitness) the
generated wire matrix must equal the generator's full witness cell for cell, satisfy every gate and copy constraint
(the witness check and the restated verifier are the reference's own acceptance criterion), and prove to the same bytes."""
import numpy as np
import pytest

from oracle_binding import OracleCircuit

pytestmark = pytest.mark.gpu

CASES = [
    dict(degree_bits=6, num_wires=24, num_routed=16, num_public_inputs=3, seed=71),
    dict(degree_bits=8, seed=72, poseidon=True, base_sum=True),
    dict(degree_bits=9, seed=73, poseidon=True, base_sum=True, ext_arith=True, recursion=True),
    dict(degree_bits=7, num_wires=80, num_routed=48, num_public_inputs=0, seed=74, recursion=True),
    dict(degree_bits=9, seed=78, poseidon=True, base_sum=True, ext_arith=True, recursion=True, hints=True),
    dict(degree_bits=6, num_wires=24, num_routed=16, num_public_inputs=2, seed=79, base_sum=True, hints=True),
]


@pytest.fixture(params=["per_level", "runs"])
def launch_mode(request, monkeypatch):
    """Both launch schedules of the generation plan: one launch per dependency level, and runs of narrow levels walked by one
    workgroup per witness (witness_run_kernel; the default only for plans of 1024 levels and more)."""
    monkeypatch.setenv("QPGPU_WITNESS_FUSE", "1" if request.param == "runs" else "0")
    return request.param


@pytest.mark.parametrize("kw", CASES)
def test_generated_witness_equals_the_full_witness(pkg, gpu, orc, kw, launch_mode):
    kw = dict(kw)
    d = kw.pop("degree_bits")
    pack, wires, pis = pkg.synth_circuit(d, **kw)
    circ = pkg.Circuit(gpu, pack)
    try:
        gens, levels, free = circ.witness_info()
        mask = circ.witness_free_mask(*wires.shape)
        assert free == int(mask.sum()) and 0 < free < wires.size and gens > 0 and levels >= 1
        rng = np.random.default_rng(1)
        partial = np.where(mask == 1, wires, rng.integers(0, 2**63, size=wires.shape, dtype=np.uint64))   # junk in every determined cell
        full = circ.generate_witness(partial, pis)
        bad = np.argwhere(full != wires)
        assert bad.size == 0, f"{len(bad)} cells differ, first (wire, row) = {tuple(bad[0])}"
        # and the result is a witness: gate + copy constraints hold, proof bytes match the CPU restatement
        circ.set_witness_check(True)
        proof = circ.prove(full, pis)
        oc = OracleCircuit(orc, pack)
        assert proof == oc.prove(wires, pis) and oc.verify(proof) == 0
        oc.close()
    finally:
        circ.close()


def test_generation_follows_the_inputs(pkg, gpu):
    """Changing one supplied cell changes what depends on it and the result is again a valid witness."""
    pack, wires, pis = pkg.synth_circuit(8, seed=75, poseidon=True, base_sum=True, ext_arith=True, recursion=True)
    circ = pkg.Circuit(gpu, pack)
    try:
        mask = circ.witness_free_mask(*wires.shape)
        # a free routed cell of an arithmetic row (row 12: rows 3.. hold the public-input hash, whose inputs are the public
        # inputs themselves; an operation input that is not a copy)
        ROW = 12
        cols = [c for c in range(0, 40) if c % 4 != 3 and mask[c, ROW]]
        assert cols, "expected at least one free arithmetic input in row 12"
        partial = wires.copy(); partial[cols[0], ROW] = (int(partial[cols[0], ROW]) + 12345) % 0xFFFFFFFF00000001
        full = circ.generate_witness(partial, pis)
        assert (full != wires).any() and (full[mask == 1] == partial[mask == 1]).all()
        # the public-input cells are taken from the public_inputs argument, whatever the matrix holds there
        pic = pkg.pack_public_input_cells(pack)
        nw = wires.shape[0]
        junk = wires.copy(); junk[int(pic[0]) % nw, int(pic[0]) // nw] ^= np.uint64(5)
        assert (circ.generate_witness(junk, pis) == wires).all()
        circ.set_witness_check(True)
        circ.prove(full, pis)                     # QPGPU_EUNSAT would raise
    finally:
        circ.close()


def test_device_resident_generation_then_prove(pkg, gpu, orc):
    pack, wires, pis = pkg.synth_circuit(10, seed=76, poseidon=True, base_sum=True)
    circ = pkg.Circuit(gpu, pack)
    try:
        mask = circ.witness_free_mask(*wires.shape)
        partial = np.where(mask == 1, wires, 0).astype(np.uint64)
        d = gpu.to_device(partial)
        circ.generate_witness_dev(d, pis)
        gpu.sync()
        assert (d.download().reshape(wires.shape) == wires).all()
        oc = OracleCircuit(orc, pack)
        assert circ.prove_dev(d, pis) == oc.prove(wires, pis)
        oc.close()
        d.free()
    finally:
        circ.close()


def test_batched_generation(pkg, gpu, launch_mode):
    """Several witnesses of one circuit in one pass: different free cells and public inputs per witness."""
    pack, wires, pis = pkg.synth_circuit(8, seed=77, poseidon=True, base_sum=True, ext_arith=True, recursion=True)
    circ = pkg.Circuit(gpu, pack)
    try:
        mask = circ.witness_free_mask(*wires.shape)
        B = 3
        partials, all_pis = [], []
        rng = np.random.default_rng(9)
        for b in range(B):
            part = np.where(mask == 1, wires, 0).astype(np.uint64)
            if b:                                      # perturb a free arithmetic input of row 12 and the public inputs
                col = next(c for c in range(40) if c % 4 != 3 and mask[c, 12])
                part[col, 12] = np.uint64(int(rng.integers(1, 2**62)))
            partials.append(part)
            all_pis.append((pis + np.uint64(b)) % np.uint64(0xFFFFFFFF00000001))
        pic = pkg.pack_public_input_cells(pack)
        pmask = np.zeros_like(mask); pmask[(pic % np.uint64(wires.shape[0])).astype(int), (pic // np.uint64(wires.shape[0])).astype(int)] = 1
        d = gpu.to_device(np.stack(partials))
        circ.generate_witness_dev(d, np.stack(all_pis), batch=B)
        gpu.sync()
        got = d.download().reshape(B, *wires.shape)
        d.free()
        assert (got[0] == wires).all()
        circ.set_witness_check(True)
        for b in range(B):
            keep = (mask == 1) & (pmask == 0)          # the public-input cells follow the public_inputs argument
            assert (got[b][keep] == partials[b][keep]).all()
            nw_ = np.uint64(wires.shape[0])
            assert (got[b][(pic % nw_).astype(int), (pic // nw_).astype(int)] == all_pis[b]).all()
            if b:
                assert (got[b] != wires).any()
                assert (got[b] == circ.generate_witness(partials[b], all_pis[b])).all()    # same as one at a time
            circ.prove(got[b], all_pis[b])             # every witness satisfies the circuit
    finally:
        circ.close()


def test_hint_trailer(pkg, gpu, orc):
    """Generators that are not attached to a gate travel in the pack's hint trailer: their output cells are produced by
    stage s1 (not caller-supplied), chains through them resolve, and the prover ignores the trailer."""
    kw = dict(seed=80, poseidon=True, base_sum=True, ext_arith=True, recursion=True)
    plain, _, _ = pkg.synth_circuit(8, **kw)
    pack, wires, pis = pkg.synth_circuit(8, hints=True, **kw)
    n, NW = 1 << 8, 135
    tail = 2 + pis.size                                 # both packs end with the public-input cell trailer
    body = plain.size - tail                            # same shape without hints: the hint trailer starts here
    assert int(pack[body]) == 0x31544E4948 and pack.size == body + 2 + 8 * int(pack[body + 1]) + tail
    hints = pack[body + 2:pack.size - tail].reshape(-1, 8)
    assert set(int(h[0]) for h in hints) == {1, 2, 3, 4, 5, 6, 7}
    circ = pkg.Circuit(gpu, pack)
    try:
        mask = circ.witness_free_mask(NW, n)
        outs = {1: [1], 2: [3, 4], 3: [2], 4: [5, 6], 5: [1], 6: [2], 7: [2, 3]}
        for h in hints:
            for k in outs[int(h[0])]:
                row, col = divmod(int(h[k]), NW)
                assert mask[col, row] == 0, (list(map(int, h)), k)
        full = circ.generate_witness(np.where(mask == 1, wires, 0).astype(np.uint64), pis)
        assert (full == wires).all()
        circ.set_witness_check(True)
        proof = circ.prove(full, pis)
        oc = OracleCircuit(orc, pack)
        assert proof == oc.prove(wires, pis)
        oc.close()
        # a truncated or foreign trailer is refused
        with pytest.raises(pkg.QpGpuError):
            pkg.Circuit(gpu, pack[:-3])
        bad = pack.copy(); bad[body + 2] = 99
        with pytest.raises(pkg.QpGpuError):
            pkg.Circuit(gpu, bad)
    finally:
        circ.close()
