"""The qp fork's Poseidon2 gate (gate type 14) on the GPU: quotient_poseidon2_kernel (stage s6), the lane-cooperative row
generator (stage s1) and the witness check, against the oracle's separate restatement (oracle/poseidon2_gate.c) and both
verifiers — single proofs, lockstep batches, zero knowledge, the default wire layout and a different one read from the pack's
layout table, and the exact shape bench.py times. LAYOUT UNPINNED (see tests/test_poseidon2_gate.py); the permutation is the
KAT-pinned one, which test_generated_digests_equal_the_kat_pinned_hash ties to the generator's output."""
import numpy as np
import pytest

from oracle_binding import OracleCircuit
from test_poseidon2_gate import KW, P, check_sites


@pytest.mark.gpu
@pytest.mark.parametrize("alt", [False, True])
def test_proof_bytes_with_poseidon2_rows_equal_the_oracle(pkg, gpu, orc, alt):
    for d, kw, zk in ((7, dict(seed=21), False), (8, dict(seed=22, ext_arith=True, recursion=True, hints=True), True)):
        pack, wires, pis = pkg.synth_circuit(d, p2_alt_layout=alt, **KW, **kw)
        if zk:
            pack[14] = 1
        circ = pkg.Circuit(gpu, pack, max_batch=4); oc = OracleCircuit(orc, pack); ver = pkg.Verifier(pack, circuit=circ)
        try:
            circ.set_blinding_seed(5)
            got = circ.prove(wires, pis)
            assert got == oc.prove(wires, pis, seed=5)
            assert oc.verify(got) == 0 and ver.verify(got)
            # lockstep batch of four different witnesses (other public inputs; the Poseidon2 rows keep their preimages)
            mask = circ.witness_free_mask(*wires.shape)
            ws, ps = [], []
            for b in range(4):
                p_b = (pis + np.uint64(b)) % np.uint64(P)
                part = np.where(mask == 1, wires, 0).astype(np.uint64)
                ws.append(circ.generate_witness(part, p_b)); ps.append(p_b)
            d_w = gpu.to_device(np.stack(ws))
            circ.set_blinding_seed(40)
            batch = circ.prove_batch_dev([d_w.ptr + b * wires.nbytes for b in range(4)], ps)
            d_w.free(scrub=True)
            for b in range(4):
                assert batch[b] == oc.prove(ws[b], ps[b], seed=40 + b), b
                assert oc.verify(batch[b]) == 0 and ver.verify(batch[b])
        finally:
            ver.close(); circ.close(); oc.close()


@pytest.mark.gpu
@pytest.mark.parametrize("alt", [False, True])
def test_generated_digests_equal_the_kat_pinned_hash(pkg, gpu, alt):
    """Stage s1 from the free cells only: the Poseidon2 row generator must reproduce the full witness, and the digests it
    writes must equal qpgpu_poseidon2_hash_pad10 (host, held to the reference's vectors) and the device sponge over the same
    preimages — after the preimages have been CHANGED, so nothing can come from the synthetic generator's own values."""
    d = 8
    pack, wires, pis = pkg.synth_circuit(d, seed=23, p2_alt_layout=alt, **KW)
    circ = pkg.Circuit(gpu, pack)
    try:
        mask = circ.witness_free_mask(*wires.shape)
        part = np.where(mask == 1, wires, 0).astype(np.uint64)
        full = circ.generate_witness(part.copy(), pis)
        assert np.array_equal(full, wires)
        sites = pkg.synth_p2_sites(d, 21, poseidon2=True, p2_alt_layout=alt)
        changed = 0
        for site in sites:
            for c, r in pkg.p2_site_cells(pack, site)[0]:
                if mask[c, r]:
                    part[c, r] = np.uint64((int(part[c, r]) * 3 + 17) % P); changed += 1
        assert changed > 20
        full2 = circ.generate_witness(part.copy(), pis)
        assert not np.array_equal(full2, wires)
        check_sites(pkg, pack, full2, d, alt=alt)
        site = sites[5]                                           # the 45-element (block-header-sized) preimage, on the device sponge too
        pre, dig = pkg.p2_site_cells(pack, site)
        dev = gpu.poseidon2_hash_pad10(np.array([full2[c, r] for c, r in pre], dtype=np.uint64))
        assert dev[0].tolist() == [int(full2[c, r]) for c, r in dig]
        # the changed witness still satisfies the circuit: the witness check passes and the proof verifies
        circ.set_witness_check(True)
        ver = pkg.Verifier(pack, circuit=circ)
        assert ver.verify(circ.prove(full2, pis))
        ver.close()
    finally:
        circ.close()


@pytest.mark.gpu
def test_witness_check_names_a_broken_poseidon2_row(pkg, gpu):
    pack, wires, pis = pkg.synth_circuit(7, seed=24, **KW)
    lay = pkg.pack_p2_layout(pack)
    circ = pkg.Circuit(gpu, pack)
    try:
        circ.set_witness_check(True)
        circ.prove(wires, pis)
        row = 8 * pkg.synth_p2_sites(7, 21, poseidon2=True)[0][2] + 3
        for col in (lay["w_partial"] + 7, lay["w_full1"] + 30, lay["w_delta"] + 1):
            bad = wires.copy(); bad[col, row] = (int(bad[col, row]) + 1) % P
            with pytest.raises(pkg.QpGpuError) as e:
                circ.prove(bad, pis)
            assert e.value.code == -4 and f"row {row}" in str(e.value)
    finally:
        circ.close()


@pytest.mark.gpu
def test_witness_generation_schedules_agree_on_poseidon2_rows(pkg, gpu, monkeypatch):
    """Both launch schedules of stage s1 (one launch per level; runs of narrow levels in one launch) drive the lane-cooperative
    Poseidon2 generator, also in a batch and next to PoseidonGate rows of the same level."""
    pack, wires, pis = pkg.synth_circuit(8, seed=25, ext_arith=True, recursion=True, hints=True, **KW)
    for fuse in ("0", "1"):
        monkeypatch.setenv("QPGPU_WITNESS_FUSE", fuse)
        circ = pkg.Circuit(gpu, pack, max_batch=3)
        try:
            mask = circ.witness_free_mask(*wires.shape)
            part = np.where(mask == 1, wires, 0).astype(np.uint64)
            assert np.array_equal(circ.generate_witness(part.copy(), pis), wires), fuse
            d_w = gpu.to_device(np.stack([part] * 3))
            circ.generate_witness_dev(d_w, np.stack([pis] * 3), batch=3)
            out = d_w.download().reshape(3, *wires.shape)
            d_w.free(scrub=True)
            assert all(np.array_equal(out[b], wires) for b in range(3)), fuse
        finally:
            circ.close()


@pytest.mark.gpu
def test_leaf_profile_bench_shape_byte_parity(pkg, gpu, orc):
    """The circuit bench.py times from round 3 on: 2^13 rows x 135 wires, 80 routed, PoseidonGate + BaseSum rows and the leaf
    profile's 61 Poseidon2-gate rows (+ 4 free-standing), 21 public inputs."""
    pack, wires, pis = pkg.synth_circuit(13, num_wires=135, num_routed=80, num_public_inputs=21, seed=1000, **KW)
    check_sites(pkg, pack, wires, 13)
    circ = pkg.Circuit(gpu, pack, max_batch=2); oc = OracleCircuit(orc, pack)
    try:
        mask = circ.witness_free_mask(*wires.shape)
        assert np.array_equal(circ.generate_witness(np.where(mask == 1, wires, 0).astype(np.uint64), pis), wires)
        d_w = gpu.to_device(np.stack([wires, wires]))
        got = circ.prove_batch_dev([d_w.ptr, d_w.ptr + wires.nbytes], [pis, pis])
        d_w.free(scrub=True)
        orc.set_threads(16)
        want = oc.prove(wires, pis)
        assert got[0] == want and got[1] == want and oc.verify(got[0]) == 0
    finally:
        circ.close(); oc.close()
