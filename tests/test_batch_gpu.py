"""Lockstep batches (qpgpu_prove_batch_dev): B proofs of one circuit with different witnesses, every stage launched once for
all of them. Each proof must be byte-equal to the CPU oracle's proof of the same witness (and therefore to the single-proof
entry), for dense and scattered witness layouts, with zero-knowledge salts, through the batched proving pool, and on the
exact shape bench.py times."""
import ctypes

import numpy as np
import pytest

from oracle_binding import OracleCircuit


def _witnesses(pkg, gpu, pack, wires, count):
    """`count` different satisfied witnesses of one circuit: the template's free cells with different public inputs, completed
    by stage s1 on the device. Returns (list of public inputs, list of host witnesses)."""
    agg = pkg.aggregation
    tp = agg.TemplateProver(gpu, pack, wires)
    npis = pkg.pack_header(pack)["num_public_inputs"]
    out_p, out_w = [], []
    for i in range(count):
        pis = tp.commit(agg.leaf_public_inputs(i, npis))
        out_p.append(pis.copy()); out_w.append(tp.witness().copy())
    tp.close()
    return out_p, out_w


@pytest.mark.gpu
@pytest.mark.parametrize("batch", [4, 16])
def test_batch_equals_oracle_per_proof(pkg, gpu, orc, batch):
    pack, wires, _ = pkg.synth_circuit(9, num_wires=135, num_routed=80, num_public_inputs=21, seed=300 + batch, poseidon=True,
                                       base_sum=True, ext_arith=True, recursion=True)
    pis, ws = _witnesses(pkg, gpu, pack, wires, batch)
    circ = pkg.Circuit(gpu, pack, max_batch=batch)
    oc = OracleCircuit(orc, pack)
    try:
        dense = gpu.to_device(np.stack(ws))                    # [B][num_wires][n] back to back: used in place
        mat = ws[0].size * 8
        got = circ.prove_batch_dev([dense.ptr + b * mat for b in range(batch)], pis)
        assert len(set(got)) == batch
        for b in range(batch):
            assert got[b] == oc.prove(ws[b], pis[b]), b
            assert oc.verify(got[b]) == 0
        # scattered witnesses (separate allocations, reversed order) are gathered into the workspace: same bytes
        bufs = [gpu.to_device(w) for w in ws]
        order = list(range(batch))[::-1]
        got2 = circ.prove_batch_dev([bufs[b] for b in order], [pis[b] for b in order])
        assert got2 == [got[b] for b in order]
        # a partial batch and the single-proof entry on the same handle
        got3 = circ.prove_batch_dev([bufs[1], bufs[0]], [pis[1], pis[0]])
        assert got3 == [got[1], got[0]]
        assert circ.prove_dev(bufs[2], pis[2]) == got[2]
        with pytest.raises(pkg.QpGpuError):
            circ.prove_batch_dev([bufs[0]] * (batch + 1), [pis[0]] * (batch + 1))
        for x in bufs:
            x.free()
        dense.free()
    finally:
        circ.close(); oc.close()


@pytest.mark.gpu
def test_zero_knowledge_batch_seeded(pkg, gpu, orc):
    """Salted oracles: proof b of a seeded batch uses the key derived from seed + b; unseeded batches draw fresh keys."""
    pack, wires, _ = pkg.synth_circuit(8, num_wires=135, num_routed=60, num_public_inputs=5, seed=311, poseidon=True, base_sum=True)
    pack[14] = 1
    pis, ws = _witnesses(pkg, gpu, pack, wires, 4)
    circ = pkg.Circuit(gpu, pack, max_batch=4)
    oc = OracleCircuit(orc, pack)
    try:
        bufs = [gpu.to_device(w) for w in ws]
        circ.set_blinding_seed(900)
        got = circ.prove_batch_dev(bufs, pis)
        for b in range(4):
            assert got[b] == oc.prove(ws[b], pis[b], seed=900 + b), b
            assert oc.verify(got[b]) == 0
        fresh1 = circ.prove_batch_dev(bufs, pis)
        fresh2 = circ.prove_batch_dev(bufs, pis)
        for b in range(4):
            assert oc.verify(fresh1[b]) == 0 and fresh1[b] != fresh2[b] and fresh1[b] != got[b]
        # salts of one proof say nothing about another's: same witness twice in one unseeded batch, different openings
        twice = circ.prove_batch_dev([bufs[0], bufs[0]], [pis[0], pis[0]])
        assert twice[0] != twice[1] and oc.verify(twice[0]) == 0 and oc.verify(twice[1]) == 0
        for x in bufs:
            x.free()
    finally:
        circ.close(); oc.close()


@pytest.mark.gpu
def test_batched_pool(pkg, gpu, orc):
    """qpgpu_pool_create_batched: two workers of four; tickets waited out of order; every proof equals the oracle's."""
    pack, wires, _ = pkg.synth_circuit(8, num_wires=135, num_routed=80, num_public_inputs=21, seed=320, poseidon=True, base_sum=True)
    pis, ws = _witnesses(pkg, gpu, pack, wires, 6)
    pool = pkg.ProvingPool(pack, workers=2, device=0, max_batch=4)
    oc = OracleCircuit(orc, pack)
    try:
        bufs = [gpu.to_device(w) for w in ws]
        tickets = [pool.submit(bufs[i % 6], pis[i % 6]) for i in range(22)]
        res = {}
        for t in tickets[::-1]:
            res[t] = pool.wait(t)
        want = [oc.prove(ws[i], pis[i]) for i in range(6)]
        for k, t in enumerate(tickets):
            assert res[t] == want[k % 6], k
        for x in bufs:
            x.free()
    finally:
        pool.close(); oc.close()


@pytest.mark.gpu
def test_pool_failure_concerns_only_the_failing_job(pkg, gpu, orc):
    """Unrelated callers' jobs share a lockstep batch. In the reference a failing prove affects its caller only
    (every proof is its own `prove` call, wormhole/prover/src/lib.rs:171-175): one unsatisfied witness among valid ones must
    fail alone, with its own message, and the others must come out byte-equal to the oracle's proofs. Bad arguments are
    refused at submit, before they can reach a batch."""
    pack, wires, _ = pkg.synth_circuit(8, num_wires=135, num_routed=80, num_public_inputs=21, seed=321, poseidon=True, base_sum=True)
    pis, ws = _witnesses(pkg, gpu, pack, wires, 4)
    pool = pkg.ProvingPool(pack, workers=1, device=0, max_batch=8)
    oc = OracleCircuit(orc, pack)
    try:
        pool.set_witness_check(True)
        bad = ws[2].copy()
        bad[3, 40] = (int(bad[3, 40]) + 1) % pkg.P            # an arithmetic output no longer matches its operands
        bufs = [gpu.to_device(w) for w in (ws[0], ws[1], bad, ws[3])]
        # queued together: one worker takes all six as one lockstep batch
        order = [0, 1, 2, 3, 2, 0]
        tickets = [pool.submit(bufs[i], pis[i]) for i in order]
        want = [oc.prove(ws[i], pis[i]) for i in range(4)]
        for i, t in zip(order, tickets):
            if i == 2:
                with pytest.raises(pkg.QpGpuError) as e:
                    pool.wait(t)
                assert e.value.code == -4 and "row" in str(e.value) and "of the batch" not in str(e.value)
            else:
                assert pool.wait(t) == want[i], i
        # argument checks happen per job, at submit
        small = np.empty(pool.proof_size() - 1, dtype=np.uint8)
        with pytest.raises(pkg.QpGpuError) as e:
            pool.submit(bufs[0], pis[0], small)
        assert e.value.code == -5
        t = ctypes.c_uint64()
        out = np.empty(pool.proof_size(), dtype=np.uint8)
        assert pool.lib.qpgpu_pool_submit(pool.h, bufs[0].ptr, None, out.ctypes.data, out.size, ctypes.byref(t)) == -1   # no public inputs
        # the pool still works afterwards
        assert pool.wait(pool.submit(bufs[3], pis[3])) == want[3]
        for x in bufs:
            x.free()
    finally:
        pool.close(); oc.close()


@pytest.mark.gpu
def test_bench_shape_byte_parity(pkg, gpu, orc):
    """The exact circuit bench.py times (2^13 rows, 135 wires, 80 routed, Poseidon + BaseSum rows, 21 public inputs):
    a batch of four different witnesses, byte-equal to the oracle."""
    pack, wires, pis0 = pkg.synth_circuit(13, num_wires=135, num_routed=80, num_public_inputs=21, seed=1000, poseidon=True, base_sum=True)
    pis, ws = _witnesses(pkg, gpu, pack, wires, 3)
    pis.append(pis0); ws.append(wires)                         # the generator's own witness too
    circ = pkg.Circuit(gpu, pack, max_batch=4)
    oc = OracleCircuit(orc, pack)
    try:
        dense = gpu.to_device(np.stack(ws))
        mat = ws[0].size * 8
        got = circ.prove_batch_dev([dense.ptr + b * mat for b in range(4)], pis)
        orc.set_threads(16)
        for b in range(4):
            assert got[b] == oc.prove(ws[b], pis[b]), b
        assert oc.verify(got[3]) == 0
        dense.free()
    finally:
        circ.close(); oc.close()


@pytest.mark.gpu
def test_set_twice_with_different_values(pkg, gpu):
    """plonky2 panics "set twice with different values" when a PartialWitness assignment contradicts a generated value or
    another member of its copy class (reference wormhole/tests/src/circuit/block_header_tests.rs:34-95: a public hash that
    does not match the preimage; nullifier_tests.rs:53-58). The partial-witness entry reports it as QPGPU_EUNSAT."""
    agg = pkg.aggregation
    pack, wires, pis = pkg.synth_circuit(8, num_wires=135, num_routed=80, num_public_inputs=21, seed=330, poseidon=True, base_sum=True)
    hdr = pkg.pack_header(pack)
    nw, n = hdr["num_wires"], 1 << hdr["degree_bits"]
    tp = agg.TemplateProver(gpu, pack, wires)
    try:
        tp.commit(pis)
        assert np.array_equal(tp.witness(), wires)                     # the generator's witness is reproduced from its free cells
        pi_cells = pkg.pack_public_input_cells(pack)
        mask = tp.circ.witness_free_mask(nw, n)
        # (1) the public-input hash cells of the PublicInputGate row are outputs of the in-circuit hash: a wrong value there
        #     is the reference's "wrong block_hash public input"
        cases = [(0 * nw + 0, (int(wires[0, 0]) + 1) % pkg.P)]
        # (2) an output of a Poseidon row hashing the public inputs (row 3, wire 12): "wrong hash for this preimage"
        cases.append((3 * nw + 12, (int(wires[12, 3]) + 1) % pkg.P))
        # (3) a generated, non-routed cell (a BaseSum limb would be routed; the Poseidon S-box wires are not): row 3, wire 100
        cases.append((3 * nw + 100, (int(wires[100, 3]) + 1) % pkg.P))
        # (4) a public-input cell that disagrees with the public_inputs argument
        cases.append((int(pi_cells[5]), (int(pis[5]) + 1) % pkg.P))
        # (5) an arithmetic output (generated) in the middle of the trace
        r5 = 20
        cases.append((r5 * nw + 3, (int(wires[3, r5]) + 1) % pkg.P))
        for cell, bad in cases:
            with pytest.raises(pkg.QpGpuError) as e:
                tp.commit(pis, [cell], [bad])
            assert e.value.code == -4 and "set twice with different values" in str(e.value), (cell, str(e.value))
        # (6) the same free target assigned twice with different values (caught before anything runs)
        col, row = np.argwhere(mask == 1)[50]
        cell = int(row) * nw + int(col)
        with pytest.raises(pkg.QpGpuError) as e:
            tp.commit(pis, [cell, cell], [int(wires[col, row]), (int(wires[col, row]) + 1) % pkg.P])
        assert e.value.code == -4 and "set twice with different values" in str(e.value)
        # agreeing duplicates are fine (plonky2 accepts setting a target twice to the same value)
        tp.commit(pis, [cases[0][0], int(pi_cells[5])], [int(wires[0, 0]), int(pis[5])])
        assert np.array_equal(tp.witness(), wires)
    finally:
        tp.close()


@pytest.mark.gpu
def test_staging_fallback_path_gives_the_same_bytes(pkg, gpu, orc, monkeypatch):
    """Stager::put's synchronous route (tables that do not fit the staging ring go through the context's pinned bounce buffer):
    forced for every table by the QPGPU_STAGE_FALLBACK hook, it must produce the proofs of the ring route."""
    pack, wires, _ = pkg.synth_circuit(8, num_wires=135, num_routed=80, num_public_inputs=21, seed=340, poseidon=True, base_sum=True,
                                       ext_arith=True, recursion=True)
    pis, ws = _witnesses(pkg, gpu, pack, wires, 3)
    ring = pkg.Circuit(gpu, pack, max_batch=3)
    monkeypatch.setenv("QPGPU_STAGE_FALLBACK", "1")
    bounce = pkg.Circuit(gpu, pack, max_batch=3)          # the hook is read when the handle is created
    monkeypatch.delenv("QPGPU_STAGE_FALLBACK")
    oc = OracleCircuit(orc, pack)
    try:
        bufs = [gpu.to_device(w) for w in ws]
        a = ring.prove_batch_dev(bufs, pis)
        b = bounce.prove_batch_dev(bufs, pis)
        assert a == b and all(b[k] == oc.prove(ws[k], pis[k]) for k in range(3))
        assert bounce.prove_dev(bufs[1], pis[1]) == a[1]
        for x in bufs:
            x.free()
    finally:
        ring.close(); bounce.close(); oc.close()
