"""The multi-device proving pool behind the C ABI (qpgpu_pool_create_multi, include/qpgpu.h): one queue, one worker set per
device, proofs land in the caller's host buffers. On the one-GPU box the device list is {0, 0} — two worker sets on the same
GPU — which exercises everything but the second physical device: job routing (pinned device jobs, host-witness jobs,
PartialWitness jobs), byte parity of every proof with the oracle, failure isolation, draining at destroy."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

import leaf_cases as lc
import oracle_binding as ob

pytestmark = pytest.mark.gpu


def oracle_proof(orc, circuit, x):
    cells, values, pis = circuit.commit(x)
    rc, wires, _ = orc.generate_witness(circuit.pack, cells, values, pis)
    assert rc == orc.WIT_OK
    oc = ob.OracleCircuit(orc, circuit.pack)
    proof = oc.prove(wires, pis)
    oc.close()
    return wires, proof, pis


def test_two_worker_sets_three_job_forms(pkg, gpu, orc):
    L = pkg.leaf
    full = L.LeafCircuit()
    xs = [lc.dummy_inputs(L), lc.test_inputs(L, 0), lc.real_inputs(L, depth=3), lc.test_inputs(L, 1), lc.real_inputs(L, depth=16, seed=4)]
    want = [oracle_proof(orc, full, x) for x in xs]
    pool = pkg.ProvingPool(full.pack, workers=2, max_batch=4, devices=[0, 0], host_witness=True)
    assert pool.lib.qpgpu_pool_devices(pool.h) == 2 and pool.lib.qpgpu_pool_workers(pool.h) == 4
    com = [full.commit(x) for x in xs]
    pool.set_partial_cells(com[0][0])
    # (3) PartialWitness jobs, a failing one among them: every other proof equals the oracle's, the offender alone gets the error
    bad = lc.real_inputs(L, depth=3); bad.block_hash[1] ^= 4
    cb = full.commit(bad)
    tickets = []
    for rep in range(6):
        for k, c in enumerate(com):
            tickets.append((k, pool.submit_partial(c[1], c[2])))
        if rep == 2:
            t_bad = pool.submit_partial(cb[1], cb[2])
    for k, t in tickets:
        assert pool.wait(t) == want[k][1], k
    with pytest.raises(pkg.QpGpuError) as e:
        pool.wait(t_bad)
    assert e.value.code == -4 and "set twice with different values" in str(e.value)
    # (2) full wire matrices from host memory (a patched prove() after full_witness())
    tickets = [(k, pool.submit_host(want[k][0], want[k][2])) for k in range(len(xs)) for _ in range(3)]
    for k, t in tickets:
        assert pool.wait(t) == want[k][1], k
    # (1) device-resident witnesses pinned to device index 0 or 1
    bufs = [gpu.to_device(w[0]) for w in want]
    tickets = [(k, pool.submit_on(i % 2, bufs[k], want[k][2])) for i in range(4) for k in range(len(xs))]
    for k, t in tickets:
        assert pool.wait(t) == want[k][1], k
    with pytest.raises(pkg.QpGpuError):
        pool.submit_on(2, bufs[0], want[0][2])          # no such device index
    # destroy drains both worker sets: queue work and close without waiting
    outs = [np.zeros(pool.proof_size(), dtype=np.uint8) for _ in range(12)]
    for i, o in enumerate(outs):
        pool.submit_partial(com[i % len(com)][1], com[i % len(com)][2], o)
    pool.close()
    for i, o in enumerate(outs):
        assert o.tobytes() == want[i % len(com)][1], i
    for b in bufs:
        b.free(scrub=True)


def test_pool_without_workspace_refuses_host_jobs(pkg, gpu):
    pack, wires, pis = pkg.synth_circuit(8, seed=3)
    pool = pkg.ProvingPool(pack, workers=1, max_batch=2, devices=[0])
    with pytest.raises(pkg.QpGpuError) as e:
        pool.submit_host(wires, pis)
    assert e.value.code == -1 and "QPGPU_POOL_HOST_WITNESS" in str(e.value)
    d = gpu.to_device(wires)
    assert len(pool.wait(pool.submit(d, pis))) == pool.proof_size()
    pool.close(); d.free()


def test_leaf_c_example_runs(pkg):
    """examples/leaf_prove_example.c: the leaf path from plain C over devices {0, 0} (no Python, no torch in the process)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = os.path.join(tempfile.gettempdir(), "qpgpu_leaf_prove_example_gpu")
    subprocess.check_call(["gcc", "-O2", "-I", os.path.join(root, "include"), os.path.join(root, "examples", "leaf_prove_example.c"),
                           "-L", os.path.join(root, "qp-zk-circuits_amd"), "-lqpgpu", "-lpthread",
                           "-Wl,-rpath," + os.path.join(root, "qp-zk-circuits_amd"), "-o", out])
    res = subprocess.run([out, "0", "0,0", "2", "4", "2"], capture_output=True, text=True, timeout=180)
    assert res.returncode == 0, res.stderr + res.stdout
    assert "ok devices=2 workers=2 lockstep=4" in res.stdout and "unsatisfiable job alone" in res.stdout
    # the same with the front-end's hash hints riding along (sixth argument): the same proof (the example prints its checksum)
    hinted = subprocess.run([out, "0", "0,0", "2", "4", "2", "1"], capture_output=True, text=True, timeout=180)
    assert hinted.returncode == 0, hinted.stderr + hinted.stdout
    fnv = lambda text: [w for w in text.split() if w.startswith("fnv1a=")][0]
    assert "(hash hints)" in hinted.stdout and "unsatisfiable job alone" in hinted.stdout and fnv(hinted.stdout) == fnv(res.stdout)


def test_private_batches_through_the_pool(pkg, gpu, orc):
    """The private layer through the multi-device pool (two worker sets on the one GPU): a zero-knowledge private-batch circuit
    over two leaf proofs, jobs = fill_private_batch_witness's values only — the blinding wires are drawn on the device
    (qpgpu_pool_set_partial_cells_blinded), the public inputs are read out of each job's witness (public_inputs = NULL) — and every
    proof verifies and carries the public inputs the host restatement predicts; a job that hands in OTHER public inputs, or inner
    proofs the layer does not accept, fails alone."""
    import leaf_cases as lc
    import oracle_binding as ob
    L, R, A = pkg.leaf, pkg.recursion, pkg.aggregation
    leaf = L.LeafCircuit()
    lp = L.LeafProver(pkg, gpu, leaf)
    sp = lc.shared_tree_inputs(L, 3, seed=31)
    proofs = [lp.prove(x)[0] for x in sp] + [lp.prove(lc.dummy_inputs(L))[0]]
    other = lp.prove(lc.real_inputs(L, depth=2))[0]
    ver = pkg.Verifier(leaf.pack, circuit=lp.circ)
    w = R.WrapperCircuit(leaf.pack, ver, 2, num_routed_wires=60, logic="private_batch", verify=True, zero_knowledge=True)
    pool = pkg.ProvingPool(w.pack, workers=1, devices=[0, 0], max_batch=2)
    pre = [np.arange(8, dtype=np.uint64).reshape(2, 4) + 10 * k for k in range(4)]
    slots = [[proofs[0], proofs[1]], [proofs[3], proofs[2]], [proofs[1], proofs[3]], [proofs[2], proofs[0]]]
    com = [w.commit(s, preimages=pre[k], device_blinding=True) for k, s in enumerate(slots)]
    pool.set_partial_cells(com[0][0], n_blinding=w.blinding_cells.size)
    tickets = [pool.submit_partial(c[1], None) for c in com]
    # the same slots with the host restatement's public inputs handed in (compared with the witness's), with wrong ones, and slots
    # of two blocks
    t_given = pool.submit_partial(com[0][1], com[0][2])
    wrong = com[0][2].copy(); wrong[8] += 1
    t_wrong = pool.submit_partial(com[0][1], wrong)
    bad = w.commit([proofs[0], other], preimages=pre[0], device_blinding=True, public_inputs=com[0][2])
    t_bad = pool.submit_partial(bad[1], None)
    wv = pkg.Verifier(w.pack)
    npis = A.private_batch_pi_len(2)
    for k, t in enumerate(tickets + [t_given]):
        proof = pool.wait(t)
        assert wv.verify(proof), k
        assert A.proof_public_inputs(proof, npis).tolist() == com[k % 4][2].tolist(), k
    for t in (t_wrong, t_bad):
        with pytest.raises(pkg.QpGpuError) as e:
            pool.wait(t)
        assert e.value.code == -4
    oc = ob.OracleCircuit(orc, w.pack)
    assert oc.verify(pool.wait(pool.submit_partial(com[1][1], None))) == 0
    oc.close()
    pool.close(); wv.close(); ver.close(); lp.close()
