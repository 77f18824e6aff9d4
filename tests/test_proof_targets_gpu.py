"""fill_private_batch_witness end to end on the device path: the (logical target, value) list of two inner proofs and their
dummy-nullifier preimages goes through a target map into qpgpu_generate_witness_partial_dev of a stand-in wrapper circuit
(the real wrapper's pack and target map come from the Rust exporter, integration/qpgpu_backend.rs); every assigned target
must hold its value in the generated witness, and the wrapper then proves."""
import ctypes

import numpy as np
import pytest

from oracle_binding import OracleCircuit
from test_proof_targets import fill, lib  # noqa: F401  (fixture)


@pytest.mark.gpu
def test_inner_proofs_become_the_wrappers_partial_witness(pkg, gpu, orc, lib):  # noqa: F811
    ipack, iwires, ipis = pkg.synth_circuit(6, seed=61, poseidon=True, base_sum=True)
    icirc = pkg.Circuit(gpu, ipack)
    proofs = [icirc.prove(iwires, ipis), icirc.prove(iwires, (ipis + np.uint64(0)))]
    icirc.close()
    T = lib.qpgpu_proof_target_count(ipack.ctypes.data, ipack.size)
    pre = np.array([[5, 6, 7, 8], [9, 10, 11, 12]], dtype=np.uint64)
    rc, ids, vals, msg, n = fill(lib, ipack, proofs, 2, pre, 2)
    assert rc == 0 and n == 2 * (T + 4), msg
    # stand-in wrapper: a synthetic circuit with enough caller-supplied cells; target map = logical id -> the id-th free cell
    # (its free cells are operands of ArithmeticGate operations, inputs of PoseidonGate rows and unused advice wires: any value
    # satisfies the circuit — except the PoseidonGate swap bit, wire 24, which is left out of the map)
    wpack, wwires, wpis = pkg.synth_circuit(11, num_routed=60, num_public_inputs=21 * 2 + 8, seed=62, poseidon=True)
    wc = pkg.Circuit(gpu, wpack); oc = OracleCircuit(orc, wpack)
    try:
        mask = wc.witness_free_mask(*wwires.shape)                    # [wire, row]: 1 = a PartialWitness cell
        pub = set(int(c) for c in pkg.pack_public_input_cells(wpack))
        free = [int(r) * 135 + int(c) for c, r in zip(*np.nonzero(mask)) if int(c) != 24 and int(r) * 135 + int(c) not in pub]
        assert len(free) >= n, (len(free), n)
        target_map = np.array(free[:n], dtype=np.uint64)
        cells = target_map[ids]
        d_w = gpu.alloc(wwires.nbytes)
        wc.generate_witness_partial_dev(cells, vals, wpis, d_w)
        got = d_w.download().reshape(wwires.shape)
        assert all(int(got[int(c) % 135, int(c) // 135]) == int(v) for c, v in zip(cells[::7], vals[::7]))
        assert all(int(got[int(c) % 135, int(c) // 135]) == int(v) for c, v in zip(cells[-8:], vals[-8:]))     # the preimages
        # the wrapper's other free cells stayed zero, its generators ran: the witness satisfies the stand-in circuit
        wc.set_witness_check(True)
        out = np.empty(wc.proof_size(), dtype=np.uint8)
        proof = wc.prove_dev(d_w, wpis, out)
        assert oc.verify(proof) == 0
        # "set twice with different values": the same target assigned two values, as two inconsistent inner proofs would
        c2 = np.concatenate([cells, cells[:1]]); v2 = np.concatenate([vals, np.array([(int(vals[0]) + 1) % pkg.P], dtype=np.uint64)])
        with pytest.raises(pkg.QpGpuError) as e:
            wc.generate_witness_partial_dev(c2, v2, wpis, d_w)
        assert e.value.code == -4 and "set twice with different values" in str(e.value)
        d_w.free(scrub=True)
    finally:
        wc.close(); oc.close()
