#!/usr/bin/env python3
"""Random gadget programs (qpgpu_builder_gadget_circuit(1000 + seed): 40-80 applications of the native builder's gadgets over a pool
of values) as a differential soak: for every seed the circuit is built, the ORACLE generates the witness from random inputs, proves and
verifies, the library's host verifier agrees; with --gpu the device's stage s1 must reproduce the oracle's wire matrix and the device
proof the oracle's bytes. Usage: fuzz_gadget_programs.py FIRST_SEED COUNT [--gpu]. Prints one summary line (appended to
profiles/r04_soak.txt by hand)."""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g  # noqa: E402

P = 0xFFFFFFFF00000001


def main():
    first, count = int(sys.argv[1]), int(sys.argv[2])
    on_gpu = "--gpu" in sys.argv
    pkg = g.load_package()
    import oracle_binding as ob
    orc = ob.Oracle()
    L = pkg.load_library(); c = ctypes
    L.qpgpu_builder_gadget_circuit.restype = c.c_int
    L.qpgpu_builder_gadget_circuit.argtypes = [c.c_uint, c.c_void_p, c.c_size_t, c.POINTER(c.c_size_t), c.c_void_p, c.c_size_t, c.POINTER(c.c_size_t), c.POINTER(c.c_size_t), c.c_char_p]
    gpu = pkg.QpGpu(0) if on_gpu else None
    t0 = time.time()
    bad, rows, zero_inv = [], {}, 0
    for seed in range(first, first + count):
        kind = 1000 + seed
        n, ni, no = c.c_size_t(), c.c_size_t(), c.c_size_t(); err = c.create_string_buffer(400)
        if L.qpgpu_builder_gadget_circuit(kind, None, 0, c.byref(n), None, 0, c.byref(ni), c.byref(no), err):
            bad.append((seed, "build: " + err.value.decode())); continue
        pack = np.empty(n.value, dtype=np.uint64); cells = np.empty(ni.value + no.value, dtype=np.uint64)
        L.qpgpu_builder_gadget_circuit(kind, pack.ctypes.data, pack.size, c.byref(n), cells.ctypes.data, cells.size, c.byref(ni), c.byref(no), err)
        cin, cout = cells[:ni.value], cells[ni.value:]
        rng = np.random.default_rng(100000 + seed)
        NB = 4
        vals = rng.integers(0, P, (NB, cin.size), dtype=np.uint64)
        vals[1, rng.integers(0, cin.size)] = 0                               # an input that is zero
        vals[2, :] = 0                                                       # every input zero (equalities hold, selectors pick the other way)
        vals[3, :] = np.uint64(P - 1)                                        # every input p - 1
        db = int(pack[1]); rows[db] = rows.get(db, 0) + 1
        oc = ob.OracleCircuit(orc, pack)
        ver = pkg.Verifier(pack)
        want = []
        for b in range(NB):
            rc, wires, _ = orc.generate_witness(pack, cin, vals[b], None)
            if rc == orc.WIT_ZERO_INVERSE:
                zero_inv += 1; want.append(None); continue
            if rc != orc.WIT_OK:
                bad.append((seed, "oracle witness rc %d" % rc)); want.append(None); continue
            pis = np.array([wires[int(x) % 135, int(x) // 135] for x in cout], dtype=np.uint64)
            proof = oc.prove(wires, pis)
            if oc.verify(proof) != 0 or not ver.verify(proof):
                bad.append((seed, "honest proof rejected (input set %d): %s" % (b, ver.reason)))
            want.append((wires, pis, proof))
        if on_gpu:
            circ = pkg.Circuit(gpu, pack, max_batch=NB)
            d = gpu.alloc(NB * 135 * (1 << db) * 8)
            st = circ.generate_witness_partial_batch_dev(cin, vals, None, d)
            got = d.download().reshape(NB, 135, 1 << db)
            for b in range(NB):
                if want[b] is None:
                    if st[b] == 0:
                        bad.append((seed, "device accepted what the oracle refused (input set %d)" % b))
                    continue
                if st[b] != 0 or not np.array_equal(got[b], want[b][0]):
                    bad.append((seed, "device witness differs (input set %d, status %d)" % (b, st[b]))); continue
            if want[0] is not None and st[0] == 0:
                if circ.prove_dev(d, want[0][1]) != want[0][2]:
                    bad.append((seed, "device proof bytes differ"))
            circ.close(); d.free(scrub=True)
        ver.close(); oc.close()
    print("fuzz_gadget_programs seeds %d..%d %s: %d mismatches, %d zero-inverse refusals (both sides), rows %s, %.0f s" %
          (first, first + count - 1, "device+oracle" if on_gpu else "oracle+host verifier", len(bad), zero_inv, dict(sorted(rows.items())), time.time() - t0))
    for b in bad[:20]:
        print("  ", b)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
