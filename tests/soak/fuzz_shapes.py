"""Randomised differential run: circuit shapes, gate mixes, FRI knobs, zero knowledge and witness hints drawn at random;
GPU proof bytes against the CPU restatement, the restated verifier and the library's own host verifier, a lockstep batch of
three against the single-proof bytes, and device witness generation against the full witness.
usage: fuzz_shapes.py [cases] [seed] [max degree_bits] [poseidon|poseidon2]   (last: the proof-system hasher; poseidon2 = qp-poseidon-core's
parameters on the context, the process default and the oracle)"""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
import oracle_binding
from test_prove_gpu import _with_fri_config
orc = oracle_binding.Oracle()
hasher = sys.argv[4] if len(sys.argv) > 4 else "poseidon"
if hasher == "poseidon2":
    qp = pkg.poseidon2_qp_params()
    pkg.set_hasher_poseidon2(*qp)      # the synthetic generator and the host verifier hash with the process default
    orc.select_poseidon2(*qp)
    gpu = pkg.QpGpu(0, hasher=qp)
else:
    gpu = pkg.QpGpu(0)
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
max_d = int(sys.argv[3]) if len(sys.argv) > 3 else 11
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0 = time.time()
tally = {"poseidon2": 0, "poseidon2_alt_layout": 0, "zero_knowledge": 0, "hints": 0, "recursion_gates": 0, "lockstep_batches": 0, "rejected_at_load": 0}
for i in range(cases):
    d = int(rng.integers(3, max_d + 1))
    big = bool(rng.integers(0, 2))
    routed = int(rng.integers(12, 21)) * 4 if big else int(rng.integers(2, 12)) * 4
    wires_n = max(135 if big else routed + int(rng.integers(0, 20)), routed)
    kw = dict(num_wires=wires_n, num_routed=routed, num_public_inputs=int(rng.integers(0, 9)), seed=1000 + i,
              poseidon=big and bool(rng.integers(0, 2)), base_sum=bool(rng.integers(0, 2)), ext_arith=routed >= 8 and bool(rng.integers(0, 2)),
              recursion=routed >= 48 and wires_n >= 64 and bool(rng.integers(0, 2)), hints=bool(rng.integers(0, 2)))
    # the qp fork's Poseidon2 gate (needs 135 wires, 40 routed wires, 32 rows), default or alternative wire layout
    if big and routed >= 40 and d >= 5 and bool(rng.integers(0, 2)):
        kw.update(poseidon2=True, p2_alt_layout=bool(rng.integers(0, 2)))
    pack, wires, pis = pkg.synth_circuit(d, **kw)
    tally["poseidon2"] += bool(kw.get("poseidon2")); tally["poseidon2_alt_layout"] += bool(kw.get("p2_alt_layout")); tally["hints"] += kw["hints"]; tally["recursion_gates"] += kw["recursion"]
    rate = int(rng.choice([3, 3, 3, 4, 5]))
    knobs = dict(cap_height=int(rng.integers(0, min(7, d + rate) + 1)), pow_bits=int(rng.choice([0, 4, 12, 16])),
                 num_queries=int(rng.integers(1, 32)), rate_bits=rate)
    pack = _with_fri_config(pack, **knobs)
    pack[6] = int(rng.integers(1, 5))                  # num_challenges
    zk = bool(rng.integers(0, 2)); pack[14] = 1 if zk else 0
    desc = f"case {i}: d={d} {kw} {knobs} nch={int(pack[6])} zk={zk}"
    try:
        circ = pkg.Circuit(gpu, pack)
    except pkg.QpGpuError as e:
        tally["rejected_at_load"] += 1
        print("skip (rejected at load):", desc, e); continue
    oc = oracle_binding.OracleCircuit(orc, pack)
    circ.set_blinding_seed(7 + i)
    got = circ.prove(wires, pis)
    want = oc.prove(wires, pis, seed=7 + i)
    assert got == want, "BYTES DIFFER " + desc
    assert oc.verify(got) == 0, "REJECTED " + desc
    ver = pkg.Verifier(pack, circuit=circ, hasher=1 if hasher == "poseidon2" else 0)
    assert ver.verify(got), "LIBRARY VERIFIER REJECTED " + desc + " " + ver.reason
    bad = bytearray(got); bad[int(rng.integers(0, len(got)))] ^= 1 << int(rng.integers(0, 8))
    assert not ver.verify(bytes(bad)) and oc.verify(bytes(bad)) != 0, "TAMPERING ACCEPTED " + desc
    ver.close()
    tally["zero_knowledge"] += zk
    if d <= 9 or i % 4 == 0:      # the lockstep path over the same shape: proof b of the batch is blinded with seed + b
        tally["lockstep_batches"] += 1
        cb = pkg.Circuit(gpu, pack, max_batch=3)
        cb.set_blinding_seed(7 + i)
        dense = gpu.to_device(np.stack([wires] * 3))
        outs = cb.prove_batch_dev([dense.ptr + b * wires.nbytes for b in range(3)], [pis] * 3)
        assert outs[0] == got, "BATCH PROOF 0 DIFFERS " + desc
        for b in (1, 2):
            assert outs[b] == (oc.prove(wires, pis, seed=7 + i + b) if zk else got), f"BATCH PROOF {b} DIFFERS " + desc
        dense.free(); cb.close()
    mask = circ.witness_free_mask(*wires.shape)
    full = circ.generate_witness(np.where(mask == 1, wires, 0).astype(np.uint64), pis)
    assert (full == wires).all(), "WITNESS DIFFERS " + desc
    if kw.get("poseidon2"):   # the Poseidon2 rows' digests are the KAT-pinned sponge of their preimages
        for site in pkg.synth_p2_sites(d, kw["num_public_inputs"], poseidon2=True, p2_alt_layout=kw["p2_alt_layout"]):
            pre, dig = pkg.p2_site_cells(pack, site)
            x = np.ascontiguousarray([full[c, r] for c, r in pre], dtype=np.uint64); out = np.empty(4, dtype=np.uint64)
            assert pkg.load_library().qpgpu_poseidon2_hash_pad10(None, 0, x.ctypes.data if x.size else None, x.size, out.ctypes.data) == 0
            assert [int(full[c, r]) for c, r in dig] == [int(v) for v in out], "POSEIDON2 DIGEST DIFFERS " + desc
    circ.close(); oc.close()
    if (i + 1) % 100 == 0:
        print(f"  {i + 1} cases so far, {time.time()-t0:.0f} s", flush=True)
print(f"{cases} cases ok under the {hasher} hasher in {time.time()-t0:.1f} s (seed {sys.argv[2] if len(sys.argv) > 2 else 1}, max degree_bits {max_d}): " + ", ".join(f"{k} {v}" for k, v in tally.items()))
