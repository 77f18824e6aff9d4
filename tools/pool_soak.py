"""Soak: thousands of proofs through the proving pool, every one compared with the first (same witness -> same bytes),
plus a second witness interleaved, through the batched pool (workers take up to max_batch queued jobs per lockstep pass, so
batches mix the two witnesses in every composition). Catches rare races between workers / streams / batch slots.
usage: pool_soak.py [seconds] [workers] [max_batch]"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import __graft_entry__ as g
pkg = g.load_package()
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
gpu = pkg.QpGpu(0)
pack, wires, pis = pkg.synth_circuit(11, seed=5, poseidon=True, base_sum=True, ext_arith=True, recursion=True)
circ = pkg.Circuit(gpu, pack)
mask = circ.witness_free_mask(*wires.shape)
part = np.where(mask == 1, wires, 0).astype(np.uint64)
col = next(c for c in (0, 1, 2, 4, 5, 6) if mask[c, 3])
part2 = part.copy(); part2[col, 3] = np.uint64(424242)
w2 = circ.generate_witness(part2, pis)
want = [circ.prove(wires, pis), circ.prove(w2, pis)]
circ.close()
d = [gpu.to_device(wires), gpu.to_device(w2)]
workers = int(sys.argv[2]) if len(sys.argv) > 2 else 3
max_batch = int(sys.argv[3]) if len(sys.argv) > 3 else 16
pool = pkg.ProvingPool(pack, workers=workers, max_batch=max_batch)
rng = np.random.default_rng(1)
t0 = time.time(); n = 0; bad = 0
while time.time() - t0 < secs:
    ts = [(pool.submit(d[k], pis), k) for k in rng.integers(0, 2, 96).tolist()]
    for t, k in ts:
        bad += pool.wait(t) != want[k]
        n += 1
print(f"{n} proofs in {time.time()-t0:.1f} s ({n/(time.time()-t0):.0f}/s) through {workers} workers x {max_batch} lockstep slots, mismatches: {bad}")
pool.close()
assert bad == 0
