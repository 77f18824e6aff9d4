"""Plan shape and time of device witness generation (stage s1) for the bench circuits."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import __graft_entry__ as g
pkg = g.load_package()
gpu = pkg.QpGpu(0)
for d, kw in ((13, dict(poseidon=True, base_sum=True)), (16, dict(poseidon=True, base_sum=True, ext_arith=True, recursion=True))):
    pack, wires, pis = pkg.synth_circuit(d, seed=1, **kw)
    circ = pkg.Circuit(gpu, pack)
    t0 = time.perf_counter(); info = circ.witness_info(); t_plan = time.perf_counter() - t0
    mask = circ.witness_free_mask(*wires.shape)
    dbuf = gpu.to_device(np.where(mask == 1, wires, 0).astype(np.uint64))
    circ.generate_witness_dev(dbuf, pis); gpu.sync()
    ok = bool((dbuf.download().reshape(wires.shape) == wires).all())
    t0 = time.perf_counter()
    for _ in range(10):
        circ.generate_witness_dev(dbuf, pis)
    gpu.sync()
    dt = (time.perf_counter() - t0) / 10
    B = 16 if d == 13 else 4
    big = gpu.to_device(np.tile(np.where(mask == 1, wires, 0).astype(np.uint64), (B, 1, 1)))
    bp = np.tile(pis, (B, 1))
    circ.generate_witness_dev(big, bp, batch=B); gpu.sync()
    t0 = time.perf_counter()
    for _ in range(5):
        circ.generate_witness_dev(big, bp, batch=B)
    gpu.sync()
    dtb = (time.perf_counter() - t0) / 5
    big.free()
    print(f"d={d}: batch of {B}: {dtb*1e3:.3f} ms = {dtb/B*1e3:.3f} ms per witness", flush=True)
    print(f"d={d}: generators={info[0]} levels={info[1]} free_cells={info[2]} plan_build_s={t_plan:.2f} generate_ms={dt*1e3:.3f} equal={ok}", flush=True)
    circ.close(); dbuf.free()
