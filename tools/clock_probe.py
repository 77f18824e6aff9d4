#!/usr/bin/env python3
"""Do sparse, tiny, dependent kernels run at a lower clock? Times one pass of stage s1 (2^13-row leaf circuit: 280 dependency
levels, one small launch each) alone and while a second stream keeps the GPU busy with 2^20-point NTTs."""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as ge
pkg = ge.load_package()
gpu = pkg.QpGpu(0)
pack, wires, pis = pkg.synth_circuit(13, num_wires=135, num_routed=80, num_public_inputs=21, seed=1000, poseidon=True, base_sum=True)
circ = pkg.Circuit(gpu, pack)
mask = circ.witness_free_mask(*wires.shape)
d = gpu.to_device(np.where(mask == 1, wires, 0).astype(np.uint64))
def s1(reps=5):
    circ.generate_witness_dev(d, pis); gpu.sync()
    t0 = time.perf_counter()
    for _ in range(reps): circ.generate_witness_dev(d, pis)
    gpu.sync()
    return (time.perf_counter() - t0) / reps * 1e3
def prove(reps=5):
    out = circ.prove_dev(d, pis); gpu.sync()
    t0 = time.perf_counter()
    for _ in range(reps): circ.prove_dev(d, pis)
    return (time.perf_counter() - t0) / reps * 1e3
print("alone: s1 %.2f ms, single proof %.2f ms" % (s1(), prove()))
g2 = pkg.QpGpu(0)
dev = torch.device("cuda", 0)
x = torch.zeros((64, 1 << 20), dtype=torch.int64, device=dev); y = torch.empty_like(x)
stop = False
def busy():
    while not stop:
        for _ in range(20): g2.ntt_dev(x, y, 20, 64)
        g2.sync()
th = threading.Thread(target=busy); th.start()
time.sleep(0.5)
print("with a busy second stream: s1 %.2f ms, single proof %.2f ms" % (s1(), prove()))
stop = True; th.join()
time.sleep(0.2)
print("alone again: s1 %.2f ms, single proof %.2f ms" % (s1(), prove()))
circ.close(); g2.close(); gpu.close()
