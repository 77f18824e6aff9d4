#!/usr/bin/env python3
"""One pass of stage s1 on a 2^16-row recursion-mix circuit (the public batch stand-in), for rocprofv3 --kernel-trace:
which dependency levels cost what. usage: rocprofv3 --kernel-trace --output-format csv -d DIR -o w -- python3 tools/witness_profile.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
gpu = pkg.QpGpu(0)
rec = dict(poseidon=True, base_sum=True, ext_arith=True, recursion=True)
pack, wires, pis = pkg.synth_circuit(16, num_wires=135, num_routed=80, num_public_inputs=908, seed=77, **rec)
circ = pkg.Circuit(gpu, pack)
mask = circ.witness_free_mask(*wires.shape)
d = gpu.to_device(np.where(mask == 1, wires, 0).astype(np.uint64))
for _ in range(2):
    circ.generate_witness_dev(d, pis)
gpu.sync()
print("levels", circ.witness_info())
circ.close(); gpu.close()
