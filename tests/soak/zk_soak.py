"""Soak with zero-knowledge salts: fresh randomness per proof (no injected seed), every proof checked by the restated
verifier; proofs of the same witness must differ (salted leaves) and all verify. usage: zk_soak.py [count]"""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
import oracle_binding
orc = oracle_binding.Oracle()
count = int(sys.argv[1]) if len(sys.argv) > 1 else 200
pack, wires, pis = pkg.synth_circuit(10, num_routed=60, seed=8, poseidon=True, base_sum=True, ext_arith=True, recursion=True)
pack[14] = 1
gpu = pkg.QpGpu(0)
d = gpu.to_device(wires)
pool = pkg.ProvingPool(pack, workers=2, max_batch=16)
circ = pkg.Circuit(gpu, pack)
ver = pkg.Verifier(pack, circuit=circ)
oc = oracle_binding.OracleCircuit(orc, pack)
tickets = [pool.submit(d, pis) for _ in range(count)]
proofs = [pool.wait(t) for t in tickets]
bad = sum(not ok for ok in ver.verify_many(proofs)) + sum(oc.verify(p) != 0 for p in proofs[::10])
print(f"{count} zero-knowledge proofs through the batched pool: {bad} rejected (library verifier on all, oracle on every tenth), {len(set(proofs))} distinct")
ver.close(); circ.close()
pool.close()
assert bad == 0 and len(set(proofs)) == count
