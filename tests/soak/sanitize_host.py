#!/usr/bin/env python3
"""AddressSanitizer + UndefinedBehaviorSanitizer pass over the host-only sources of the library (CPU build with g++; GPU
sanitizers are not available on the pool): builds tools/sanitize/driver.cpp with csrc/{batch,proof_targets,wire,leaf_witness,verifier,circuit,
poseidon_constants}.cpp, lets the CPU oracle make one valid proof, and runs the driver on it (mutated / truncated / random
proofs and packs, random public-input rows, random config text). usage: python tests/soak/sanitize_host.py [iterations]"""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as ge
import oracle_binding

iters = sys.argv[1] if len(sys.argv) > 1 else "300"
pkg = ge.load_package()
orc = oracle_binding.Oracle()
pack, wires, pis = pkg.synth_circuit(7, num_wires=135, num_routed=80, num_public_inputs=21, seed=31, poseidon=True, base_sum=True, ext_arith=True, recursion=True, poseidon2=True)
oc = oracle_binding.OracleCircuit(orc, pack)
proof = oc.prove(wires, pis)
oc.close()
work = tempfile.mkdtemp(prefix="qp_host_asan_")
np.ascontiguousarray(pack, dtype="<u8").tofile(os.path.join(work, "pack.bin"))
open(os.path.join(work, "proof.bin"), "wb").write(proof)
csrc = os.path.join(ROOT, "qp-zk-circuits_amd", "csrc")
srcs = [os.path.join(csrc, f) for f in ("batch.cpp", "proof_targets.cpp", "wire.cpp", "leaf_witness.cpp", "verifier.cpp", "circuit.cpp", "poseidon_constants.cpp")]
exe = os.path.join(work, "driver")
cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-pthread",
       "-I", csrc, os.path.join(ROOT, "tools", "sanitize", "driver.cpp")] + srcs + ["-o", exe]
subprocess.check_call(cmd)
env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1")
rc = subprocess.call([exe, os.path.join(work, "pack.bin"), os.path.join(work, "proof.bin"), iters], env=env)
print("sanitizer pass (parsers, verifier, batch host side)", "clean" if rc == 0 else f"FAILED (exit {rc})")
if rc:
    sys.exit(rc)
# second driver: the circuit builder and the circuits restated on it (builder.cpp, leaf_circuit.cpp, wrapper_circuit.cpp)
fake = pkg.leaf.LeafCircuit(fragment=pkg.leaf.FRAGMENT_FAKE_LEAF)
none = np.zeros(0, dtype=np.uint64)
fpis = np.arange(21, dtype=np.uint64) + 5
rcw, fw, _ = orc.generate_witness(fake.pack, none, none, fpis)
assert rcw == orc.WIT_OK
foc = oracle_binding.OracleCircuit(orc, fake.pack)
open(os.path.join(work, "fake_proof.bin"), "wb").write(foc.prove(fw, fpis))
foc.close()
srcs2 = [os.path.join(csrc, f) for f in ("builder.cpp", "leaf_circuit.cpp", "wrapper_circuit.cpp", "gadget_circuits.cpp", "batch.cpp", "proof_targets.cpp", "wire.cpp", "leaf_witness.cpp", "verifier.cpp",
                                          "circuit.cpp", "poseidon_constants.cpp")]
exe2 = os.path.join(work, "builder_driver")
subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-pthread",
                       "-I", csrc, os.path.join(ROOT, "tools", "sanitize", "builder_driver.cpp")] + srcs2 + ["-o", exe2])
rc = subprocess.call([exe2, os.path.join(work, "fake_proof.bin"), str(max(10, int(iters) // 5))], env=env)
print("sanitizer pass (circuit builder, leaf / wrapper circuits)", "clean" if rc == 0 else f"FAILED (exit {rc})")
sys.exit(rc)
