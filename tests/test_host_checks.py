"""Host build (g++) of the arithmetic the kernels share with the host through GL_HD headers: the spectral Poseidon permutation
against its layer-wise and textbook forms on random and extreme states, the multiplication-free Poseidon2 external layer, the
quotient kernels' 192-bit accumulators, and the matrix form of the partial rounds (poseidon_mfma.hpp: integer emulation of the
device schedule from the table bytes the kernels load). tools/host_checks/poseidon_permutation_check.cpp; about ten seconds."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shared_arithmetic_on_the_host(tmp_path):
    csrc = os.path.join(ROOT, "qp-zk-circuits_amd", "csrc")
    exe = str(tmp_path / "check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", csrc, os.path.join(ROOT, "tools", "host_checks", "poseidon_permutation_check.cpp"),
                           os.path.join(csrc, "poseidon_constants.cpp"), "-o", exe, "-lpthread"])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("mismatches 0") == 6, r.stdout
