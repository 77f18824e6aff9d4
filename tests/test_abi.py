"""The C-ABI library loads on a CPU-only box and exports every symbol include/qpgpu.h declares."""
import ctypes
import os


def test_library_exports_header_symbols(pkg):
    so = pkg.lib_path()
    assert os.path.exists(so), "libqpgpu.so not built (python -c 'import __graft_entry__ as g; g.build()')"
    lib = ctypes.CDLL(so)
    names = pkg.binding.exported_symbols()
    assert len(names) >= 13
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/qpgpu.h but not exported"
    lib.qpgpu_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.qpgpu_version()


def test_no_gpu_is_a_loud_error(pkg):
    import torch
    if torch.cuda.is_available():
        return
    try:
        pkg.QpGpu(0)
    except pkg.QpGpuError as e:
        assert e.code != 0
    else:
        raise AssertionError("context creation must fail without a GPU; there is no CPU fallback")


def test_product_does_not_reference_oracle():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg_dir = os.path.join(root, "qp-zk-circuits_amd")
    for dp, _, fs in os.walk(pkg_dir):
        for f in fs:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")) or f == "Makefile":
                text = open(os.path.join(dp, f), errors="ignore").read()
                assert "liboracle" not in text and "oracle/" not in text and "oracle_binding" not in text, f
    # nor do the tools, examples and the Rust-side binding: drivers that check against the oracle live under tests/soak/
    for sub in ("tools", "examples", "integration", "include"):
        for dp, _, fs in os.walk(os.path.join(root, sub)):
            if "scratch_bin" in dp or "__pycache__" in dp:
                continue
            for f in fs:
                if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h", ".c", ".sh", ".rs")):
                    text = open(os.path.join(dp, f), errors="ignore").read()
                    assert "liboracle" not in text and "oracle_binding" not in text and "load_oracle" not in text, os.path.join(dp, f)


def test_c_example_compiles_against_the_header():
    """examples/prove_example.c uses only include/qpgpu.h; it must compile and link with plain gcc."""
    import subprocess, tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = os.path.join(tempfile.gettempdir(), "qpgpu_prove_example")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-I", os.path.join(root, "include"), os.path.join(root, "examples", "prove_example.c"),
                           "-L", os.path.join(root, "qp-zk-circuits_amd"), "-lqpgpu",
                           "-Wl,-rpath," + os.path.join(root, "qp-zk-circuits_amd"), "-o", out])
    assert os.path.exists(out)


def test_library_challenger_matches_oracle(pkg, orc):
    """qpgpu_challenger_* (host only) against the restated duplex challenger on a mixed observe/squeeze schedule."""
    import numpy as np
    from oracle_binding import Challenger as OracleChallenger
    rng = np.random.default_rng(11)
    a, b = pkg.Challenger(), OracleChallenger(orc)
    for step in range(40):
        k = int(rng.integers(0, 20))
        xs = rng.integers(0, 0xFFFFFFFF00000001, size=k, dtype=np.uint64)
        if k:
            a.observe(xs); b.observe(xs)
        m = int(rng.integers(0, 11))
        assert a.get_n(m) == b.get_n(m), step
    assert a.state.input_len == 0 or a.state.output_len == 0


def test_crash_trace_leaves_a_native_backtrace(pkg, tmp_path):
    """QPGPU_CRASH_TRACE (csrc/crash_trace.cpp): a fatal signal in a process that has the library loaded leaves the faulting
    address, its mapping and the native backtrace in the named file, then the previous handler (python's faulthandler) runs and
    the process still dies of the signal. Off by default."""
    import subprocess
    import sys
    so = pkg.lib_path()
    log = tmp_path / "crash.txt"
    code = ("import ctypes, faulthandler, sys; faulthandler.enable(); lib = ctypes.CDLL(%r); "
            "assert lib.qpgpu_crash_trace_armed() == int(sys.argv[1]); ctypes.string_at(16)" % so)
    env = dict(os.environ, QPGPU_CRASH_TRACE=str(log))
    r = subprocess.run([sys.executable, "-c", code, "1"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == -11, (r.returncode, r.stderr[-400:])
    text = log.read_text()
    assert "libqpgpu crash trace: signal 11 (SIGSEGV)" in text and "address 0x0000000000000010" in text
    assert "not mapped" in text and "native backtrace" in text and "end of libqpgpu crash trace" in text
    assert "Fatal Python error: Segmentation fault" in r.stderr          # faulthandler still had its turn
    env.pop("QPGPU_CRASH_TRACE")
    r = subprocess.run([sys.executable, "-c", code, "0"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == -11


def test_every_entry_point_takes_null_arguments_without_faulting():
    """tools/null_arg_sweep.py: each of the exported qpgpu_* functions, called with NULL / 0 for every argument in a child process of its
    own, returns; none of them faults (a NULL context, handle or buffer is an error code, not a crash)."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, os.path.join(root, "tools", "null_arg_sweep.py")], capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-1000:]
    assert ", 0 crashed, 0 without a prototype" in res.stdout


import pytest


@pytest.mark.gpu
def test_pci_bus_id_names_the_cards_sysfs_directory(gpu):
    """qpgpu_ctx_pci_bus_id: "dddd:bb:dd.f", lower case, the key bench.py reads the card's engine clock under; a short buffer is refused."""
    import re
    bdf = gpu.pci_bus_id()
    assert re.fullmatch(r"[0-9a-f]{4}:[0-9a-f]{2}:[0-9a-f]{2}\.[0-7]", bdf), bdf
    assert os.path.isdir(os.path.join("/sys/bus/pci/devices", bdf))
    buf = ctypes.create_string_buffer(8)
    assert gpu.lib.qpgpu_ctx_pci_bus_id(gpu.ctx, buf, 8) == -5 and buf.value == b""
