"""Every exported qpgpu_* function whose prototype the ctypes binding declares, called with NULL for every pointer and 0 for every
number, one child process per function: each must return (an error code, a null, nothing) — a crash of the child is a missing
argument check. Host only; no GPU is touched (a NULL context or handle has to be refused before anything else happens).
usage: python tools/null_arg_sweep.py            (prints one line per function that crashed, then a summary; exit 1 if any)"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def declared():
    import __graft_entry__ as g
    pkg = g.load_package()
    lib = pkg.load_library()
    pkg.leaf._lib(); pkg.aggregation._lib()
    out = subprocess.run(["nm", "-D", "--defined-only", pkg.binding.lib_path() if hasattr(pkg, "binding") else os.path.join(ROOT, "qp-zk-circuits_amd", "libqpgpu.so")],
                         capture_output=True, text=True).stdout
    names = sorted({l.split()[-1] for l in out.splitlines() if " T " in l and l.split()[-1].startswith("qpgpu_")})
    return lib, names


def header_prototypes():
    """name -> list of 'p' (pointer) / 'f' (floating) / 'i' (integer) per parameter, read off include/*.h"""
    import glob, re
    protos = {}
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        text = re.sub(r"/\*.*?\*/", " ", open(h).read(), flags=re.S)
        for m in re.finditer(r"\b(qpgpu_\w+)\s*\(([^;{}]*?)\)\s*;", text, flags=re.S):
            params = [q.strip() for q in m.group(2).split(",")]
            if params == ["void"] or params == [""]:
                params = []
            protos[m.group(1)] = ["p" if "*" in q or "[" in q else ("f" if re.search(r"\b(double|float)\b", q) else "i") for q in params]
    return protos


def child(name):
    lib, _ = declared()
    fn = getattr(lib, name)
    if fn.argtypes is None:
        kinds = header_prototypes()[name]
        fn.argtypes = [ctypes.c_void_p if k == "p" else (ctypes.c_double if k == "f" else ctypes.c_uint64) for k in kinds]
    args = []
    for t in fn.argtypes or []:
        if t in (ctypes.c_void_p, ctypes.c_char_p) or hasattr(t, "contents") or getattr(t, "_type_", None) == "P":
            args.append(None)
        elif t in (ctypes.c_double, ctypes.c_float):
            args.append(0.0)
        else:
            args.append(0)
    fn(*args)
    return 0


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        sys.exit(child(sys.argv[2]))
    lib, names = declared()
    protos = header_prototypes()
    skipped, crashed, ok = [], [], 0
    for n in names:
        fn = getattr(lib, n)
        if fn.argtypes is None and n not in protos:
            skipped.append(n); continue
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", n], capture_output=True, text=True, timeout=120)
        if r.returncode < 0 or r.returncode in (134, 139):
            crashed.append(n); print("CRASH", n, r.returncode, flush=True)
        else:
            ok += 1
    print("null_arg_sweep: %d exported, %d called, %d crashed, %d without a prototype in the binding or the headers" % (len(names), ok + len(crashed), len(crashed), len(skipped)))
    sys.exit(1 if crashed else 0)


if __name__ == "__main__":
    main()
