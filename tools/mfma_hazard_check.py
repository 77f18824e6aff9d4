#!/usr/bin/env python3
"""Build-time check of the matrix-pipe hashing kernels (csrc/merkle_kernels_mx.hip): between a v_mfma_* and the first instruction
that READS its destination, no other instruction may WRITE a register of that destination. With several waves per SIMD an MFMA
can still be in flight when the compiler reuses a register of its 16-register destination that it knows to be dead (a ds_read
into v[26:27] behind v_mfma ... v[16:31]): one correct wave per SIMD, every other wave wrong in all lanes
(profiles/r03_poseidon_mfma.txt item 3). poseidon_mfma.hpp keeps both destinations allocated until the chain has been issued;
this script makes a compiler that allocates differently fail the BUILD (csrc/Makefile runs it), not only the start-up self-test.

usage: mfma_hazard_check.py [--first-layout] [--excerpt FILE] [file.hip]
  --first-layout  compile with -DPMF_NO_KEEPALIVE (the layout the hazard was found in) — expected to FAIL; with --excerpt the
                  offending instruction and its surroundings are written to FILE
exit status 0 = no violation."""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = [a for a in sys.argv[1:]]
first = "--first-layout" in args
excerpt = args[args.index("--excerpt") + 1] if "--excerpt" in args else None
files = [a for a in args if a.endswith(".hip")]
src = files[0] if files else os.path.join(ROOT, "qp-zk-circuits_amd", "csrc", "merkle_kernels_mx.hip")
asm = os.path.join(tempfile.gettempdir(), os.path.basename(src) + (".first" if first else "") + ".hazard.s")
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--offload-device-only", "-S", src, "-o", asm, "-I", os.path.dirname(os.path.abspath(src))]
if first:
    cmd.insert(1, "-DPMF_NO_KEEPALIVE")
subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
text = open(asm).read()

REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
def regs(op):
    out = set()
    for m in REG.finditer(op):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out

NO_DEST = re.compile(r"^(global_store|buffer_store|flat_store|scratch_store|ds_write|ds_store|s_|v_cmp|v_cmpx|buffer_wbl2|global_wb|v_nop|ds_nop)")
violations, kernels, mfmas = [], 0, 0
for m in re.finditer(r"^(_Z\S+):\s*;[^\n]*\n(.*?)\n\s*s_endpgm", text, re.S | re.M):
    name, body = m.group(1), m.group(2).split("\n")
    kernels += 1
    open_ranges = []        # (frozenset of registers, line index of the mfma)
    for idx, raw in enumerate(body):
        line = raw.split(";")[0].strip()
        if not line or line[0] in "./":
            continue
        if line.endswith(":"):            # a label: control flow joins here; what was in flight has been consumed or is re-checked on the next pass
            open_ranges = []
            continue
        parts = line.split(None, 1)
        op, rest = parts[0], parts[1] if len(parts) > 1 else ""
        operands = [o.strip() for o in rest.split(",")]
        is_mfma = op.startswith("v_mfma")
        if NO_DEST.match(op):
            dst, srcs = set(), set().union(*[regs(o) for o in operands]) if operands else set()
        elif op.startswith("v_permlane") and "swap" in op:
            dst = regs(operands[0]) | regs(operands[1]); srcs = set(dst)
        else:
            dst = regs(operands[0]) if operands else set()
            srcs = set().union(*[regs(o) for o in operands[1:]]) if len(operands) > 1 else set()
        if is_mfma:
            mfmas += 1
            d = frozenset(dst)
            # an accumulate chain re-issues into the same destination: that is the chain, not a violation
            for r, at in list(open_ranges):
                if r != d and (r & dst):
                    violations.append((name, idx, at, body))
            open_ranges = [(r, at) for r, at in open_ranges if r != d] + [(d, idx)]
            continue
        # a reader of a destination closes it: from here on the value has been consumed under the hardware's own interlock
        still = []
        for r, at in open_ranges:
            if r & dst and not (r & srcs):
                violations.append((name, idx, at, body))
            elif not (r & srcs):
                still.append((r, at))
        open_ranges = still
print(f"{os.path.basename(src)}{' (-DPMF_NO_KEEPALIVE)' if first else ''}: {kernels} kernels, {mfmas} MFMA instructions, {len(violations)} write(s) into an unread MFMA destination")
if violations:
    name, idx, at, body = violations[0]
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    lines = [f"; {dem}", f"; first violation: line {idx} writes into the destination of the v_mfma at line {at} before anything has read it"]
    lo, hi = max(0, at - 3), min(len(body), idx + 6)
    for k in range(lo, hi):
        mark = ">>" if k == idx else ("MF" if k == at else "  ")
        lines.append(f"{mark} {body[k].rstrip()}")
    print("\n".join(lines))
    if excerpt:
        with open(excerpt, "w") as f:
            f.write("\n".join([f"tools/mfma_hazard_check.py {' '.join(sys.argv[1:])}", f"{len(violations)} violation(s) in {kernels} kernels; the first:"] + lines) + "\n")
sys.exit(1 if violations else 0)
