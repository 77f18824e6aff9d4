#!/usr/bin/env python3
"""Static ISA histogram of the gfx950 kernels in one .hip file: compiles it device-only to assembly (hipcc -S) and counts
instruction mnemonics per kernel, grouped into the classes that matter for Goldilocks arithmetic (64x32 multiply-adds, carry
chains, selects, 64-bit adds, shifts, VMEM, LDS, scalar). Loops are counted once (static), so straight-line kernels (the NTT
passes are fully unrolled) read as dynamic counts per thread; kernels with loops need the trip counts applied by hand.
usage: isa_hist.py <file.hip> [kernel-name substring] [--json]"""
import collections, json, os, re, subprocess, sys, tempfile
src = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else ""
as_json = "--json" in sys.argv
asm = os.path.join(tempfile.gettempdir(), os.path.basename(src) + ".s")
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--offload-device-only", "-S", src, "-o", asm,
                       "-I", os.path.dirname(os.path.abspath(src))], stderr=subprocess.DEVNULL)
text = open(asm).read()
CLASSES = [("mad_u64_u32", r"^v_mad_u64_u32"), ("mul_32", r"^v_mul_(lo|hi)_u32|^v_mul_u32_u24|^v_mad_u32_u24"), ("carry_add_sub", r"^v_(add|sub|subrev)_co_|^v_(addc|subb|subbrev)_co_"),
           ("cndmask", r"^v_cndmask"), ("add64_lshl", r"^v_lshl_add_u64"), ("cmp", r"^v_cmp"), ("shift_align", r"^v_(lshlrev|lshrrev|ashrrev|alignbit|lshl_or|lshl_add|and_or|bfe)"),
           ("add_sub_32", r"^v_(add|sub|subrev)_u32|^v_add3"), ("logic_mov", r"^v_(and|or|xor|not|mov|bfi|perm|readlane|readfirstlane|writelane)"),
           ("vmem_load", r"^(global|buffer|flat)_load"), ("vmem_store", r"^(global|buffer|flat)_store"), ("lds", r"^ds_"), ("dpp_shuffle", r"^ds_bpermute|^v_.*dpp"),
           ("salu", r"^s_(?!waitcnt|barrier|endpgm|nop|branch|cbranch|setpc|swappc|getpc)"), ("waitcnt_barrier", r"^s_(waitcnt|barrier)"), ("branch", r"^s_(branch|cbranch|setpc|swappc)")]
out = {}
for m in re.finditer(r"^(_Z\S+):\s*;[^\n]*\n(.*?)\n\s*s_endpgm", text, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if sub and sub not in name:
        continue
    c = collections.Counter()
    for line in body.split("\n"):
        line = line.strip()
        if not line or line[0] in ".;/" or line.endswith(":"):
            continue
        c[line.split()[0]] += 1
    cls = collections.Counter()
    for op, n in c.items():
        for cname, pat in CLASSES:
            if re.match(pat, op):
                cls[cname] += n
                break
        else:
            cls["other:" + op] += n
    valu = sum(n for op, n in c.items() if op.startswith("v_"))
    meta = {}
    for key in ("num_vgpr", "private_seg_size", "numbered_sgpr"):
        mm = re.search(r"\.set " + re.escape(name) + r"\." + key + r", (\d+)", text)
        if mm:
            meta[key] = int(mm.group(1))
    out[name] = {"instructions": sum(c.values()), "valu": valu, "classes": dict(cls.most_common()), **meta}
if as_json:
    print(json.dumps(out, indent=1))
else:
    for name, v in out.items():
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
        print(f"{dem}\n  instructions {v['instructions']}  VALU {v['valu']}  VGPRs {v.get('num_vgpr')}  scratch {v.get('private_seg_size')}")
        for k, n in v["classes"].items():
            print(f"    {k:18s} {n}")
