#!/usr/bin/env python3
"""Registers, scratch, LDS and occupancy of every gfx950 kernel in the library, as the compiler reports them
(hipcc -Rpass-analysis=kernel-resource-usage). usage: tools/kernel_resources.py > profiles/rNN_kernel_resource_usage.txt"""
import os, re, subprocess, sys
CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "qp-zk-circuits_amd", "csrc")
FILES = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
KEYS = {"VGPRs": "vgpr", "AGPRs": "agpr", "SGPRs": "sgpr", "ScratchSize [bytes/lane]": "scratch", "Occupancy [waves/SIMD]": "occ", "LDS Size [bytes/block]": "lds"}
print(f"{'file':22s} {'kernel':72s} {'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'scratch B/lane':>15s} {'lds B':>7s} {'waves/SIMD':>10s}")
for f in FILES:
    p = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--offload-device-only", "-c", f, "-o", os.devnull,
                        "-Rpass-analysis=kernel-resource-usage"], cwd=CSRC, capture_output=True, text=True)
    rows, cur = [], None
    for line in p.stderr.split("\n"):
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}
            rows.append(cur)
            continue
        for k, short in KEYS.items():
            m = re.search(re.escape(k) + r": (\d+)", line)
            if m and cur is not None and "remark:" in line:
                cur[short] = int(m.group(1))
    names = subprocess.run(["/usr/bin/c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.split("\n")
    for r, nm in zip(rows, names):
        nm = re.sub(r"\(anonymous namespace\)::|qpgpu::", "", nm).split("(")[0].replace("void ", "")
        print(f"{f:22s} {nm[:72]:72s} {r.get('vgpr', 0):5d} {r.get('agpr', 0):5d} {r.get('sgpr', 0):5d} {r.get('scratch', 0):15d} {r.get('lds', 0):7d} {r.get('occ', 0):10d}")
