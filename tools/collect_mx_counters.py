#!/usr/bin/env python3
"""profiles/<round>_mx_leaf_hash_counters.json (usage: collect_mx_counters.py <round>) from the rocprofv3 --pmc passes of tools/gpurun_scripts/mx_pmc.sh
(gpurun_out/mx_pmc/{mx,tp}.json + kernel_source_id.txt): per-permutation instruction counts of the matrix-pipe leaf-hash kernel,
the throughput build beside it."""
import glob, json, os, sqlite3, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(ROOT, "gpurun_out", "mx_pmc")
mx = json.load(open(os.path.join(O, "mx.json")))
tp = json.load(open(os.path.join(O, "tp.json")))
kid = open(os.path.join(O, "kernel_source_id.txt")).read().strip()
PERMS = 17 << 21


def derive(k, perms):
    d = dict(k)
    cyc = k["GRBM_GUI_ACTIVE"] / 8
    d["valu_insts_per_permutation"] = round(k["SQ_INSTS_VALU"] * 64 / perms, 1)
    d["cycles_per_valu_inst_per_simd"] = round(1024 * cyc / k["SQ_INSTS_VALU"], 3)
    d["wait_share"] = round(k["SQ_WAIT_INST_ANY"] / k["SQ_WAVE_CYCLES"], 3)
    if "SQ_INSTS_MFMA" in k:
        d["mfma_insts_per_permutation"] = round(k["SQ_INSTS_MFMA"] * 64 / perms, 1)
        d["mfma_busy_cycles_per_mfma"] = round(k["SQ_VALU_MFMA_BUSY_CYCLES"] / k["SQ_INSTS_MFMA"], 1)
        d["mfma_busy_frac"] = round(k["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc), 4)
    d["G_permutations_per_s_under_pmc"] = round(perms / k["duration_ns_under_pmc"], 3)
    return d


# only the launches of the probe's tree (2^21 threads): the device self-test at context set-up launches the same kernel on a small
# input since round 4, and an average over all dispatches of the name would mix it in
mk = {}
for sub in ("p1", "p2"):
    for path in glob.glob(os.path.join(O, sub, "**", "*.db"), recursive=True):
        db = sqlite3.connect(path)
        for cname, val, dur in db.execute("select counter_name, avg(value), avg(duration) from counters_collection where kernel_name like "
                                          "'%mx::leaf_hash_kernel<mx::PoseidonV1>%' and grid_size = 2097152 group by counter_name"):
            mk[cname] = val; mk["duration_ns_under_pmc"] = dur
m = derive(mk, PERMS)
# the throughput build's kernel names collapse in pmc_db_summary.py (template arguments in anonymous namespaces): read its pass directly
tk = {}
for path in glob.glob(os.path.join(O, "p3", "**", "*.db"), recursive=True):
    db = sqlite3.connect(path)
    for cname, val, dur in db.execute("select counter_name, avg(value), avg(duration) from counters_collection where kernel_name like "
                                      "'%tp::%leaf_hash_kernel%' and grid_size = 2097152 group by counter_name"):
        tk[cname] = val; tk["duration_ns_under_pmc"] = dur
t = derive(tk, PERMS)
out = {"source": "rocprofv3 --kernel-trace --pmc ... -- python3 tools/hash_probe.py (tools/gpurun_scripts/mx_pmc.sh), MI355X; "
                 "2^21 leaves x 135 columns = 17 permutations per leaf (one lockstep batch's wires commitment)",
       "kernel_source_id": kid,
       "definitions": {"valu_insts_per_permutation": "SQ_INSTS_VALU (wave instructions) x 64 lanes / permutations of the launch",
                       "mfma_insts_per_permutation": "SQ_INSTS_MFMA x 64 / permutations (a wave instruction serves 64 permutations: 120 per wave)",
                       "mfma_busy_frac": "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)",
                       "cycles_per_valu_inst_per_simd": "1024 SIMDs x (GRBM_GUI_ACTIVE / 8 XCDs) / SQ_INSTS_VALU"},
       "valu_insts_per_permutation": m["valu_insts_per_permutation"], "mfma_insts_per_permutation": m["mfma_insts_per_permutation"],
       "mfma_busy_frac": m["mfma_busy_frac"],
       "mx::leaf_hash_kernel": m, "tp::leaf_hash_kernel<PoseidonV1> (QPGPU_MX=0)": t}
json.dump(out, open(os.path.join(ROOT, "profiles", (sys.argv[1] if len(sys.argv) > 1 else "r04") + "_mx_leaf_hash_counters.json"), "w"), indent=1)
print(json.dumps({k: out[k] for k in ("valu_insts_per_permutation", "mfma_insts_per_permutation", "mfma_busy_frac")}),
      m["cycles_per_valu_inst_per_simd"], m["G_permutations_per_s_under_pmc"], "| tp:", t["valu_insts_per_permutation"], t["cycles_per_valu_inst_per_simd"], t["G_permutations_per_s_under_pmc"])
