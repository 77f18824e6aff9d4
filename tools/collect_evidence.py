#!/usr/bin/env python3
"""Turns the output of the round's evidence script (tools/r04_final.sh -> gpurun_out/<round>_final) into the tracked files under profiles/
(prefix <round>_final_; round 3's files were made by this script under its earlier name tools/r03_collect.py):
  _bench.json / _bench_2rank_gloo.json / _bench_6rank_gloo.json   bench lines (default command; self-launched gloo rehearsals)
  _bench_kernel_stats.csv, _bench_ntt_2p20_launches.json          rocprofv3 --kernel-trace --stats of the default command
  _ntt_only_*, _headline_*, _single_worker_*                       the same for the NTT loop alone, the headline leg, one worker
  _ntt_pmc_summary.json      HBM bytes per transform (FETCH_SIZE / WRITE_SIZE passes), with the NTT kernels' source identity
  _ntt_valu_summary.json     VALU instructions per element, issue-slot share, LDS bank conflicts, and the VALU ROOFLINE of the
                             two 2^20 kernels: their instruction mix priced with the per-class issue costs measured on this
                             chip at their occupancy (profiles/r01_b_valu_rates.txt, 4 waves per SIMD), against measured cycles
  _ntt_isa_hist.json         static instruction mix of the compiled NTT kernels (tools/isa_hist.py)
Every counter summary carries kernel_source_id (tools/kernel_id.py): bench.py reports its figures only while the id matches.
usage: python tools/collect_evidence.py <round, e.g. r04> [src_dir]"""
import glob, json, os, re, shutil, sqlite3, subprocess, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from kernel_id import kernel_source_id  # noqa: E402

ROUND = sys.argv[1] if len(sys.argv) > 1 else "r04"
SRC = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", ROUND + "_final")
DST = os.path.join(ROOT, "profiles")
PRE = ROUND + "_final"
PY = sys.executable
N_ELEMS = (1 << 20) * 128
SIMDS, XCDS = 256 * 4, 8
# the identity of the library the box measured: written by the evidence script on the box (the tree here may have moved on)
_idf = os.path.join(SRC, "kernel_source_id.txt")
if os.path.exists(_idf):
    KID = open(_idf).read().strip()
else:   # older runs: the bench line of the same run carries it
    try:
        KID = json.loads([l for l in open(os.path.join(SRC, "bench.json")).read().strip().split("\n") if l.startswith("{")][-1])["roofline"]["kernel_source_id"]
    except Exception:
        KID = kernel_source_id("ntt")


def last_json_line(path):
    for line in reversed(open(path).read().strip().split("\n")):
        if line.startswith("{"):
            return json.loads(line)
    raise SystemExit(f"no JSON line in {path}")


for name in ("bench", "bench_2rank_gloo", "bench_6rank_gloo"):
    p = os.path.join(SRC, name + ".json")
    if os.path.exists(p) and os.path.getsize(p):
        json.dump(last_json_line(p), open(os.path.join(DST, f"{PRE}_{name}.json"), "w"), indent=1)
        print("wrote", f"{PRE}_{name}.json")

for prefix, out in (("sum_bench", "bench"), ("sum_ntt_only", "ntt_only"), ("sum_headline", "headline"), ("sum_single_worker", "single_worker")):
    for suffix in ("_kernel_stats.csv", "_ntt_2p20_launches.json"):
        src = os.path.join(SRC, prefix + suffix)
        if os.path.exists(src) and os.path.getsize(src) > 2 and not (prefix in ("sum_headline", "sum_single_worker") and suffix.endswith(".json")):
            shutil.copy(src, os.path.join(DST, f"{PRE}_{out}{suffix}"))
            print("copied", f"{PRE}_{out}{suffix}")

if os.path.isdir(os.path.join(SRC, "pmc_fetch")) and os.path.isdir(os.path.join(SRC, "pmc_write")):
    dst = os.path.join(DST, f"{PRE}_ntt_pmc_summary.json")
    subprocess.check_call([PY, os.path.join(ROOT, "tools", "pmc_summary.py"), os.path.join(SRC, "pmc_fetch"), os.path.join(SRC, "pmc_write"), dst])
    j = json.load(open(dst)); j["kernel_source_id"] = KID
    json.dump(j, open(dst, "w"), indent=1)

# static instruction mix of the NTT kernels (fully unrolled: the mix of the executed path differs only by the untaken
# coset / twiddle-table branches)
hist = subprocess.run([PY, os.path.join(ROOT, "tools", "isa_hist.py"), os.path.join(ROOT, "qp-zk-circuits_amd", "csrc", "ntt_inst_3.hip"), "--json"],
                      capture_output=True, text=True)
isa = {}
if hist.returncode == 0 and hist.stdout.strip():
    open(os.path.join(DST, f"{PRE}_ntt_isa_hist.json"), "w").write(hist.stdout)
    isa = json.loads(hist.stdout)
    print("wrote", f"{PRE}_ntt_isa_hist.json")

# measured issue cost per instruction class, cycles per wave-instruction per SIMD when the class saturates the SIMD at 4 waves
# per SIMD (the NTT kernels' occupancy): profiles/r01_b_valu_rates.txt
COST = {}
for line in open(os.path.join(DST, "r01_b_valu_rates.txt")):
    m = re.match(r"(.+?)\s+waves/SIMD 4\s+.*=>\s+([\d.]+) cycles/instr", line)
    if m:
        COST[m.group(1).strip()] = float(m.group(2))
CLASS_COST = {"mad_u64_u32": COST.get("v_mad_u64_u32", 8.4), "mul_32": COST.get("v_mul_lo_u32", 4.6), "carry_add_sub": COST.get("v_add_co + v_addc_co pair", 4.9),
              "cndmask": COST.get("v_cmp + v_cndmask", 3.5), "cmp": COST.get("v_cmp + v_cndmask", 3.5), "add64_lshl": COST.get("v_lshl_add_u64", 4.4),
              "shift_align": COST.get("v_lshlrev_b64", 4.4), "add_sub_32": COST.get("v_add_u32", 2.8), "logic_mov": COST.get("v_xor_b32", 2.7)}
VALU_CLASSES = set(CLASS_COST)


def mix_cost(kernel_substr):
    """(average modelled issue cycles per VALU wave-instruction, class shares) of the compiled kernel whose name holds the substring"""
    for name, e in isa.items():
        if kernel_substr in name:
            cls = {k: v for k, v in e["classes"].items() if k in VALU_CLASSES or k.startswith("other:v_")}
            tot = sum(cls.values())
            avg = sum(n * CLASS_COST.get(k, 4.4) for k, n in cls.items()) / tot
            return avg, {k: round(n / tot, 3) for k, n in sorted(cls.items(), key=lambda kv: -kv[1])[:8]}
    return None, None


per = defaultdict(lambda: defaultdict(list))
meta = {}
for db_path in sorted(glob.glob(os.path.join(SRC, "pmc_sq*", "**", "*results.db"), recursive=True)):
    db = sqlite3.connect(db_path)
    for kname, cname, val, dur, vgpr, lds in db.execute(
            "select kernel_name, counter_name, value, duration, vgpr_count, lds_block_size from counters_collection"):
        if "_kernel<5, 5" not in kname:
            continue
        key = kname.split("(anonymous namespace)::")[-1].split("(")[0]
        per[key][cname].append(val)
        per[key]["_dur_" + os.path.basename(os.path.dirname(db_path))].append(dur)
        # (rocprofv3's vgpr_count column reads 64 for these kernels; the compiler's resource report — 128 VGPRs, 0 AGPRs,
        # profiles/*kernel_resource_usage.txt and the isa histogram's num_vgpr — is what the occupancy follows)
        meta[key] = {"vgpr_count_column_of_rocprofv3": vgpr, "lds_bytes": lds}
if per:
    kernels = {}
    for key, cs in sorted(per.items()):
        e = dict(meta[key])
        for c, v in cs.items():
            if c.startswith("_dur_"):
                e["duration_us_under_" + c[5:]] = round(sum(v) / len(v) / 1e3, 1)
            else:
                e[c] = round(sum(v) / len(v), 1)
        inv, rows = [s.strip() == "true" for s in key.split("<")[1].rstrip(">").split(",")[2:4]]
        e["pass"] = ("inverse " if inv else "forward ") + ("rows" if rows else "strided")
        wave_insts = e.get("SQ_INSTS_VALU")
        if wave_insts:
            e["valu_insts_per_element"] = round(wave_insts * 64 / N_ELEMS, 1)
            cyc = e.get("GRBM_GUI_ACTIVE")
            if cyc:
                e["gpu_cycles"] = round(cyc / XCDS)     # the counter is summed over the 8 XCDs
                e["valu_issue_frac"] = round(wave_insts * 4 / (SIMDS * cyc / XCDS), 3)
                e["measured_cycles_per_valu_inst_per_simd"] = round(SIMDS * cyc / XCDS / wave_insts, 3)
                mangled = "ntt_pass_split_kernelILi5ELi5ELb%dELb%dEE" % (1 if inv else 0, 1 if rows else 0)
                for nm, ee in isa.items():
                    if mangled in nm:
                        e["vgpr_compiler"] = ee.get("num_vgpr")
                avg, shares = mix_cost(mangled)
                if avg:
                    e["valu_roofline"] = {"modelled_cycles_per_valu_inst": round(avg, 3), "instruction_mix": shares,
                                          "modelled_issue_cycles": round(wave_insts * avg / SIMDS), "measured_cycles": e["gpu_cycles"],
                                          "frac": round(wave_insts * avg / SIMDS / (cyc / XCDS), 3)}
        if e.get("SQ_LDS_IDX_ACTIVE"):
            e["lds_bank_conflict_share"] = round(e.get("SQ_LDS_BANK_CONFLICT", 0.0) / e["SQ_LDS_IDX_ACTIVE"], 3)
        if e.get("SQ_WAVE_CYCLES"):
            e["wait_any_share_of_wave_cycles"] = round(e.get("SQ_WAIT_ANY", 0.0) / e["SQ_WAVE_CYCLES"], 3)
        kernels[key] = e
    fwd = [e for e in kernels.values() if e["pass"].startswith("forward")]
    out = {"source": "rocprofv3 --kernel-trace --pmc <SQ set 1 | SQ set 2 + GRBM_GUI_ACTIVE> -- python3 tools/ntt_only.py 3 (the evidence script), MI355X",
           "kernel_source_id": KID, "workload": "2^20 points x 128 columns", "elements_per_transform": N_ELEMS,
           "definitions": {"valu_insts_per_element": "SQ_INSTS_VALU (wave instructions) x 64 lanes / (2^20 x 128 elements), summed over the strided and the rows launch of one forward transform",
                           "valu_issue_frac": "SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8): a CONVENTION (one wave64 instruction per SIMD per 4 cycles), not a ceiling",
                           "valu_roofline": "the VALU bound with measured prices: every class of the kernel's compiled instruction mix (tools/isa_hist.py) costs what it cost when it alone "
                                            "saturated a SIMD at 4 waves per SIMD (tools/valu_rates.hip, profiles/r01_b_valu_rates.txt: v_mad_u64_u32 8.4 cycles, carry adds 4.9, 64-bit adds "
                                            "and shifts 4.4, compares / selects 3.5, 32-bit adds 2.8, moves / logic 2.7); modelled_issue_cycles = SQ_INSTS_VALU x that average / 1024 SIMDs; "
                                            "frac = modelled / measured cycles of the launch (1.0 = the launch takes exactly its instructions' issue time)"},
           "kernels": kernels}
    if len(fwd) == 2 and all("valu_insts_per_element" in e for e in fwd):
        out["valu_insts_per_element"] = round(sum(e["valu_insts_per_element"] for e in fwd), 1)
        if all("GRBM_GUI_ACTIVE" in e for e in fwd):
            out["valu_issue_frac"] = round(sum(e["SQ_INSTS_VALU"] for e in fwd) * 4 / (SIMDS * sum(e["GRBM_GUI_ACTIVE"] for e in fwd) / XCDS), 3)
        if all("valu_roofline" in e for e in fwd):
            out["valu_roofline_frac"] = round(sum(e["valu_roofline"]["modelled_issue_cycles"] for e in fwd) / sum(e["valu_roofline"]["measured_cycles"] for e in fwd), 3)
    json.dump(out, open(os.path.join(DST, f"{PRE}_ntt_valu_summary.json"), "w"), indent=1)
    print("wrote", f"{PRE}_ntt_valu_summary.json", out.get("valu_insts_per_element"), out.get("valu_issue_frac"), out.get("valu_roofline_frac"))

for extra in ("crash_trace_prof_bench.txt", "summary.txt", "poseidon_microbench.txt", "runtime_stacks.txt", "leaf_driver_headline.txt", "leaf_driver_single_worker.txt"):
    p = os.path.join(SRC, extra)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(DST, f"{PRE}_{extra}"))
