#!/bin/bash
# NTT LDS pitch experiment: the 2^20 x 128 transform pair with the odd row pitch of round 2 (QPGPU_NTT_PITCH=0) and the
# conflict-free pitch (default), kernel durations from rocprofv3 and LDS counters from a --pmc pass; NTT parity tests first.
set -o pipefail
O=gpurun_out/r03_d
mkdir -p $O
R=$GRAFT_REPO_ROOT
python -m pytest tests/test_ntt_gpu.py tests/test_proof_targets_gpu.py -m gpu -q -x > $O/pytest.txt 2>&1; rc=$?; echo "pytest rc=$rc" | tee -a $O/summary.txt; tail -3 $O/pytest.txt
[ $rc -eq 0 ] || exit $rc
cd /tmp && export TMPDIR=/tmp
for P in 0 1; do
  export QPGPU_NTT_PITCH=$P
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_p$P -o ntt -- python3 $R/tools/ntt_only.py 20 > $R/$O/prof_p$P.log 2>&1; rc=$?; echo "prof pitch=$P rc=$rc" | tee -a $R/$O/summary.txt
  [ $rc -eq 0 ] || { tail -20 $R/$O/prof_p$P.log; exit $rc; }
  grep -h "ntt_pass" $R/$O/prof_p$P/*kernel_stats.csv | cut -c1-200 | tee -a $R/$O/summary.txt
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES --output-format csv -d $R/$O/pmc_p$P -o ntt -- python3 $R/tools/ntt_only.py 3 > $R/$O/pmc_p$P.log 2>&1; rc=$?; echo "pmc pitch=$P rc=$rc" | tee -a $R/$O/summary.txt
  [ $rc -eq 0 ] || { tail -20 $R/$O/pmc_p$P.log; exit $rc; }
  python3 - <<PY | tee -a $R/$O/summary.txt
import csv, glob, collections
f = glob.glob("$R/$O/pmc_p$P/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for path in f:
    for row in csv.DictReader(open(path)):
        k = row["Kernel_Name"]
        if "ntt_pass" not in k: continue
        k = k.split("(")[0][-60:]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); 
for k, d in acc.items():
    print("pitch=$P", k, {c: round(v / 1e6, 2) for c, v in d.items()}, "conflict/active = %.3f" % (d["SQ_LDS_BANK_CONFLICT"] / max(d["SQ_LDS_IDX_ACTIVE"], 1)), "wait_any/wave_cycles = %.3f" % (d["SQ_WAIT_ANY"] / max(d["SQ_WAVE_CYCLES"], 1)))
PY
  find $R/$O -name "*kernel_trace.csv" -delete; find $R/$O -name "*counter_collection.csv" -delete
done
