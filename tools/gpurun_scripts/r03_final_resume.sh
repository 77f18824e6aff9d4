#!/bin/bash
# The steps of tools/r03_final.sh from prof_headline on, for the call after a profiled step died (its logs stay where they are:
# this run writes prof_headline_run2.*). Same rules: every step once, crash tracer armed, stop on a non-zero return code.
set -o pipefail
O=gpurun_out/r03_final
mkdir -p $O
R=$GRAFT_REPO_ROOT
step() { echo "$1 rc=$2" | tee -a $R/$O/summary.txt; [ $2 -eq 0 ] || { echo "STOP: $1 failed"; find $R/$O -name "*kernel_trace.csv" -delete; exit $2; }; }
python tools/kernel_id.py ntt > $O/kernel_source_id.txt
echo "resumed after a failed prof_headline (profiles/r03_crash_trace_prof_headline.txt)" | tee -a $O/summary.txt
cd /tmp && export TMPDIR=/tmp
export QPGPU_CRASH_TRACE=$R/$O/crash_trace_prof_run2.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_bench -o bench -- python3 -X faulthandler $R/bench.py > $R/$O/prof_bench.log 2>&1; step prof_bench $?
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_headline -o bench -- python3 -X faulthandler $R/bench.py --steps 120 --warmup 5 --no-tree --no-ntt --no-cpu-baseline --headline-only > $R/$O/prof_headline_run2.log 2>&1; step prof_headline_run2 $?
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_single_worker -o sw -- python3 -X faulthandler $R/bench.py --streams 1 --batch 32 --steps 8 --warmup 2 --no-tree --no-ntt --no-cpu-baseline --headline-only > $R/$O/prof_single_worker.log 2>&1; step prof_single_worker $?
unset QPGPU_CRASH_TRACE
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_ntt -o ntt -- python3 $R/tools/ntt_only.py 40 > $R/$O/prof_ntt.log 2>&1; step prof_ntt $?
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $R/$O/pmc_sq1 -o ntt -- python3 $R/tools/ntt_only.py 3 > $R/$O/pmc_sq1.log 2>&1; step pmc_sq1 $?
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE -d $R/$O/pmc_sq2 -o ntt -- python3 $R/tools/ntt_only.py 3 > $R/$O/pmc_sq2.log 2>&1; step pmc_sq2 $?
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$O/pmc_fetch -o ntt -- python3 $R/tools/ntt_only.py 3 > $R/$O/pmc_fetch.log 2>&1; step pmc_fetch $?
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$O/pmc_write -o ntt -- python3 $R/tools/ntt_only.py 3 > $R/$O/pmc_write.log 2>&1; step pmc_write $?
cd $R
bash tools/gpurun_scripts/r03_mx_pmc.sh > $O/mx_pmc.log 2>&1; step mx_pmc $?
[ -s $O/crash_trace_prof_run2.txt ] && { echo "CRASH TRACE WRITTEN" | tee -a $O/summary.txt; head -60 $O/crash_trace_prof_run2.txt; }
python tools/profile_summary.py $O/prof_bench $O/sum_bench "python3 bench.py (default command)" >> $O/summary.txt 2>&1
python tools/profile_summary.py $O/prof_ntt $O/sum_ntt_only "python3 tools/ntt_only.py 40" >> $O/summary.txt 2>&1
python tools/profile_summary.py $O/prof_single_worker $O/sum_single_worker "python3 bench.py --streams 1 --batch 32 --steps 8 --warmup 2 --no-tree --no-ntt --no-cpu-baseline --headline-only" >> $O/summary.txt 2>&1
python tools/profile_summary.py $O/prof_headline $O/sum_headline "python3 bench.py --steps 120 --warmup 5 --no-tree --no-ntt --no-cpu-baseline --headline-only" >> $O/summary.txt 2>&1
find $O -name "*kernel_trace.csv" -delete
grep -a "^{" $O/prof_bench.log | tail -1 > $O/bench_under_prof.json || true
du -sh $O | tee -a $O/summary.txt
