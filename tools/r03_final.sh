#!/bin/bash
# Final evidence of round 3: parity tests, smoke, the default bench line, two self-launched gloo rehearsals (2 and 6 ranks on the
# box's one GPU), kernel-trace stats of the default command / the NTT loop / the headline leg / one worker, the PMC passes of the
# 2^20 x 128 NTT kernels (SQ counters in two passes, FETCH_SIZE and WRITE_SIZE in passes of their own, as MI355X_MICROARCH.md
# prescribes). Every profiled step runs ONCE with the crash tracer armed (QPGPU_CRASH_TRACE + python -X faulthandler); a
# non-zero return code ends the script where it is, logs in place. Output: gpurun_out/r03_final; tools/r03_collect.py -> profiles/.
set -o pipefail
O=gpurun_out/r03_final
mkdir -p $O
R=$GRAFT_REPO_ROOT
step() { echo "$1 rc=$2" | tee -a $R/$O/summary.txt; [ $2 -eq 0 ] || { echo "STOP: $1 failed"; find $R/$O -name "*kernel_trace.csv" -delete; exit $2; }; }
python tools/kernel_id.py ntt > $O/kernel_source_id.txt
python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; rc=$?; tail -3 $O/pytest.txt | tee -a $O/summary.txt; step pytest $rc
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; step smoke $?
python bench.py > $O/bench.json 2> $O/bench.err; step bench $?
QPGPU_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 10 --warmup 2 --no-ntt --no-cpu-baseline --headline-only --batch-degree-bits 13 > $O/bench_2rank_gloo.json 2> $O/bench_2rank_gloo.err; step bench_2rank_gloo $?
# six ranks (the most that may share this box's GPU) with a third of the per-rank load: sizes the root's collector
QPGPU_BENCH_BACKEND=gloo python bench.py --gpus 6 --streams 2 --batch 16 --steps 12 --warmup 2 --no-ntt --no-cpu-baseline --no-tree --headline-only > $O/bench_6rank_gloo.json 2> $O/bench_6rank_gloo.err; step bench_6rank_gloo $?
cd /tmp && export TMPDIR=/tmp
export QPGPU_CRASH_TRACE=$R/$O/crash_trace_prof_bench.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_bench -o bench -- python3 -X faulthandler $R/bench.py > $R/$O/prof_bench.log 2>&1; step prof_bench $?
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_headline -o bench -- python3 -X faulthandler $R/bench.py --steps 120 --warmup 5 --no-tree --no-ntt --no-cpu-baseline --headline-only > $R/$O/prof_headline.log 2>&1; step prof_headline $?
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_single_worker -o sw -- python3 -X faulthandler $R/bench.py --streams 1 --batch 32 --steps 8 --warmup 2 --no-tree --no-ntt --no-cpu-baseline --headline-only > $R/$O/prof_single_worker.log 2>&1; step prof_single_worker $?
unset QPGPU_CRASH_TRACE
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_ntt -o ntt -- python3 $R/tools/ntt_only.py 40 > $R/$O/prof_ntt.log 2>&1; step prof_ntt $?
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $R/$O/pmc_sq1 -o ntt -- python3 $R/tools/ntt_only.py 3 > $R/$O/pmc_sq1.log 2>&1; step pmc_sq1 $?
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE -d $R/$O/pmc_sq2 -o ntt -- python3 $R/tools/ntt_only.py 3 > $R/$O/pmc_sq2.log 2>&1; step pmc_sq2 $?
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$O/pmc_fetch -o ntt -- python3 $R/tools/ntt_only.py 3 > $R/$O/pmc_fetch.log 2>&1; step pmc_fetch $?
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$O/pmc_write -o ntt -- python3 $R/tools/ntt_only.py 3 > $R/$O/pmc_write.log 2>&1; step pmc_write $?
cd $R
bash tools/gpurun_scripts/r03_mx_pmc.sh > $O/mx_pmc.log 2>&1; step mx_pmc $?
[ -s $O/crash_trace_prof_bench.txt ] && { echo "CRASH TRACE WRITTEN" | tee -a $O/summary.txt; head -60 $O/crash_trace_prof_bench.txt; }
python tools/profile_summary.py $O/prof_bench $O/sum_bench "python3 bench.py (default command)" >> $O/summary.txt 2>&1
python tools/profile_summary.py $O/prof_ntt $O/sum_ntt_only "python3 tools/ntt_only.py 40" >> $O/summary.txt 2>&1
python tools/profile_summary.py $O/prof_single_worker $O/sum_single_worker "python3 bench.py --streams 1 --batch 32 --steps 8 --warmup 2 --no-tree --no-ntt --no-cpu-baseline --headline-only" >> $O/summary.txt 2>&1
python tools/profile_summary.py $O/prof_headline $O/sum_headline "python3 bench.py --steps 120 --warmup 5 --no-tree --no-ntt --no-cpu-baseline --headline-only" >> $O/summary.txt 2>&1
find $O -name "*kernel_trace.csv" -delete
grep -a "^{" $O/prof_bench.log | tail -1 > $O/bench_under_prof.json || true
du -sh $O | tee -a $O/summary.txt
