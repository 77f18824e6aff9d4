#!/bin/bash
# Final evidence of round 4 (one gpurun call; every profiled step runs ONCE, a non-zero return code ends the script where it is):
#   parity tests, smoke, the default bench line (what the driver runs), the leaf path from plain C (examples/leaf_prove_example.c),
#   kernel-trace stats of the default command, of the headline through the torch-free C driver (six workers) and of one worker,
#   of the NTT loop, the PMC passes of the 2^20 x 128 NTT kernels (SQ counters in two passes, FETCH_SIZE and WRITE_SIZE in passes of
#   their own, as MI355X_MICROARCH.md prescribes) and of the matrix-pipe leaf-hash kernel.
# Runtime stacks: no profiled process imports torch — bench.py on one rank, tools/ntt_only.py, tools/hash_probe.py
# and the C driver load the ROCm 7.2 libamdhip64 / libhsa-runtime64 that libqpgpu.so links and that rocprofv3 preloads; each step
# prints the libraries it ended up with (runtime_stacks.txt). The crash tracer stays armed (QPGPU_CRASH_TRACE).
# Output: gpurun_out/r04_final; tools/collect_evidence.py r04 -> profiles/.
set -o pipefail
O=gpurun_out/r04_final
mkdir -p $O
R=$GRAFT_REPO_ROOT
step() { echo "$1 rc=$2" | tee -a $R/$O/summary.txt; [ $2 -eq 0 ] || { echo "STOP: $1 failed"; find $R/$O -name "*kernel_trace.csv" -delete; exit $2; }; }
python tools/kernel_id.py ntt > $O/kernel_source_id.txt
python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; rc=$?; tail -3 $O/pytest.txt | tee -a $O/summary.txt; step pytest $rc
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; step smoke $?
python bench.py > $O/bench.json 2> $O/bench.err; step bench $?
grep -a "runtime stack" $O/bench.err | sed 's/^/bench.py (unprofiled): /' >> $O/runtime_stacks.txt
QPGPU_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 10 --warmup 2 --no-ntt --no-cpu-baseline --headline-only --no-tree > $O/bench_2rank_gloo.json 2> $O/bench_2rank_gloo.err; step bench_2rank_gloo $?
gcc -O2 -I include examples/leaf_prove_example.c -L qp-zk-circuits_amd -lqpgpu -lpthread -Wl,-rpath,$R/qp-zk-circuits_amd -o $O/leaf_driver; step build_leaf_driver $?
$O/leaf_driver 13 0 6 32 10 > $O/leaf_driver_unprofiled.txt 2>&1; step leaf_driver_unprofiled $?
cd /tmp && export TMPDIR=/tmp
export QPGPU_CRASH_TRACE=$R/$O/crash_trace_prof_bench.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_bench -o bench -- python3 -X faulthandler $R/bench.py > $R/$O/prof_bench.log 2>&1; step prof_bench $?
grep -a "runtime stack" $R/$O/prof_bench.log | sed 's/^/bench.py under rocprofv3: /' >> $R/$O/runtime_stacks.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_headline -o lp -- $R/$O/leaf_driver 13 0 6 32 40 > $R/$O/leaf_driver_headline.txt 2>&1; step prof_headline $?
grep -a "^runtime:" $R/$O/leaf_driver_headline.txt | sed 's/^/leaf_driver under rocprofv3: /' >> $R/$O/runtime_stacks.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_single_worker -o lp -- $R/$O/leaf_driver 13 0 1 32 8 > $R/$O/leaf_driver_single_worker.txt 2>&1; step prof_single_worker $?
unset QPGPU_CRASH_TRACE
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_ntt -o ntt -- python3 $R/tools/ntt_only.py 40 > $R/$O/prof_ntt.log 2>&1; step prof_ntt $?
grep -a "runtime stack" $R/$O/prof_ntt.log | sed 's/^/ntt_only.py under rocprofv3: /' >> $R/$O/runtime_stacks.txt
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $R/$O/pmc_sq1 -o ntt -- python3 $R/tools/ntt_only.py 3 > $R/$O/pmc_sq1.log 2>&1; step pmc_sq1 $?
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE -d $R/$O/pmc_sq2 -o ntt -- python3 $R/tools/ntt_only.py 3 > $R/$O/pmc_sq2.log 2>&1; step pmc_sq2 $?
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$O/pmc_fetch -o ntt -- python3 $R/tools/ntt_only.py 3 > $R/$O/pmc_fetch.log 2>&1; step pmc_fetch $?
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$O/pmc_write -o ntt -- python3 $R/tools/ntt_only.py 3 > $R/$O/pmc_write.log 2>&1; step pmc_write $?
cd $R
bash tools/gpurun_scripts/mx_pmc.sh > $O/mx_pmc.log 2>&1; step mx_pmc $?
[ -s $O/crash_trace_prof_bench.txt ] && { echo "CRASH TRACE WRITTEN" | tee -a $O/summary.txt; head -60 $O/crash_trace_prof_bench.txt; }
python tools/profile_summary.py $O/prof_bench $O/sum_bench "python3 bench.py (default command)" >> $O/summary.txt 2>&1
python tools/profile_summary.py $O/prof_ntt $O/sum_ntt_only "python3 tools/ntt_only.py 40" >> $O/summary.txt 2>&1
python tools/profile_summary.py $O/prof_single_worker $O/sum_single_worker "leaf_prove_example 13 0 1 32 8 (one worker, lockstep 32)" >> $O/summary.txt 2>&1
python tools/profile_summary.py $O/prof_headline $O/sum_headline "leaf_prove_example 13 0 6 32 40 (the headline's pool from plain C)" >> $O/summary.txt 2>&1
find $O -name "*kernel_trace.csv" -delete
grep -a "^{" $O/prof_bench.log | tail -1 > $O/bench_under_prof.json || true
cat $O/runtime_stacks.txt | tee -a $O/summary.txt
du -sh $O | tee -a $O/summary.txt
