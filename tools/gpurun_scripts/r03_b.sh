#!/bin/bash
# round 3: the leaf-profile bench shape (Poseidon2 gate rows) — new tests, the default bench line, and the isolated per-kernel
# cost of one lockstep batch (one worker, profiled). A failing step ends the script.
set -o pipefail
O=gpurun_out/r03_b
mkdir -p $O
R=$GRAFT_REPO_ROOT
python -m pytest tests/test_multirank_gpu.py tests/test_poseidon2_gate_gpu.py tests/test_batch_gpu.py -m gpu -q -x > $O/pytest.txt 2>&1; rc=$?; echo "pytest rc=$rc" | tee -a $O/summary.txt; tail -5 $O/pytest.txt | tee -a $O/summary.txt
[ $rc -eq 0 ] || exit $rc
python bench.py > $O/bench.json 2> $O/bench.err; rc=$?; echo "bench rc=$rc" | tee -a $O/summary.txt
[ $rc -eq 0 ] || { tail -20 $O/bench.err; exit $rc; }
cd /tmp && export TMPDIR=/tmp
export QPGPU_CRASH_TRACE=$R/$O/crash_trace.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_single_worker -o sw -- python3 -X faulthandler $R/bench.py --streams 1 --batch 32 --steps 8 --warmup 2 --no-tree --no-ntt --no-cpu-baseline --headline-only > $R/$O/prof_single_worker.log 2>&1; rc=$?
echo "prof_single_worker rc=$rc" | tee -a $R/$O/summary.txt
cd $R
[ $rc -eq 0 ] || { tail -40 $O/prof_single_worker.log; cat $O/crash_trace.txt; find $O -name "*kernel_trace.csv" -delete; exit $rc; }
python tools/profile_summary.py $O/prof_single_worker $O/sum_single_worker "python3 bench.py --streams 1 --batch 32 --steps 8 --warmup 2 --no-tree --no-ntt --no-cpu-baseline --headline-only" >> $O/summary.txt 2>&1
find $O -name "*kernel_trace.csv" -delete
du -sh $O | tee -a $O/summary.txt
