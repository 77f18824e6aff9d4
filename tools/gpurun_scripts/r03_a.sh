#!/bin/bash
# round 3, first GPU call: parity tests on the rebuilt library, then ONE profiled run of the default bench command with the
# crash tracer armed (QPGPU_CRASH_TRACE + python -X faulthandler). No step is repeated: a non-zero rc ends the script, and
# its log stays where it is.
set -o pipefail
O=gpurun_out/r03_a
mkdir -p $O
R=$GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x > $O/pytest.txt 2>&1; rc=$?; echo "pytest rc=$rc" | tee -a $O/summary.txt; tail -5 $O/pytest.txt | tee -a $O/summary.txt
[ $rc -eq 0 ] || exit $rc
python bench.py > $O/bench.json 2> $O/bench.err; rc=$?; echo "bench rc=$rc" | tee -a $O/summary.txt
[ $rc -eq 0 ] || { tail -20 $O/bench.err; exit $rc; }
cd /tmp && export TMPDIR=/tmp
ulimit -c unlimited || true
export QPGPU_CRASH_TRACE=$R/$O/crash_trace_prof_bench.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_bench -o bench -- python3 -X faulthandler $R/bench.py > $R/$O/prof_bench.log 2>&1; rc=$?
echo "prof_bench rc=$rc" | tee -a $R/$O/summary.txt
cd $R
[ -s $O/crash_trace_prof_bench.txt ] && { echo "CRASH TRACE WRITTEN"; cat $O/crash_trace_prof_bench.txt | head -80; }
[ $rc -eq 0 ] || { tail -40 $O/prof_bench.log; find $O -name "*kernel_trace.csv" -delete; exit $rc; }
python tools/profile_summary.py $O/prof_bench $O/sum_bench "python3 bench.py (default command)" >> $O/summary.txt 2>&1
find $O -name "*kernel_trace.csv" -delete
grep -a "^{" $O/prof_bench.log | tail -1 > $O/bench_under_prof.json || true
du -sh $O | tee -a $O/summary.txt
