# rocprofv3 --kernel-trace --stats of the default bench command (what roofline.avg_ms must agree with)
set -o pipefail
O=gpurun_out/r02_final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_bench -o bench -- python3 -X faulthandler $R/bench.py > $R/$O/prof_bench.log 2>&1; echo "prof rc=$?" | tee -a $R/$O/summary.txt
cd $R
python tools/profile_summary.py $O/prof_bench $O/sum_bench "python3 bench.py (default command)" >> $O/summary.txt 2>&1
find $O -name "*kernel_trace.csv" -delete
grep -a "^{" $O/prof_bench.log | tail -1 > $O/bench_under_prof.json || true
