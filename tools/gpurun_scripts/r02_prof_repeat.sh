# the two multi-worker commands under rocprofv3, several times over (crash statistics of the profiled runs; outputs discarded)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_repeat
mkdir -p $O
for i in 1 2 3 4; do
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pr_h$i -o bench -- python3 $R/bench.py --steps 60 --warmup 5 --no-tree --no-ntt --no-cpu-baseline --headline-only > $O/headline_$i.log 2>&1; echo "headline $i rc=$?" | tee -a $O/summary.txt
  rm -rf /tmp/pr_h$i
done
for i in 1 2; do
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pr_d$i -o bench -- python3 $R/bench.py --no-cpu-baseline > $O/default_$i.log 2>&1; echo "default $i rc=$?" | tee -a $O/summary.txt
  rm -rf /tmp/pr_d$i
done
