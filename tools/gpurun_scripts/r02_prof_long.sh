# long multi-worker runs under rocprofv3 (600 steps = 115 200 proofs each): crash statistics of the profiled proving path
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_long
mkdir -p $O
for i in 1 2 3; do
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pl_$i -o bench -- python3 $R/bench.py --steps 600 --warmup 5 --no-tree --no-ntt --no-cpu-baseline --headline-only > $O/long_$i.log 2>&1; echo "long $i rc=$?" | tee -a $O/summary.txt
  grep -a "^{" $O/long_$i.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['window_proofs_per_s'])" | tee -a $O/summary.txt
  rm -rf /tmp/pl_$i
done
