#!/bin/bash
# one worker, lockstep batch of 32: kernel-trace stats (every kernel alone on the device)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=gpurun_out/sw3; mkdir -p $R/$O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o sw -- python3 $R/bench.py --streams 1 --batch 32 --steps 8 --warmup 2 --no-tree --no-ntt --no-cpu-baseline --headline-only > $R/$O/prof.log 2>&1 || { tail -5 $R/$O/prof.log; exit 1; }
cd $R
python tools/profile_summary.py $O/prof $O/sum "single worker" > $O/summary.txt 2>&1
find $O -name "*kernel_trace.csv" -delete
head -40 $O/sum*kernel_stats.csv
