#!/bin/bash
set -o pipefail
O=gpurun_out/r02_final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_headline -o bench -- python3 $R/bench.py --steps 120 --warmup 5 --no-tree --no-ntt --no-cpu-baseline --headline-only > $R/$O/prof_headline.log 2>&1; echo "prof_headline rc=$?" | tee -a $R/$O/summary.txt
