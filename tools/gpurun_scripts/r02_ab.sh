#!/bin/bash
set -o pipefail
O=gpurun_out/r02_ab
mkdir -p $O
for cfg in "2 32 60" "4 32 40" "3 64 30" "4 64 30" "2 128 20" "6 64 20" "4 64 30"; do
set -- $cfg
python bench.py --steps $3 --warmup 3 --no-tree --no-ntt --no-cpu-baseline --headline-only --streams $1 --batch $2 > $O/b_$1_$2.json 2> $O/b_$1_$2.err
python -c "
import json
d=json.loads([l for l in open('$O/b_$1_$2.json') if l.startswith('{')][-1]); print('bench $1x$2', d['value'], d['window_proofs_per_s'])" | tee -a $O/summary.txt
done
