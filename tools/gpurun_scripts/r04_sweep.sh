#!/bin/bash
# worker / lockstep sweep of the commit+prove headline (bench.py --headline-only): one line per configuration
set -o pipefail
O=gpurun_out/r04_sweep; mkdir -p $O
for cfg in "6 32" "7 32" "8 32" "6 48" "8 24" "5 40" "6 32"; do
  set -- $cfg
  python bench.py --streams $1 --batch $2 --steps 20 --warmup 3 --no-tree --no-ntt --no-cpu-baseline --headline-only > $O/b_$1_$2.json 2> $O/b_$1_$2.err || { tail -3 $O/b_$1_$2.err; exit 1; }
  python -c "
import json; d=json.load(open('$O/b_$1_$2.json')); print('workers $1 lockstep $2:', d['value'], 'commit+prove;', d['prove_only_resident_witness']['proofs_per_s'], 'prove only; windows', d['window_proofs_per_s'])" | tee -a $O/summary.txt
done
