#!/bin/bash
# headline with and without hash hints, alternated inside one gpurun call (boxes differ by several per cent)
set -o pipefail
O=gpurun_out/abh; mkdir -p $O
for i in 1 2 3; do for h in 0 1; do
  python bench.py --steps 40 --warmup 5 --no-tree --no-ntt --no-cpu-baseline --headline-only --hash-hints $h > $O/h${h}_$i.json 2> $O/h${h}_$i.err || { tail -5 $O/h${h}_$i.err; exit 1; }
  python3 -c "
import json; j=json.loads([l for l in open('$O/h${h}_$i.json') if l.startswith('{')][-1]); print('hints', $h, 'round', $i, j['value'], j.get('window_proofs_per_s'))"
done; done
