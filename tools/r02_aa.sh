#!/bin/bash
set -o pipefail
O=gpurun_out/r02_aa
mkdir -p $O
python bench.py --steps 150 --warmup 5 --no-tree --no-ntt --no-cpu-baseline --headline-only > $O/b150.json 2> $O/b150.err; echo "bench150 rc=$?" | tee -a $O/summary.txt
python -c "
import json
d=json.loads([l for l in open('$O/b150.json') if l.startswith('{')][-1]); print('bench150', d['value'], d['window_proofs_per_s'])" | tee -a $O/summary.txt
python bench.py --steps 3 --warmup 1 --no-tree --no-ntt --no-cpu-baseline --headline-only > $O/b3.json 2> $O/b3.err; echo "bench3 rc=$?" | tee -a $O/summary.txt
python bench.py --steps 12 --warmup 2 --no-tree --no-ntt --no-cpu-baseline --headline-only --streams 4 --batch 64 > $O/b4x64.json 2> $O/b4x64.err; echo "bench4x64 rc=$?" | tee -a $O/summary.txt
python -c "
import json
d=json.loads([l for l in open('$O/b4x64.json') if l.startswith('{')][-1]); print('bench 4x64', d['value'])" | tee -a $O/summary.txt
QPGPU_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 40 --warmup 3 --no-tree --no-ntt --no-cpu-baseline --headline-only > $O/b2r.json 2> $O/b2r.err; echo "2rank rc=$?" | tee -a $O/summary.txt
python -c "
import json
d=json.loads([l for l in open('$O/b2r.json') if l.startswith('{')][-1]); print('2rank', d['value'], d['n_gpus'])" | tee -a $O/summary.txt
bash tools/r02_z.sh
