#!/bin/bash
set -o pipefail
O=gpurun_out/r02_r
mkdir -p $O
python bench.py --steps 20 --warmup 3 --no-tree --no-ntt --no-cpu-baseline > $O/b.json 2> $O/b.err; echo "bench rc=$?" | tee -a $O/summary.txt
tail -3 $O/b.err | tee -a $O/summary.txt
python -c "
import json
d=json.loads([l for l in open('$O/b.json') if l.startswith('{')][-1]); print('bench', d['value'], d['witness_generation'], d['end_to_end_with_witness_generation'])" | tee -a $O/summary.txt
timeout -k 10 700 python tools/fuzz_shapes.py 1200 77 12 > $O/fuzz.txt 2>&1; echo "fuzz rc=$?" | tee -a $O/summary.txt
tail -3 $O/fuzz.txt | tee -a $O/summary.txt
