#!/bin/bash
set -o pipefail
O=gpurun_out/r02_l
mkdir -p $O
python -m pytest tests/test_batch_gpu.py tests/test_prove_gpu.py tests/test_staged_gpu.py tests/test_multirank_gpu.py -m gpu -q -x > $O/pytest.txt 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -2 $O/pytest.txt | tee -a $O/summary.txt
for cfg in "2 32" "2 64" "3 32"; do
set -- $cfg
python bench.py --steps 20 --warmup 3 --no-tree --no-ntt --no-cpu-baseline --headline-only --streams $1 --batch $2 > $O/b_$1_$2.json 2> $O/b_$1_$2.err
python -c "
import json
d=json.loads([l for l in open('$O/b_$1_$2.json') if l.startswith('{')][-1]); print('bench $1x$2', d['value'], d['window_proofs_per_s'])" | tee -a $O/summary.txt
done
