#!/bin/bash
set -o pipefail
O=gpurun_out/r02_h
mkdir -p $O
for sp in 1 2; do
QPGPU_NTT_SPLIT=$sp python -m pytest tests/test_ntt_gpu.py tests/test_prove_gpu.py -m gpu -q -x > $O/pytest_$sp.txt 2>&1; echo "pytest split=$sp rc=$?" | tee -a $O/summary.txt
tail -2 $O/pytest_$sp.txt | tee -a $O/summary.txt
done
for cfg in "0 -1" "1 -1" "1 2" "0 2"; do
  set -- $cfg
  QPGPU_NTT_SPLIT=$1 QPGPU_NTT_LOGT=$2 python tools/ntt_time.py "split$1_logt$2" >> $O/ntt_variants.jsonl 2>>$O/err.txt
done
cat $O/ntt_variants.jsonl | tee -a $O/summary.txt
for sp in 0 1 2; do
QPGPU_NTT_SPLIT=$sp python bench.py --steps 20 --warmup 3 --no-tree --no-ntt --no-cpu-baseline --headline-only > $O/b_$sp.json 2> $O/b_$sp.err
python -c "
import json
d=json.loads([l for l in open('$O/b_$sp.json') if l.startswith('{')][-1]); print('split$sp', d['value'], d['window_proofs_per_s'])" | tee -a $O/summary.txt
done
