#!/bin/bash
set -o pipefail
O=gpurun_out/r02_i
mkdir -p $O
python -m pytest tests/test_ntt_gpu.py tests/test_prove_gpu.py -m gpu -q -x > $O/pytest.txt 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -2 $O/pytest.txt | tee -a $O/summary.txt
for sp in 0 1; do
  QPGPU_NTT_SPLIT=$sp python tools/ntt_time.py "split$sp" >> $O/ntt_variants.jsonl 2>>$O/err.txt
done
cat $O/ntt_variants.jsonl | tee -a $O/summary.txt
python bench.py --steps 20 --warmup 3 --no-tree --no-ntt --no-cpu-baseline --headline-only > $O/b.json 2> $O/b.err
python -c "
import json
d=json.loads([l for l in open('$O/b.json') if l.startswith('{')][-1]); print('bench', d['value'], d['window_proofs_per_s'])" | tee -a $O/summary.txt
