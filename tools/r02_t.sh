#!/bin/bash
set -o pipefail
O=gpurun_out/r02_t
mkdir -p $O
python -m pytest tests/test_ntt_gpu.py tests/test_prove_gpu.py tests/test_batch_gpu.py tests/test_staged_gpu.py -m gpu -q -x > $O/pytest.txt 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -3 $O/pytest.txt | tee -a $O/summary.txt
for sp in 0 1 0 1; do
QPGPU_NTT_SPARSE=$sp python bench.py --steps 25 --warmup 4 --no-tree --no-ntt --no-cpu-baseline --headline-only > $O/b_$sp.json 2> $O/b_$sp.err
python -c "
import json
d=json.loads([l for l in open('$O/b_$sp.json') if l.startswith('{')][-1]); print('sparse$sp', d['value'], d['window_proofs_per_s'])" | tee -a $O/summary.txt
done
python tools/big_proof.py 16 --routed 60 --zk > $O/big16.txt 2>&1; tail -4 $O/big16.txt | tee -a $O/summary.txt
QPGPU_NTT_SPARSE=0 python tools/big_proof.py 16 --routed 60 --zk > $O/big16_0.txt 2>&1; tail -4 $O/big16_0.txt | tee -a $O/summary.txt
