#!/bin/bash
# full-table inter-pass twiddle (QPGPU_NTT_TW=3, default) against the running product (=1): NTT parity tests, the 2^20 x 128 transform and
# the headline, alternated inside one call
set -o pipefail
O=gpurun_out/r04_ntt_ab; mkdir -p $O
python -m pytest tests/test_ntt_gpu.py tests/test_prove_gpu.py -m gpu -q 2>&1 | tail -2 | tee -a $O/summary.txt
for v in 1 3 1 3; do QPGPU_NTT_TW=$v python tools/ntt_ab.py tw$v | tee -a $O/summary.txt; done
for v in 1 3 1 3; do
  QPGPU_NTT_TW=$v python bench.py --steps 30 --warmup 3 --no-tree --no-ntt --no-cpu-baseline --headline-only > $O/b.json 2> $O/b.err || { tail -3 $O/b.err; exit 1; }
  python -c "
import json; d=json.load(open('$O/b.json')); print('QPGPU_NTT_TW=$v headline:', d['value'], 'commit+prove;', d['prove_only_resident_witness']['proofs_per_s'], 'prove only')" | tee -a $O/summary.txt
done
