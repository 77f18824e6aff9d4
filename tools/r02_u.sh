#!/bin/bash
set -o pipefail
O=gpurun_out/r02_u
mkdir -p $O
python -m pytest tests/test_witness_gpu.py tests/test_batch_gpu.py tests/test_aggregation_gpu.py tests/test_multirank_gpu.py -m gpu -q -x > $O/pytest.txt 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -3 $O/pytest.txt | tee -a $O/summary.txt
for c in 0 1; do
QPGPU_WITNESS_COMBINED=$c python tools/tree_timing.py > $O/tree_$c.txt 2>&1; echo "tree combined=$c rc=$?" | tee -a $O/summary.txt
grep "levels\|commit" $O/tree_$c.txt | tee -a $O/summary.txt
done
python bench.py --steps 20 --warmup 3 --no-tree --no-ntt --no-cpu-baseline > $O/b.json 2> $O/b.err
python -c "
import json
d=json.loads([l for l in open('$O/b.json') if l.startswith('{')][-1]); print('bench', d['value'], d['witness_generation']['single_ms'], d['witness_generation']['batched_ms_per_witness'], d['end_to_end_with_witness_generation']['proofs_per_s'])" | tee -a $O/summary.txt
timeout -k 10 300 python tools/fuzz_shapes.py 300 5 11 > $O/fuzz.txt 2>&1; echo "fuzz rc=$?" | tee -a $O/summary.txt; tail -1 $O/fuzz.txt | tee -a $O/summary.txt
