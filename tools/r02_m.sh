#!/bin/bash
set -o pipefail
O=gpurun_out/r02_m
mkdir -p $O
python -m pytest tests/test_aggregation_gpu.py tests/test_multirank_gpu.py tests/test_batch_gpu.py -m gpu -q -x > $O/pytest.txt 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -25 $O/pytest.txt | tee -a $O/summary.txt
python bench.py --steps 6 --warmup 2 --no-ntt --no-cpu-baseline --headline-only > $O/bench_tree.json 2> $O/bench_tree.err; echo "bench rc=$?" | tee -a $O/summary.txt
tail -5 $O/bench_tree.err | tee -a $O/summary.txt
python -c "
import json
j=json.loads([l for l in open('$O/bench_tree.json') if l.startswith('{')][-1]); print(j['value'], j['aggregation_tree']['seconds'], j['aggregation_tree'].get('levels_rank0'), j['aggregation_tree'].get('checked'))" | tee -a $O/summary.txt
