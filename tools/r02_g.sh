#!/bin/bash
set -o pipefail
O=gpurun_out/r02_g
mkdir -p $O
python -m pytest tests/test_multirank_gpu.py tests/test_batch_gpu.py tests/test_hasher_plug.py -m gpu -q -x > $O/pytest.txt 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -3 $O/pytest.txt | tee -a $O/summary.txt
python bench.py --steps 10 --warmup 3 --no-ntt --no-cpu-baseline --headline-only > $O/bench_tree.json 2> $O/bench_tree.err; echo "bench rc=$?" | tee -a $O/summary.txt
python -c "
import json
j=json.loads([l for l in open('$O/bench_tree.json') if l.startswith('{')][-1]); print(j['value'], j['aggregation_tree']['seconds'], j['aggregation_tree'].get('levels_rank0'), j['aggregation_tree'].get('checked'))" | tee -a $O/summary.txt
QPGPU_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 30 --warmup 5 --no-tree --no-ntt --no-cpu-baseline --headline-only > $O/bench_2rank.json 2> $O/bench_2rank.err; echo "2rank rc=$?" | tee -a $O/summary.txt
python -c "
import json
j=json.loads([l for l in open('$O/bench_2rank.json') if l.startswith('{')][-1]); print(j['value'], j['window_proofs_per_s'], j['step_ms_rank0'])" | tee -a $O/summary.txt
