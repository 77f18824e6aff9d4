#!/bin/bash
set -o pipefail
O=gpurun_out/r02_n
mkdir -p $O
python -m pytest tests/test_verifier_gpu.py tests/test_aggregation_gpu.py tests/test_multirank_gpu.py -m gpu -q -x > $O/pytest.txt 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -25 $O/pytest.txt | tee -a $O/summary.txt
python bench.py --steps 6 --warmup 2 --no-ntt --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/summary.txt
tail -5 $O/bench.err | tee -a $O/summary.txt
python -c "
import json
j=json.loads([l for l in open('$O/bench.json') if l.startswith('{')][-1]); print(j['value'], j['aggregation_tree']['seconds'], j['aggregation_tree'].get('levels_rank0'), j.get('host_verifier'))" | tee -a $O/summary.txt
