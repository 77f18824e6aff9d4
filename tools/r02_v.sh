#!/bin/bash
set -o pipefail
O=gpurun_out/r02_v
mkdir -p $O
python -m pytest tests -m gpu -q -x > $O/pytest.txt 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -3 $O/pytest.txt | tee -a $O/summary.txt
python tools/tree_timing.py > $O/tree.txt 2>&1; echo "tree rc=$?" | tee -a $O/summary.txt
grep "levels\|commit" $O/tree.txt | tee -a $O/summary.txt
python bench.py --steps 20 --warmup 3 --no-tree --no-ntt --no-cpu-baseline > $O/b.json 2> $O/b.err
python -c "
import json
d=json.loads([l for l in open('$O/b.json') if l.startswith('{')][-1]); print('bench', d['value'], d['single_proof_latency_ms'], d['witness_generation']['single_ms'], d['witness_generation']['batched_ms_per_witness'], d['end_to_end_with_witness_generation']['proofs_per_s'])" | tee -a $O/summary.txt
