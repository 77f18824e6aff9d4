#!/bin/bash
# throughput-mode routing of lockstep batches (QPGPU_TPUT_BATCH): tests, then the default bench legs that show it
set -o pipefail
O=gpurun_out/tput; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1 || { tail -30 $O/pytest.txt; exit 1; }
tail -2 $O/pytest.txt
for i in 1 2; do
  for v in 1000000 8; do
    QPGPU_TPUT_BATCH=$v python bench.py --steps 40 --warmup 5 --no-ntt --no-cpu-baseline > $O/b${v}_$i.json 2> $O/b${v}_$i.err || { tail -5 $O/b${v}_$i.err; exit 2; }
    python - <<PY
import json
j=json.loads([l for l in open("$O/b${v}_$i.json") if l.startswith("{")][-1])
print("QPGPU_TPUT_BATCH=$v", $i, j["value"], j["window_proofs_per_s"], "d12", j["degree_bits_12"], "tree", j["aggregation_tree"]["seconds"], "p2", j["poseidon2_hasher"]["proofs_per_s"], "lat", j["single_proof_latency_ms"])
PY
  done
done
