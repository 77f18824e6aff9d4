#!/bin/bash
# matrix-pipe build of the hashing kernels: parity tests, then the headline with it off / on / off / on inside one call
set -o pipefail
O=gpurun_out/mx; mkdir -p $O
python -m pytest tests/test_merkle_gpu.py tests/test_batch_gpu.py tests/test_prove_gpu.py -m gpu -x -q > $O/pytest_merkle.txt 2>&1 || { tail -30 $O/pytest_merkle.txt; exit 1; }
tail -3 $O/pytest_merkle.txt
for i in 1 2; do
  for v in 0 1; do
    QPGPU_MX=$v python bench.py --steps 40 --warmup 5 --no-tree --no-ntt --no-cpu-baseline --headline-only > $O/mx${v}_$i.json 2> $O/mx${v}_$i.err || { tail -5 $O/mx${v}_$i.err; exit 2; }
    python - <<PY
import json
j=json.loads([l for l in open("$O/mx${v}_$i.json") if l.startswith("{")][-1])
print("QPGPU_MX=$v", $i, j["value"], j["window_proofs_per_s"])
PY
  done
done
