#!/bin/bash
# threshold of the large-launch builds of the hashing kernels (QPGPU_TP_MIN_THREADS), headline leg, alternated inside one call
set -o pipefail
O=gpurun_out/thr; mkdir -p $O
for i in 1 2; do
  for v in 262144 131072 65536; do
    QPGPU_TP_MIN_THREADS=$v python bench.py --steps 40 --warmup 5 --no-tree --no-ntt --no-cpu-baseline --headline-only > $O/t${v}_$i.json 2> $O/t${v}_$i.err || { tail -5 $O/t${v}_$i.err; exit 2; }
    python - <<PY
import json
j=json.loads([l for l in open("$O/t${v}_$i.json") if l.startswith("{")][-1])
print("QPGPU_TP_MIN_THREADS=$v", $i, j["value"], j["window_proofs_per_s"])
PY
  done
done
