#!/bin/bash
# workers x lockstep sweep of the headline leg on the final kernels (one call)
set -o pipefail
O=gpurun_out/ws3; mkdir -p $O
for i in 1 2; do
  for cfg in "6 32" "4 48" "8 24" "6 48" "3 64" "8 32" "12 16"; do
    set -- $cfg
    python bench.py --streams $1 --batch $2 --steps 40 --warmup 5 --no-tree --no-ntt --no-cpu-baseline --headline-only > $O/w$1_b$2_$i.json 2> $O/w$1_b$2_$i.err || { tail -3 $O/w$1_b$2_$i.err; continue; }
    python - <<PY
import json
j=json.loads([l for l in open("$O/w$1_b$2_$i.json") if l.startswith("{")][-1])
print("workers $1 lockstep $2 run $i:", j["value"], j["window_proofs_per_s"], "proofs/step", j["config"]["proofs_per_step_per_gpu"])
PY
  done
done
