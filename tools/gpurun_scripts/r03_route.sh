#!/bin/bash
# routing of the small hashing launches under six saturating workers: default (lane-cooperative kernels up to 16 384 hashes, fused
# tree top, latency build below 2^18) against thread-per-hash on the matrix build for everything; headline leg, alternated
set -o pipefail
O=gpurun_out/route; mkdir -p $O
run() { # label, env...
  local label=$1; shift
  env "$@" python bench.py --steps 40 --warmup 5 --no-tree --no-ntt --no-cpu-baseline --headline-only > $O/$label.json 2> $O/$label.err || { tail -5 $O/$label.err; exit 2; }
  python - <<PY
import json
j=json.loads([l for l in open("$O/$label.json") if l.startswith("{")][-1])
print("$label", j["value"], j["window_proofs_per_s"])
PY
}
for i in 1 2; do
  run default_$i X=1
  run all_mx_$i QPGPU_TP_MIN_THREADS=0 QPGPU_COOP_MAX=0 QPGPU_TREE_TOP=0
  run nocoop_$i QPGPU_COOP_MAX=0 QPGPU_TREE_TOP=0
  run mx_nocoop_keep_top_$i QPGPU_TP_MIN_THREADS=0 QPGPU_COOP_MAX=0
done
