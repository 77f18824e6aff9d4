#!/bin/bash
set -o pipefail
O=gpurun_out/r02_d
mkdir -p $O
python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -4 $O/pytest.txt | tee -a $O/summary.txt
for cfg in "1 4" "1 8" "1 16" "2 4" "2 8" "2 16" "3 8" "4 4" "4 1"; do
  set -- $cfg
  python bench.py --steps 20 --warmup 2 --streams $1 --batch $2 --no-tree --no-ntt --no-cpu-baseline > $O/bench_$1_$2.json 2> $O/bench_$1_$2.err
  python - <<PY | tee -a $O/summary.txt
import json
try:
    d=json.load(open("$O/bench_$1_$2.json")); print("W=$1 B=$2", d["value"], "proofs/s", d["ms_per_proof"], "ms/proof", "single", d.get("single_proof_latency_ms"))
except Exception as e:
    print("W=$1 B=$2 failed", e)
PY
done
QPGPU_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 10 --warmup 2 --no-ntt --no-cpu-baseline --batch-degree-bits 12 > $O/bench_2rank_gloo.json 2> $O/bench_2rank_gloo.err; echo "2rank rc=$?" | tee -a $O/summary.txt
tail -c 600 $O/bench_2rank_gloo.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/$O/prof -o bench -- python3 $R/bench.py --steps 20 --warmup 2 --no-tree --no-ntt --no-cpu-baseline > $R/$O/prof.log 2>&1; echo "prof rc=$?" | tee -a $R/$O/summary.txt
