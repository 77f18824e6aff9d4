#!/bin/bash
set -o pipefail
O=gpurun_out/r02_e
mkdir -p $O
python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -4 $O/pytest.txt | tee -a $O/summary.txt
run() { # label, env..., args
  label=$1; shift
  env "$@" python bench.py --steps 20 --warmup 2 --no-tree --no-ntt --no-cpu-baseline --headline-only $ARGS > $O/b_$label.json 2> $O/b_$label.err
  python - <<PY | tee -a $O/summary.txt
import json
try:
    d=json.load(open("$O/b_$label.json")); print("$label", d["value"], "proofs/s", d["window_proofs_per_s"])
except Exception as e:
    print("$label failed", e)
PY
}
for cm in 1024 4096 16384 65536; do
  ARGS="--streams 2 --batch 16" run coop${cm}_2x16 QPGPU_COOP_MAX=$cm
  ARGS="--streams 3 --batch 8" run coop${cm}_3x8 QPGPU_COOP_MAX=$cm
done
ARGS="--streams 2 --batch 16" run top_2x16 QPGPU_TREE_TOP=1
ARGS="--streams 3 --batch 8" run top_3x8 QPGPU_TREE_TOP=1
ARGS="--streams 3 --batch 16" run w3x16 A=1
ARGS="--streams 2 --batch 32" run w2x32 A=1
ARGS="--streams 4 --batch 8" run w4x8 A=1
rocprofv3 -L > $O/counters.txt 2>&1
grep -i -E "ICACHE|IFETCH|SQC_" $O/counters.txt | head -40 > $O/icache_counters.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/$O/prof -o bench -- python3 $R/bench.py --steps 30 --warmup 2 --no-tree --no-ntt --no-cpu-baseline --headline-only --streams 2 --batch 16 > $R/$O/prof.log 2>&1; echo "prof rc=$?" | tee -a $R/$O/summary.txt
cd $R
python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default bench rc=$?" | tee -a $O/summary.txt
