#!/bin/bash
set -o pipefail
O=gpurun_out/r02_f
mkdir -p $O
python -m pytest tests -m gpu -q -x > $O/pytest.txt 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -4 $O/pytest.txt | tee -a $O/summary.txt
for cfg in "1 1"; do
  set -- $cfg
  QPGPU_NTT_TPB_S=$1 QPGPU_NTT_TPB_R=$2 python tools/ntt_time.py "tpbS$1_R$2" >> $O/ntt_variants.jsonl 2>>$O/err.txt
done
cat $O/ntt_variants.jsonl | tee -a $O/summary.txt
run() { label=$1; shift
  env "$@" python bench.py --steps 20 --warmup 3 --no-tree --no-ntt --no-cpu-baseline --headline-only $ARGS > $O/b_$label.json 2> $O/b_$label.err
  python - <<PY | tee -a $O/summary.txt
import json
try:
    d=json.load(open("$O/b_$label.json")); print("$label", d["value"], "proofs/s", d["window_proofs_per_s"])
except Exception as e:
    print("$label failed", e)
PY
}
for tt in 0 32 256 512; do
  ARGS="--streams 2 --batch 16" run top${tt}_2x16 QPGPU_TREE_TOP=$tt
  ARGS="--streams 2 --batch 32" run top${tt}_2x32 QPGPU_TREE_TOP=$tt
done
ARGS="--streams 3 --batch 32" run w3x32 A=1
ARGS="--streams 2 --batch 64" run w2x64 A=1
