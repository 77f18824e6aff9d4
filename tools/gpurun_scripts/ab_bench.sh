#!/bin/bash
# A/B of two builds of the library inside ONE gpurun call (boxes of the pool differ by several per cent, runs on different boxes
# do not compare): usage ab_bench.sh <b.so> [rounds]; A = qp-zk-circuits_amd/libqpgpu.so. Alternates A, B, A, B.
set -o pipefail
B=$1; N=${2:-2}
L=qp-zk-circuits_amd/libqpgpu.so
O=gpurun_out/ab; mkdir -p $O
cp $L /tmp/a.so; cp $B /tmp/b.so
for i in $(seq 1 $N); do
  for v in a b; do
    cp /tmp/$v.so $L
    python bench.py --steps 40 --warmup 5 --no-tree --no-ntt --no-cpu-baseline --headline-only > $O/${v}_$i.json 2> $O/${v}_$i.err || { cp /tmp/a.so $L; exit 1; }
    python tools/ntt_time.py ${v}_$i 10 >> $O/ntt.txt 2>&1 || { cp /tmp/a.so $L; exit 2; }
    python - <<PY
import json
j=json.loads([l for l in open("$O/${v}_$i.json") if l.startswith("{")][-1])
print("$v", $i, j["value"], j["window_proofs_per_s"])
PY
  done
done
cp /tmp/a.so $L
cat $O/ntt.txt
