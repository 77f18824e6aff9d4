# headline rate against the lane-cooperative threshold and the fused tree top (6 workers x 32)
O=gpurun_out/coop_sweep.txt
: > $O
for cfg in "16384 1" "4096 1" "1024 1" "0 1" "16384 0" "1024 0"; do
  set -- $cfg
  v=$(QPGPU_COOP_MAX=$1 QPGPU_TREE_TOP=$2 python bench.py --steps 40 --warmup 5 --no-tree --no-ntt --no-cpu-baseline --headline-only 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['window_proofs_per_s'])")
  echo "COOP_MAX=$1 TREE_TOP=$2 -> $v" | tee -a $O
done
