O=gpurun_out/ws_sweep2.txt
: > $O
for cfg in "6 32" "6 64" "8 32" "4 64" "12 16"; do
  set -- $cfg
  v=$(python bench.py --streams $1 --batch $2 --steps 40 --warmup 5 --no-tree --no-ntt --no-cpu-baseline --headline-only 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['window_proofs_per_s'])")
  echo "workers=$1 lockstep=$2 -> $v" | tee -a $O
done
