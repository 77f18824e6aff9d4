set -o pipefail
O=gpurun_out/r04_sweep2; mkdir -p $O
for cfg in "6 32" "3 32" "4 32" "3 64" "4 48" "2 96" "6 32"; do
  set -- $cfg
  python bench.py --steps 30 --warmup 4 --streams $1 --batch $2 --no-tree --no-ntt --no-cpu-baseline --headline-only > $O/w$1_b$2.json 2> $O/w$1_b$2.err || exit 1
  python - <<PY
import json
j=json.loads([l for l in open("$O/w$1_b$2.json") if l.startswith("{")][-1])
print("workers $1 lockstep $2:", j["value"], j["window_proofs_per_s"], j["engine_clock_mhz"]["mean"], j["engine_clock_mhz"].get("board_power_w_mean"))
PY
done
