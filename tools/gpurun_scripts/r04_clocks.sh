#!/bin/bash
# Engine clock and power while the three loops the bench line reports on run (one gpurun call): the headline (six workers), the
# 2^20 x 128 NTT loop, the matrix-pipe leaf-hash loop. tools/clock_sampler.py reads sysfs beside a child process.
# Output: gpurun_out/r04_clocks/*.json (+ summary.txt); record: profiles/r04_clocks.txt.
set -o pipefail
O=gpurun_out/r04_clocks; mkdir -p $O
ls /sys/class/drm/ > $O/drm.txt 2>&1
for c in /sys/class/drm/card*/device; do echo "$c: $(cat $c/pp_dpm_sclk 2>/dev/null | tr '\n' ' ')"; done >> $O/drm.txt 2>&1
python tools/clock_sampler.py headline $O/headline.json -- python bench.py --steps 60 --warmup 5 --no-tree --no-ntt --no-cpu-baseline --headline-only > $O/headline.out 2> $O/headline.err || exit 1
python tools/clock_sampler.py ntt $O/ntt.json -- python tools/ntt_only.py 2000 > $O/ntt.out 2> $O/ntt.err || exit 2
python tools/clock_sampler.py leaf_hash $O/leaf.json -- python tools/leaf_time.py clocks 400 > $O/leaf.out 2> $O/leaf.err || exit 3
python tools/clock_sampler.py idle $O/idle.json -- sleep 3 > $O/idle.out 2>&1
tail -n 2 $O/headline.out $O/ntt.out $O/leaf.out | cut -c1-1500 | tee $O/summary.txt
