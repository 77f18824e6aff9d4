#!/bin/bash
set -o pipefail
O=gpurun_out/r02_s
mkdir -p $O
timeout -k 10 200 python tools/pool_soak.py 60 3 16 > $O/pool_soak.txt 2>&1; echo "pool_soak rc=$?" | tee -a $O/summary.txt
tail -2 $O/pool_soak.txt | tee -a $O/summary.txt
timeout -k 10 200 python tools/pool_soak.py 40 2 32 >> $O/pool_soak.txt 2>&1; echo "pool_soak2 rc=$?" | tee -a $O/summary.txt
tail -1 $O/pool_soak.txt | tee -a $O/summary.txt
timeout -k 10 300 python tools/zk_soak.py 1000 > $O/zk_soak.txt 2>&1; echo "zk_soak rc=$?" | tee -a $O/summary.txt
tail -2 $O/zk_soak.txt | tee -a $O/summary.txt
