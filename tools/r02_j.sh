#!/bin/bash
set -o pipefail
O=gpurun_out/r02_j
mkdir -p $O
for cfg in "-1 -1" "4 -1" "-1 4" "4 4"; do
  set -- $cfg
  QPGPU_NTT_LOGT_S=$1 QPGPU_NTT_LOGT_R=$2 python tools/ntt_time.py "S$1_R$2" >> $O/ntt_variants.jsonl 2>>$O/err.txt
done
cat $O/ntt_variants.jsonl | tee -a $O/summary.txt
