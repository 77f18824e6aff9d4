#!/bin/bash
set -o pipefail
O=gpurun_out/r02_b
mkdir -p $O
for cfg in "3 3" "3 2" "3 1" "2 2" "2 3" "4 2"; do
  set -- $cfg
  QPGPU_NTT_TW=1 QPGPU_NTT_LOGT_S=$1 QPGPU_NTT_LOGT_R=$2 python tools/ntt_time.py "S$1_R$2" >> $O/ntt_variants.jsonl 2>>$O/err.txt
done
cat $O/ntt_variants.jsonl
python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?"
tail -5 $O/pytest.txt
