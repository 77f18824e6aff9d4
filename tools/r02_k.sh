#!/bin/bash
set -o pipefail
O=gpurun_out/r02_k
mkdir -p $O
python -m pytest tests -m gpu -q -x > $O/pytest.txt 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -2 $O/pytest.txt | tee -a $O/summary.txt
python tools/ntt_time.py "default" >> $O/ntt_variants.jsonl 2>>$O/err.txt
cat $O/ntt_variants.jsonl | tee -a $O/summary.txt
python tools/big_proof.py 19 --routed 60 --zk > $O/big.txt 2>&1; tail -3 $O/big.txt | tee -a $O/summary.txt
