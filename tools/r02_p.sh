#!/bin/bash
set -o pipefail
O=gpurun_out/r02_p
mkdir -p $O
python -m pytest tests/test_prove_gpu.py -m gpu -q -x -k "salts or zero_knowledge" > $O/pytest.txt 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -3 $O/pytest.txt | tee -a $O/summary.txt
timeout -k 10 600 python tools/fuzz_shapes.py 400 21 11 > $O/fuzz.txt 2>&1; echo "fuzz rc=$?" | tee -a $O/summary.txt
tail -4 $O/fuzz.txt | tee -a $O/summary.txt
