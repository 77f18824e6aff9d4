#!/bin/bash
set -o pipefail
O=gpurun_out/r02_w
mkdir -p $O
python -m pytest tests/test_witness_gpu.py tests/test_batch_gpu.py tests/test_aggregation_gpu.py -m gpu -q -x > $O/pytest.txt 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -3 $O/pytest.txt | tee -a $O/summary.txt
python tools/tree_timing.py > $O/tree.txt 2>&1; echo "tree rc=$?" | tee -a $O/summary.txt
grep "levels\|commit" $O/tree.txt | tee -a $O/summary.txt
timeout -k 10 300 python tools/fuzz_shapes.py 300 9 11 > $O/fuzz.txt 2>&1; echo "fuzz rc=$?" | tee -a $O/summary.txt; tail -1 $O/fuzz.txt | tee -a $O/summary.txt
