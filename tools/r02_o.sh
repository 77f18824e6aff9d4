#!/bin/bash
set -o pipefail
O=gpurun_out/r02_o
mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?" | tee -a $O/summary.txt
tail -2 $O/smoke.txt | tee -a $O/summary.txt
python -m pytest tests/test_abi.py tests/test_witness_gpu.py -m gpu -q -x > $O/pytest.txt 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -3 $O/pytest.txt | tee -a $O/summary.txt
python tools/tree_timing.py > $O/tree.txt 2>&1; echo "tree rc=$?" | tee -a $O/summary.txt
cat $O/tree.txt | tee -a $O/summary.txt
