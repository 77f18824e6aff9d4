#!/bin/bash
O=gpurun_out/r02_af; mkdir -p $O
for sp in 1 2; do QPGPU_NTT_SPLIT=$sp python tools/lde_time.py 13 4320 2>&1 | tail -2 | tee -a $O/summary.txt; done
QPGPU_NTT_SPARSE=0 python tools/lde_time.py 13 4320 2>&1 | tail -2 | tee -a $O/summary.txt
python tools/lde_time.py 12 4320 2>&1 | tail -2 | tee -a $O/summary.txt
