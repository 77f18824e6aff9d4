#!/bin/bash
O=gpurun_out/r02_y; mkdir -p $O
timeout -k 10 200 python tools/clock_probe.py > $O/clock.txt 2>&1; echo "rc=$?" | tee -a $O/summary.txt; cat $O/clock.txt | tail -5 | tee -a $O/summary.txt
