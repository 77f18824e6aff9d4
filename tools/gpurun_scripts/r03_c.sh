#!/bin/bash
# Poseidon permutation microbenchmark (tools/poseidon_microbench.hip, built before the call): the round-3 permutation (partial
# rounds in the spectral domain) against the round-2 form, in registers and in leaf-hash-shaped kernels; then the GPU hashing tests.
set -o pipefail
O=gpurun_out/r03_c
mkdir -p $O
timeout -k 10 300 tools/scratch_bin/poseidon_microbench > $O/poseidon_microbench.txt 2>&1; rc=$?; echo "microbench rc=$rc" | tee -a $O/summary.txt
cat $O/poseidon_microbench.txt
[ $rc -eq 0 ] || exit $rc
python -m pytest tests/test_merkle_gpu.py tests/test_prove_gpu.py -m gpu -q -x > $O/pytest.txt 2>&1; rc=$?; echo "pytest rc=$rc" | tee -a $O/summary.txt; tail -3 $O/pytest.txt
[ $rc -eq 0 ] || exit $rc
python bench.py --no-tree --no-ntt --no-cpu-baseline --headline-only > $O/bench_headline.json 2> $O/bench.err; rc=$?; echo "bench rc=$rc" | tee -a $O/summary.txt
tail -c 1500 $O/bench_headline.json
