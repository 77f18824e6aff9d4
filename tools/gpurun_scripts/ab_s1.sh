#!/bin/bash
set -o pipefail
L=qp-zk-circuits_amd/libqpgpu.so
cp $L /tmp/b.so; cp ab_old.so /tmp/a.so
timeout -k 10 500 python -m pytest tests/test_poseidon2_gate_gpu.py tests/test_leaf_circuit_gpu.py tests/test_witness_gpu.py -x -q -m gpu > gpurun_out/dpp_tests.log 2>&1 || { tail -20 gpurun_out/dpp_tests.log; exit 1; }
tail -2 gpurun_out/dpp_tests.log
for i in 1 2 3; do for v in a b; do cp /tmp/$v.so $L; echo "== $v $i"; python tools/witness_fuse_ab.py $v$i | tail -1; python3 tools/single_proof_timeline.py; done; done
cp /tmp/b.so $L
