#!/bin/bash
# The profiled headline through the torch-free C driver, once per variant (each variant a different experiment, no repeats):
#   A  the pool as it ships: workers take turns on the device when a queue-intercepting profiler is loaded
#   B  QPGPU_POOL_SERIALIZE=0 (concurrent submission, as unprofiled runs do) with the crash tracer armed: if the profiler's
#      interceptor faults again, the trace now names the mappings on either side of the faulting address
# B's death is the expected outcome and does not stop the script.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_prof_probe; mkdir -p $O
gcc -O2 -I include examples/leaf_prove_example.c -L qp-zk-circuits_amd -lqpgpu -lpthread -Wl,-rpath,$R/qp-zk-circuits_amd -o $O/leaf_driver || exit 1
cd /tmp && export TMPDIR=/tmp
export QPGPU_CRASH_TRACE=$O/crash_trace_A.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_A -o lp -- $O/leaf_driver 13 0 6 32 24 > $O/A.txt 2>&1; echo "A (serialized under the profiler) rc=$?" | tee -a $O/summary.txt
export QPGPU_CRASH_TRACE=$O/crash_trace_B.txt
QPGPU_POOL_SERIALIZE=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_B -o lp -- $O/leaf_driver 13 0 6 32 24 > $O/B.txt 2>&1; echo "B (concurrent submission under the profiler) rc=$?" | tee -a $O/summary.txt
find $O -name "*kernel_trace.csv" -delete
tail -3 $O/A.txt; tail -3 $O/B.txt; head -40 $O/crash_trace_B.txt 2>/dev/null
