# one proving worker, one lockstep batch of 32 at a time: kernels run alone on the GPU, so rocprofv3's durations are isolated costs
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/sw_prof -o sw -- python3 $R/bench.py --streams 1 --batch 32 --steps 6 --warmup 2 --no-tree --no-ntt --no-cpu-baseline --headline-only > $R/gpurun_out/sw_prof.log 2>&1
find $R/gpurun_out/sw_prof -name "*kernel_trace.csv" -delete
