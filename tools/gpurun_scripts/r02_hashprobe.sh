set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/hp_lib -o lib -- python3 $R/tools/hash_probe.py > $R/gpurun_out/hp_lib.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/hp_mb -o mb -- $R/tools/scratch_bin/poseidon_microbench > $R/gpurun_out/hp_mb.log 2>&1
cd $R/gpurun_out
find hp_lib hp_mb -name "*kernel_stats.csv" | while read f; do echo "== $f"; cut -d, -f1-4,6,7 "$f" | cut -c1-200; done
find hp_lib hp_mb -name "*kernel_trace.csv" -delete
