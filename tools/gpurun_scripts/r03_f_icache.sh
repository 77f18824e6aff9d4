# instruction-cache counters of the 2^20 NTT pass kernels and of the leaf-hash kernels (is instruction fetch part of the bound?)
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/rb7; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU -d $O/ic_ntt -o ntt -- python3 $GRAFT_REPO_ROOT/tools/ntt_only.py 3 > $O/ic_ntt.log 2>&1 || exit 2
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU -d $O/ic_hash -o hash -- python3 $GRAFT_REPO_ROOT/tools/hash_probe.py > $O/ic_hash.log 2>&1 || exit 3
cd $GRAFT_REPO_ROOT
python tools/pmc_db_summary.py ntt_pass $(find $O/ic_ntt -name "*.db") > $O/ic_ntt.json
python tools/pmc_db_summary.py _kernel $(find $O/ic_hash -name "*.db") > $O/ic_hash.json
