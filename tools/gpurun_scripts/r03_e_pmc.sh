set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/rb4; mkdir -p $O
timeout -k 10 60 $GRAFT_REPO_ROOT/tools/scratch_bin/ntt_limb_grp > $O/block.txt 2>&1 || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $O/pmc_tp -o ntt -- python3 $GRAFT_REPO_ROOT/tools/ntt_only.py 3 > $O/pmc_tp.log 2>&1 || exit 2
QPGPU_TP_MIN_THREADS=99999999999 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $O/pmc_lat -o ntt -- python3 $GRAFT_REPO_ROOT/tools/ntt_only.py 3 > $O/pmc_lat.log 2>&1 || exit 3
cd $GRAFT_REPO_ROOT
python tools/pmc_db_summary.py ntt_pass $(find $O/pmc_tp -name "*.db") > $O/tp.json
python tools/pmc_db_summary.py ntt_pass $(find $O/pmc_lat -name "*.db") > $O/lat.json
cat $O/block.txt
