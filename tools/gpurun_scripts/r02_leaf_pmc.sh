# SQ / GRBM counters of the leaf-hash kernel alone (tools/hash_probe.py: a 2^21-leaf tree over 135 columns), two passes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/leaf_pmc
mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD --output-format csv -d $O/p1 -o leaf -- python3 $R/tools/hash_probe.py > $O/p1.log 2>&1; echo "p1 rc=$?"
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/p2 -o leaf -- python3 $R/tools/hash_probe.py > $O/p2.log 2>&1; echo "p2 rc=$?"
find $O -name "*kernel_trace.csv" -delete
du -sh $O
