#!/bin/bash
# hardware counters of the matrix-pipe leaf-hash kernel (and of the throughput build beside it): rocprofv3 --pmc passes over
# tools/hash_probe.py (a 2^21-leaf x 135-column tree = one lockstep batch's wires commitment); -> tools/collect_mx_counters.py
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/mx_pmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/counters_list.txt 2>&1 || true
grep -i -o "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*" $O/counters_list.txt | sort -u > $O/mfma_counters.txt || true
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU -d $O/p1 -o hash -- python3 $R/tools/hash_probe.py > $O/p1.log 2>&1 || { tail -5 $O/p1.log; exit 2; }
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS -d $O/p2 -o hash -- python3 $R/tools/hash_probe.py > $O/p2.log 2>&1 || { tail -5 $O/p2.log; exit 3; }
QPGPU_MX=0 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d $O/p3 -o hash -- python3 $R/tools/hash_probe.py > $O/p3.log 2>&1 || { tail -5 $O/p3.log; exit 4; }
cd $R
python tools/pmc_db_summary.py _kernel $(find $O/p1 $O/p2 -name "*.db") > $O/mx.json
python tools/pmc_db_summary.py _kernel $(find $O/p3 -name "*.db") > $O/tp.json
python tools/kernel_id.py hash_mx > $O/kernel_source_id.txt
cat $O/mfma_counters.txt; python -c "
import json; d=json.load(open('$O/mx.json'))
for k,v in d.items():
    if 'leaf_hash' in k or 'node_kernel' in k: print(k, {a:b for a,b in v.items()})
"
