#!/bin/bash
# first GPU call of round 2: parity tests, NTT variants, PMC counters of the NTT kernels, a short bench
set -o pipefail
O=gpurun_out/r02_a
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -3 $O/pytest.txt | tee -a $O/summary.txt
python tools/ntt_time.py base >> $O/ntt_variants.jsonl 2>$O/ntt_err.txt
QPGPU_NTT_LOGT=4 python tools/ntt_time.py logt4 >> $O/ntt_variants.jsonl 2>>$O/ntt_err.txt
QPGPU_NTT_TW=1 python tools/ntt_time.py tw_chain >> $O/ntt_variants.jsonl 2>>$O/ntt_err.txt
QPGPU_NTT_TW=2 python tools/ntt_time.py tw_skipped >> $O/ntt_variants.jsonl 2>>$O/ntt_err.txt
QPGPU_NTT_TW=1 QPGPU_NTT_LOGT=4 python tools/ntt_time.py tw_chain_logt4 >> $O/ntt_variants.jsonl 2>>$O/ntt_err.txt
cat $O/ntt_variants.jsonl | tee -a $O/summary.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY -d $R/$O/pmc_sq -o ntt -- python3 $R/tools/ntt_only.py 3 > $R/$O/pmc_sq.log 2>&1; echo "pmc_sq rc=$?" | tee -a $R/$O/summary.txt
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE -d $R/$O/pmc_sq2 -o ntt -- python3 $R/tools/ntt_only.py 3 > $R/$O/pmc_sq2.log 2>&1; echo "pmc_sq2 rc=$?" | tee -a $R/$O/summary.txt
cd $R
python bench.py --steps 30 --warmup 3 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/summary.txt
tail -c 1500 $O/bench.json
