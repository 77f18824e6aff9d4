#!/bin/bash
# final evidence of round 2: parity tests, the default bench line, kernel-trace stats of the timed region, and the PMC passes of
# the 2^20 x 128 NTT kernels (SQ counters in two passes; FETCH_SIZE and WRITE_SIZE in passes of their own, as
# MI355X_MICROARCH.md prescribes). Everything lands under gpurun_out/r02_final; tools/r02_collect.py turns it into profiles/.
set -o pipefail
O=gpurun_out/r02_final
mkdir -p $O
python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -3 $O/pytest.txt | tee -a $O/summary.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?" | tee -a $O/summary.txt
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/summary.txt
QPGPU_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 10 --warmup 2 --no-ntt --no-cpu-baseline --headline-only --batch-degree-bits 13 > $O/bench_2rank_gloo.json 2> $O/bench_2rank_gloo.err; echo "2rank rc=$?" | tee -a $O/summary.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_bench -o bench -- python3 $R/bench.py > $R/$O/prof_bench.log 2>&1; echo "prof rc=$?" | tee -a $R/$O/summary.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_ntt -o ntt -- python3 $R/tools/ntt_only.py 40 > $R/$O/prof_ntt.log 2>&1; echo "prof_ntt rc=$?" | tee -a $R/$O/summary.txt
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $R/$O/pmc_sq1 -o ntt -- python3 $R/tools/ntt_only.py 3 > $R/$O/pmc_sq1.log 2>&1; echo "pmc_sq1 rc=$?" | tee -a $R/$O/summary.txt
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE -d $R/$O/pmc_sq2 -o ntt -- python3 $R/tools/ntt_only.py 3 > $R/$O/pmc_sq2.log 2>&1; echo "pmc_sq2 rc=$?" | tee -a $R/$O/summary.txt
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$O/pmc_fetch -o ntt -- python3 $R/tools/ntt_only.py 3 > $R/$O/pmc_fetch.log 2>&1; echo "pmc_fetch rc=$?" | tee -a $R/$O/summary.txt
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$O/pmc_write -o ntt -- python3 $R/tools/ntt_only.py 3 > $R/$O/pmc_write.log 2>&1; echo "pmc_write rc=$?" | tee -a $R/$O/summary.txt
bash $R/tools/r02_z.sh
# one worker, one lockstep batch at a time: every kernel alone on the GPU, i.e. the isolated cost of each stage of a batch of 32
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_single_worker -o sw -- python3 $R/bench.py --streams 1 --batch 32 --steps 8 --warmup 2 --no-tree --no-ntt --no-cpu-baseline --headline-only > $R/$O/prof_single_worker.log 2>&1; echo "prof_single_worker rc=$?" | tee -a $R/$O/summary.txt
# summarise the kernel traces here and drop them: gpurun brings back at most 64 MiB
cd $R
python tools/profile_summary.py $O/prof_bench $O/sum_bench "python3 bench.py (default command)" >> $O/summary.txt 2>&1
python tools/profile_summary.py $O/prof_ntt $O/sum_ntt_only "python3 tools/ntt_only.py 40" >> $O/summary.txt 2>&1
python tools/profile_summary.py $O/prof_single_worker $O/sum_single_worker "python3 bench.py --streams 1 --batch 32 --steps 8 --warmup 2 --no-tree --no-ntt --no-cpu-baseline --headline-only" >> $O/summary.txt 2>&1
python tools/profile_summary.py $O/prof_headline $O/sum_headline "python3 bench.py --steps 120 --warmup 5 --no-tree --no-ntt --no-cpu-baseline --headline-only" >> $O/summary.txt 2>&1
find $O -name "*kernel_trace.csv" -delete
du -sh $O | tee -a $O/summary.txt
