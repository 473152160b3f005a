#!/bin/bash
# Collect the judged evidence of a round on the GPU box (run from the repo root):
#   gpurun_out/rp/bench.json, kernel_stats.txt, pmc.txt
R=$(pwd); O=$R/gpurun_out/rp; mkdir -p $O
python bench.py --steps 30 --warmup 5 2>/dev/null | tail -1 > $O/bench.json
cd /tmp && export TMPDIR=/tmp
(cd $R && timeout 400 rocprofv3 --kernel-trace --stats -d $O/ks -- python bench.py --steps 30 --warmup 5 --no-cpu-baseline > /dev/null 2>&1)
python $R/tools/prof_summary.py $O/ks $O/kernel_stats.txt > /dev/null; rm -rf $O/ks
: > $O/pmc.txt
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum"; do
  d=$O/pmc_$(echo $set | cut -c1-10 | tr " " _)
  (cd $R && PITCH=832 timeout 120 rocprofv3 --kernel-trace --pmc $set -d $d -- python tools/gemm_bench.py 0 0 5 > /dev/null 2>&1)
  python $R/tools/pmc_summary.py $d gemm_nt >> $O/pmc.txt 2>&1; rm -rf $d
done
(cd $R && PITCH=832 python tools/gemm_bench.py 0 0 30 2>/dev/null | tail -1 >> $O/pmc.txt)
