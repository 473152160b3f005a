#!/bin/bash
# PMC passes over the round-2 GEMM kernels (tools/bench_r2.py pmc): run from the repo root on the GPU box.
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; : > $O/pmc_r2.txt
cd /tmp && export TMPDIR=/tmp
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum" "FETCH_SIZE"; do
  d=$O/pmc_tmp
  (cd $R && timeout 150 rocprofv3 --kernel-trace --pmc $set -d $d -- python tools/bench_r2.py pmc > /dev/null 2>&1)
  echo "== $set" >> $O/pmc_r2.txt
  python $R/tools/pmc_summary.py $d big_ >> $O/pmc_r2.txt 2>&1
  python $R/tools/pmc_summary.py $d gemm_nt_fast >> $O/pmc_r2.txt 2>&1
  rm -rf $d
done
