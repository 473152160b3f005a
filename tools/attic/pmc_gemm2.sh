#!/bin/bash
# usage: tools/pmc_gemm2.sh <tag> <shape idx> <variant>   (GPU box, repo root)
tag=$1; sh=$2; var=$3
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_INSTS_LDS" "SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  d=$R/gpurun_out/pmc3/${tag}_$(echo $set | cut -c1-10 | tr " " _)
  (cd $R && timeout 120 rocprofv3 --kernel-trace --pmc $set -d $d -- python tools/gemm_bench.py $sh $var 5 > /dev/null 2>&1)
  python $R/tools/pmc_summary.py $d gemm_nt 2>&1 | grep -v "^void"
  rm -rf $d
done
