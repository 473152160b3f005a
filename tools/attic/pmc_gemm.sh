#!/bin/bash
# usage: tools/pmc_gemm.sh <tag> <shape idx> [env assignments...]   (run on the GPU box from the repo root)
tag=$1; sh=$2; shift 2
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
for set in "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum" "TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum"; do
  d=$R/gpurun_out/pmc3/${tag}_$(echo $set | cut -c1-10 | tr " " _)
  (cd $R && env "$@" timeout 120 rocprofv3 --kernel-trace --pmc $set -d $d -- python tools/gemm_bench.py $sh 1 5 > /dev/null 2>&1)
  echo "== $tag $*"; python $R/tools/pmc_summary.py $d gemm_nt 2>&1 | grep -v "^void"
  rm -rf $d
done
