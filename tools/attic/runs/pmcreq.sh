#!/bin/bash
cd "$(dirname "$0")/../.."
R=$(pwd); O=$R/gpurun_out/pmcreq; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
d=$O/tmp
(cd $R && timeout 200 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $d -- python tools/bench_r2.py pmcreq > /dev/null 2>&1)
python $R/tools/pmc_summary.py $d big_nt | grep -v "^$" 
python $R/tools/pmc_summary.py $d gemm_nt_fast | grep -v "^$"
rm -rf $d
