#!/bin/bash
cd "$(dirname "$0")/../.."
R=$(pwd); O=$R/gpurun_out/pmcloop; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
d=$O/tmp
(cd $R && TN_LOOPS=0,2 timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS -d $d -- python tools/bench_r2.py pmcloop > /dev/null 2>&1)
python $R/tools/pmc_ratios.py $d | tee $O/ratios.txt
rm -rf $d
