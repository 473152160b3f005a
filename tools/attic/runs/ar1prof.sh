#!/bin/bash
# per-kernel durations of the batch-1 AR sampler (rocprofv3 kernel trace)
R=$(pwd); O=$R/gpurun_out/ar1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
(cd $R && timeout 300 rocprofv3 --kernel-trace --stats -d $O/ks -- python bench.py --mode ar --batch 1 --steps 24 --warmup 2 --profile-steps 0 > /dev/null 2>&1)
python $R/tools/prof_summary.py $O/ks $O/kernel_stats_b1.txt | head -30; rm -rf $O/ks
