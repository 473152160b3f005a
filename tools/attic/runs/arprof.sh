#!/bin/bash
# rocprofv3 kernel stats of the AR sampler at 32 sequences and at batch 1 (final state of the round)
R=$(pwd); O=$R/gpurun_out/arprof; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
(cd $R && timeout 200 rocprofv3 --kernel-trace --stats -d $O/k32 -- python bench.py --mode ar --steps 24 --warmup 2 --profile-steps 0 > /dev/null 2>&1)
python $R/tools/prof_summary.py $O/k32 $O/kernel_stats_ar_b32.txt | head -12; rm -rf $O/k32
(cd $R && timeout 200 rocprofv3 --kernel-trace --stats -d $O/k1 -- python bench.py --mode ar --batch 1 --steps 24 --warmup 2 --profile-steps 0 > /dev/null 2>&1)
python $R/tools/prof_summary.py $O/k1 $O/kernel_stats_ar_b1.txt | head -12; rm -rf $O/k1
