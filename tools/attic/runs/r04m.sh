#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r04m; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for sh in 16x10x360x80 16x10x240x80 16x10x120x80; do
(cd $R && SHAPES=$sh VARIANTS=5 ITERS=20 timeout 300 rocprofv3 --kernel-trace --stats -d $O/ks -- python tools/attn_bench.py > /dev/null 2>&1)
echo $sh; python $R/tools/prof_summary.py $O/ks | grep -i attn | cut -c29-200; rm -rf $O/ks
done
