#!/bin/bash
# per-kernel times of tools/attn_bench.py (rocprofv3 --kernel-trace --stats), one profiler run per (shape, variant);
# run from the repo root on the GPU box:  SHAPES=16x10x360x80,... VARIANTS=1,0 tools/attn_prof.sh
R=$(pwd); O=$R/gpurun_out/attn; mkdir -p $O; : > $O/kernel_stats.txt
cd /tmp && export TMPDIR=/tmp
for sh in $(echo ${SHAPES:-16x10x360x80} | tr , " "); do
 for v in $(echo ${VARIANTS:-1,2} | tr , " "); do
  (cd $R && SHAPES=$sh VARIANTS=$v timeout 300 rocprofv3 --kernel-trace --stats -d $O/ks -- python tools/attn_bench.py 2>&1 | grep "^B[0-9]" | cut -c1-150)
  python $R/tools/prof_summary.py $O/ks | grep -i "attn" | sed -e 's/void (anonymous namespace):://' | awk -v s=$sh -v v=$v '{printf "  %-12s v%-6s %-34s calls %4s avg %8s us\n", s, v, substr($1,1,34), $(NF-3), $(NF-1)}' | tee -a $O/kernel_stats.txt
  rm -rf $O/ks
 done
done
