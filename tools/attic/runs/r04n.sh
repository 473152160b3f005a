#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r04n; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
(cd $R && timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "attention" 2>&1 | tail -2)
for sh in 16x10x360x80 16x10x240x80; do
(cd $R && SHAPES=$sh VARIANTS=5,105 ITERS=20 timeout 300 rocprofv3 --kernel-trace --stats -d $O/ks -- python tools/attn_bench.py 2>&1 | grep "^B[0-9]" | cut -c1-140)
echo $sh; python $R/tools/prof_summary.py $O/ks | grep -i attn | cut -c29-200; rm -rf $O/ks
done
cd $R
ROUNDS=2 tools/runs/abk.sh "--opt attn_variant=105" "--opt attn_variant=5" | tee $O/ab.txt
