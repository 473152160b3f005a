#!/bin/bash
cd "$(dirname "$0")/../.."
R=$(pwd); export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -q -k "attention or ragged or supervised or headline" 2>&1 | tail -3
cd /tmp; VARIANTS=1 ITERS=20 timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/attnp -- python $R/tools/attn_bench.py > /dev/null 2>&1; cd $R
python tools/prof_summary.py gpurun_out/attnp /tmp/attn_stats.txt > /dev/null 2>&1; grep -E "attn_" /tmp/attn_stats.txt | cut -c1-70,100-160 | head -8; rm -rf gpurun_out/attnp
bash tools/runs/ab.sh ""
