#!/bin/bash
# Round 3, GPU call B: kernel timeline of the current step (rocprofv3 kernel trace) + layer view
cd "$(dirname "$0")/../.."
R=$(pwd); export TMPDIR=/tmp
O=$R/gpurun_out/r03b; mkdir -p $O
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -- python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline > $O/bench_prof.json 2> $O/bench_prof.err
cd $R
python tools/layer_view.py $O/kt > $O/layer_view.txt 2>&1
DB=$(find $O/kt -name "*.db" | head -1)
python tools/trace_gaps.py $DB > $O/gaps.txt 2>&1
python tools/prof_summary.py $O/kt $O/kernel_stats.txt > /dev/null 2>&1
rm -rf $O/kt
cat $O/gaps.txt; cat $O/layer_view.txt
