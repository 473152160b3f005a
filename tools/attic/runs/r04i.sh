#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r04i; mkdir -p $O
python tools/kprof_timeline.py $O/timeline.csv > $O/timeline.txt 2>/dev/null; head -3 $O/timeline.txt; wc -l $O/timeline.txt
