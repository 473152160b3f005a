#!/bin/bash
cd "$(dirname "$0")/../../.."; R=$(pwd); O=$R/gpurun_out; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_ops.py -q -x 2>&1 | tail -3 | tee $O/r5_h_tests.txt
timeout 1200 python -m pytest tests/test_gpu_model.py -q -x -k "fact_v5 or headline or tiny_forward or supervised or big_tile" 2>&1 | tail -3 | tee -a $O/r5_h_tests.txt
run() { local label=$1; shift
  ms=$(timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --profile-steps 1 "$@" 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])')
  echo "$label : $ms ms" | tee -a $O/r5_ab_t128b.txt; }
rm -f $O/r5_ab_t128b.txt
for round in 1 2 3 4 5; do run "tile128x160=0" --opt tile128x160=0; run "tile128x160=1 (default)"; done
