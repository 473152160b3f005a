#!/bin/bash
# call E: optimizer release point in the tail (adam_hold = k), adjusted adam_in_wgrad test
cd "$(dirname "$0")/../../.."; R=$(pwd); O=$R/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_model.py -q -x -k "adam_in_wgrad" -s 2>&1 | grep -E "adam_in_wgrad vs|passed|failed|Error" | cut -c1-400 | tee $O/r5_e_tests.txt
run() { local label=$1; shift
  ms=$(timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --profile-steps 1 "$@" 2>/dev/null | tail -1 | python -c 'import sys,json; print(json.loads(sys.stdin.read())["ms_per_step"])')
  echo "$label : $ms ms" | tee -a $O/r5_ab_adam_hold.txt; }
rm -f $O/r5_ab_adam_hold.txt
for round in 1 2 3; do
  run "adam_hold=1 (release at cross layer 0)" --opt adam_hold=1
  run "adam_hold=2 (layer 1)" --opt adam_hold=2
  run "adam_hold=3 (layer 2)" --opt adam_hold=3
  run "adam_hold=5 (layer 4)" --opt adam_hold=5
done
