#!/bin/bash
cd "$(dirname "$0")/../../.."; O=gpurun_out; mkdir -p $O
run() { local label=$1; shift
  ms=$(timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --profile-steps 1 "$@" 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); k={r["name"]:r for r in d["kernels"]}; print(d["ms_per_step"], "ms  ffn1_dgrad", k["ffn1_dgrad"]["avg_launch_us"], " qkv_dgrad", k["qkv_dgrad"]["avg_launch_us"], " loss", d["final_loss"])')
  echo "$label : $ms" | tee -a $O/r5_ab_t128long_step.txt; }
rm -f $O/r5_ab_t128long_step.txt
for round in 1 2 3 4; do run "tile128x160=1 (default)" --opt tile128x160=1; run "tile128x160=3 (+ whole-K dgrads)" --opt tile128x160=3; done
