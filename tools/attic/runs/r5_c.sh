#!/bin/bash
# round 5, call C: tests of the changed pieces + upper bound of an LN fusion (timing-only skip ablation) + adam_in_wgrad on configs[4]
cd "$(dirname "$0")/../../.."; mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_production_lib.py -q -x -k "adam_in_wgrad or production" 2>&1 | tail -5
  timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "attention" 2>&1 | tail -3 ) | tee gpurun_out/r5_c_tests.txt
run() { local label=$1; shift
  ms=$(timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --profile-steps 1 "$@" 2>/dev/null | tail -1 | python -c 'import sys,json; print(json.loads(sys.stdin.read())["ms_per_step"])')
  echo "$label : $ms ms" | tee -a gpurun_out/r5_ab_lnskip.txt; }
rm -f gpurun_out/r5_ab_lnskip.txt
for round in 1 2; do
  run "baseline"
  run "skip ln_fwd (512)" --opt skip=512
  run "skip ln_bwd (8)" --opt skip=8
  run "skip both (520)" --opt skip=520
done
timeout 300 python tools/attn_bench.py > gpurun_out/r5_attn_bench.txt 2>&1; tail -12 gpurun_out/r5_attn_bench.txt
for iw in 0 1 0 1; do
  timeout 400 python bench.py --mode scaled --steps 6 --warmup 2 --profile-steps 0 --opt adam_in_wgrad=$iw 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("scaled adam_in_wgrad='$iw' :", d["ms_per_step"], "ms")' | tee -a gpurun_out/r5_ab_scaled_adam.txt
done
