#!/bin/bash
# round 5: parity of the optimizer-in-wgrad path, then same-box A/B of the headline step (interleaved, 2 rounds)
cd "$(dirname "$0")/../../.."; R=$(pwd); mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "tn_group" 2>&1 | tail -5 | tee gpurun_out/r5_adam_tests.txt
timeout 1200 python -m pytest tests/test_gpu_model.py -q -x -k "adam_in_wgrad or fused_optimizer or adam_fused or grad_overwrite or options_survive" 2>&1 | tail -5 | tee -a gpurun_out/r5_adam_tests.txt
run() {  # $1 = label, rest = bench args
  local label=$1; shift
  ms=$(timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --profile-steps 1 "$@" 2>/dev/null | tail -1 | python -c 'import sys,json; print(json.loads(sys.stdin.read())["ms_per_step"])')
  echo "$label : $ms ms" | tee -a gpurun_out/r5_ab_adam.txt
}
rm -f gpurun_out/r5_ab_adam.txt
for round in 1 2; do
  run "bucket path (adam_in_wgrad=0)" --opt adam_in_wgrad=0
  run "adam_in_wgrad=1" --opt adam_in_wgrad=1
  run "adam_in_wgrad=1 tn_loop=2" --opt adam_in_wgrad=1 --opt tn_loop=2
  run "adam_in_wgrad=1 ln_cs=1" --opt adam_in_wgrad=1 --opt ln_cs=1
  run "adam_in_wgrad=1 ln_cs=1 tn_loop=2" --opt adam_in_wgrad=1 --opt ln_cs=1 --opt tn_loop=2
  run "adam_in_wgrad=1 wgrad_parts=1" --opt adam_in_wgrad=1 --opt wgrad_parts=1
  run "adam_in_wgrad=1 wgrad_parts=3" --opt adam_in_wgrad=1 --opt wgrad_parts=3
done
python bench.py --steps 30 --warmup 8 --no-cpu-baseline --profile-steps 3 > gpurun_out/r5_bench_adam_in_wgrad.json 2> gpurun_out/r5_bench_adam_in_wgrad.err
tail -c 600 gpurun_out/r5_bench_adam_in_wgrad.err
