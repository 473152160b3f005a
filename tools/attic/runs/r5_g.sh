#!/bin/bash
# call G: 128x160 tiles for the short-K N = 800 GEMMs - parity, stand-alone, step A/B; AR 64-frame numbers
cd "$(dirname "$0")/../../.."; R=$(pwd); O=$R/gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "tile128x160" 2>&1 | tail -3 | tee $O/r5_g_tests.txt
timeout 300 python tools/bench_r2.py t128 2>/dev/null | grep -v amdgpu | tee $O/r5_t128_standalone.txt
run() { local label=$1; shift
  ms=$(timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --profile-steps 1 "$@" 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); k={r["name"]:r for r in d["kernels"]}; print(d["ms_per_step"], "ms  out_proj+resid", k["out_proj+resid"]["avg_launch_us"], " out_proj_dgrad", k["out_proj_dgrad+heads"]["avg_launch_us"])')
  echo "$label : $ms" | tee -a $O/r5_ab_t128.txt; }
rm -f $O/r5_ab_t128.txt
for round in 1 2 3; do run "tile128x160=0" --opt tile128x160=0; run "tile128x160=1" --opt tile128x160=1; done
timeout 900 python -m pytest tests/test_gpu_model.py -q -x -k "64_frames" -s 2>&1 | grep -E "AR B=32|passed|failed" | tee $O/r5_ar64.txt
