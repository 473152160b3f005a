#!/bin/bash
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
O=gpurun_out/r03h; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_dp_world2.py -q -x -k "clip" 2>&1 | tail -15
timeout 900 python bench.py --mode ar --steps 1200 --warmup 2 --parity > $O/ar_b32_1200.json 2> $O/ar.err; cat $O/ar_b32_1200.json; tail -3 $O/ar.err
