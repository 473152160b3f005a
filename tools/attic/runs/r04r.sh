#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r04r; mkdir -p $O
timeout 300 python tools/rowops_bench.py 2>/dev/null | grep -E "dx only|partials|ln_fwd|rows 8 ws 0" | tee $O/rowops.txt
CLASSES="ln_bwd_dx ln_fwd" ROUNDS=2 tools/runs/abk.sh "--opt ln_dx=1" "--opt ln_dx=2" | tee $O/ab.txt
