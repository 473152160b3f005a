#!/bin/bash
# inference forward without the pre-activation store: AR throughput A/B (FACT_KEEP_PRE=1 restores the store) + the AR / forward tests
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
fmt() { python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"; }
for rnd in 1 2; do
  for e in "X=0" "FACT_KEEP_PRE=1"; do
    printf "%-20s B=32: " "[$e]"; env $e timeout 300 python bench.py --mode ar --steps 64 --warmup 4 2>/dev/null | tail -1 | fmt
    printf "%-20s B=1 : " "[$e]"; env $e timeout 300 python bench.py --mode ar --steps 64 --warmup 4 --batch 1 2>/dev/null | tail -1 | fmt
  done
done
timeout 900 python -m pytest tests -m gpu -q -x -k "ar or infer or forward or gemm_nt or golden" 2>&1 | grep -a "passed\|failed\|error" | tail -3
