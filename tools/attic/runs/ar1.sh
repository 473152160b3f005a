#!/bin/bash
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
fmt() { python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"; }
for rnd in 1 2; do
  for B in 1 2 4 32; do printf "B=%-3s: " $B; timeout 300 python bench.py --mode ar --steps 64 --warmup 4 --batch $B --profile-steps 0 $AROPTS 2>/dev/null | tail -1 | fmt; done
done
timeout 900 python -m pytest tests -m gpu -q -x -k "autoregressive or supervised or reference_code or tiny_forward or e2e or gemm" 2>&1 | grep -a "passed\|failed\|rror" | tail -5
