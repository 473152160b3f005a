#!/bin/bash
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -k "autoregressive or supervised or reference_code or tiny_forward or e2e" 2>&1 | grep -a "passed\|failed\|rror" | tail -5
fmt() { python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"; }
for rnd in 1 2; do
  for o in "" "--opt ln_fuse=0"; do
    printf "%-20s B=1 : " "[$o]"; timeout 300 python bench.py --mode ar --steps 64 --warmup 4 --batch 1 --profile-steps 0 $o 2>/dev/null | tail -1 | fmt
    printf "%-20s B=32: " "[$o]"; timeout 300 python bench.py --mode ar --steps 64 --warmup 4 --profile-steps 0 $o 2>/dev/null | tail -1 | fmt
  done
done
