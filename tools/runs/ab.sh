#!/bin/bash
# generic A/B of bench.py option sets on one box: tools/runs/ab.sh "<opts1>" "<opts2>" ...   (each run twice, interleaved)
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
for rnd in 1 2; do
for opts in "$@"; do
  printf "%-50s " "[$opts]"
  timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline $opts 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['final_loss'])"
done; done
