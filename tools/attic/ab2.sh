#!/bin/bash
# like ab.sh but each arg is a raw bench.py flag string
R=$(pwd); mkdir -p $R/gpurun_out
for cfg in "$@"; do
  ms=$(cd $R && timeout 200 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --profile-steps 1 $cfg 2>/dev/null | tail -1 | python -c 'import sys,json; print(json.loads(sys.stdin.read())["ms_per_step"])')
  echo "$cfg : $ms ms" | tee -a $R/gpurun_out/ab.txt
done
