#!/bin/bash
# A/B of engine options on the headline step: tools/ab.sh "optA=1 optB=2" "optC=0" ...  (each arg = one run's --opt list)
R=$(pwd); mkdir -p $R/gpurun_out
for cfg in "$@"; do
  opts=""; for kv in $cfg; do [ "$kv" != "-" ] && opts="$opts --opt $kv"; done
  ms=$(cd $R && timeout 200 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --profile-steps 1 $opts 2>/dev/null | tail -1 | python -c 'import sys,json; print(json.loads(sys.stdin.read())["ms_per_step"])')
  echo "$cfg : $ms ms" | tee -a $R/gpurun_out/ab.txt
done
