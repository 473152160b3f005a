#!/bin/bash
# A/B of runtime environment settings on the headline step: tools/ab_env.sh "VAR=1" "-" "VAR=0 OTHER=2" ...  (each arg = one run's env; "-" = none)
R=$(pwd); mkdir -p $R/gpurun_out
for round in 1 2; do
for cfg in "$@"; do
  envs=""; [ "$cfg" != "-" ] && envs="$cfg"
  ms=$(cd $R && env $envs timeout 200 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --profile-steps 1 2>/dev/null | tail -1 | python -c 'import sys,json; print(json.loads(sys.stdin.read())["ms_per_step"])')
  echo "$cfg : $ms ms" | tee -a $R/gpurun_out/ab_env.txt
done
done
