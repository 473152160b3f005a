#!/bin/bash
# A/B of environment settings on one box: tools/runs/abenv.sh "VAR=a" "VAR=b" ...  (bench options in $BOPTS)
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
for rnd in 1 2 3; do
for e in "$@"; do
  printf "%-40s " "[$e]"
  env $e timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline $BOPTS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); k={r['name']:r for r in d['kernels']}; print(d['ms_per_step'], d['final_loss'], 'col_tasks %.1f us' % k['bias/ln_param_grads']['avg_launch_us'])"
done; done
