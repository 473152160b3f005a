#!/bin/bash
# A/B of environment settings on one box: tools/runs/abenv.sh "VAR=a" "VAR=b" ...  (bench options in $BOPTS; KCLASS = kernel
# class of the in-step table to print beside the step time)
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
cat > /tmp/_abfmt.py <<'PY'
import sys, json, os
d = json.loads(sys.stdin.readline())
k = {r['name']: r for r in d['kernels']}
c = os.environ.get("KCLASS", "attention_bwd")
print(d['ms_per_step'], d['final_loss'], '%s %.1f us' % (c, k[c]['avg_launch_us']))
PY
for rnd in 1 2 3; do
for e in "$@"; do
  printf "%-40s " "[$e]"
  env $e timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline $BOPTS 2>/dev/null | python /tmp/_abfmt.py
done; done
