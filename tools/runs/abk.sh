#!/bin/bash
# same-box interleaved A/B of bench.py option sets, printing step time + the attention / chosen kernel classes:
#   tools/runs/abk.sh "<opts1>" "<opts2>" ...     (ROUNDS=2 by default; CLASSES="attention_fwd attention_bwd")
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
for rnd in $(seq 1 ${ROUNDS:-2}); do
for opts in "$@"; do
  printf "%-34s " "[$opts]"
  timeout 200 python bench.py --steps ${STEPS:-30} --warmup 5 --no-cpu-baseline $opts 2>/dev/null | CLASSES="${CLASSES:-attention_fwd attention_bwd}" python -c "
import sys,json,os
d=json.loads(sys.stdin.readline())
k={r['name']:r for r in d['kernels']}
print(d['ms_per_step'], d['final_loss'], ' '.join('%s %.1f' % (c, k[c]['avg_launch_us']) for c in os.environ['CLASSES'].split() if c in k))"
done; done
