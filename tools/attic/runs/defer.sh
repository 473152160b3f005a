#!/bin/bash
# deferred optimizer step (adam_defer) A/B on one box, with and without a low-priority optimizer stream.
# RECORD of a measured-and-dropped experiment: the engine option `adam_defer` (optimizer buckets after backward in
# forward order, one event each, the next forward waiting layer by layer, a fact_join entry point) was removed again after this
# run - the script no longer runs against the tree.
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
cat > /tmp/_abfmt.py <<'PY'
import sys, json
d = json.loads(sys.stdin.readline())
print(d['ms_per_step'], d['final_loss'])
PY
for rnd in 1 2; do
for cfg in "X=0|" "X=0|--opt adam_defer=1" "FACT_PRIO_OPT=1|--opt adam_defer=1" "FACT_PRIO_OPT=-1|--opt adam_defer=1" "FACT_PRIO_OPT=1|"; do
  e=${cfg%%|*}; o=${cfg#*|}
  printf "%-50s " "[$e $o]"
  env $e timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline $o 2>/dev/null | python /tmp/_abfmt.py
done; done
