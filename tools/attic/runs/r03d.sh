#!/bin/bash
# Round 3, GPU call D: stream priority A/B (optimizer stream low priority with / without bucket hold; side stream priority)
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
run() { echo "== $*"; env "$@" timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline $OPTS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['final_loss'])"; }
OPTS="" run A=1
OPTS="" run FACT_PRIO_OPT=1
OPTS="--opt adam_hold=0" run A=1
OPTS="--opt adam_hold=0" run FACT_PRIO_OPT=1
OPTS="" run FACT_PRIO_SIDE=1
OPTS="" run FACT_PRIO_SIDE=-1
OPTS="" run FACT_PRIO_SIDE=1 FACT_PRIO_OPT=1
OPTS="--opt adam_hold=0" run FACT_PRIO_SIDE=-1 FACT_PRIO_OPT=1
OPTS="" run A=1
