#!/bin/bash
# same-box A/B of several builds of the library on the headline step: tools/ab_libs.sh tools/bin/libfact_x.so ... (each run
# prints ms/step and the final loss; "cur" = the tree's library, run first and last)
R=$(pwd); cp $R/mint_amd/lib/libfact_hip.so /tmp/cur.so
run() {
  cp $1 $R/mint_amd/lib/libfact_hip.so
  out=$(cd $R && timeout 200 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --profile-steps 1 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms  loss", d["final_loss"], " wgrad in-step us", d["roofline"]["avg_launch_us"])')
  echo "$(basename $1) : $out"
}
run /tmp/cur.so
for r in 1 2; do for l in "$@"; do run $l; done; done
run /tmp/cur.so
cp /tmp/cur.so $R/mint_amd/lib/libfact_hip.so
