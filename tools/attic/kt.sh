#!/bin/bash
# in-step kernel-class table for a given option set: tools/kt.sh "opt=v opt=v"
R=$(pwd)
for cfg in "$@"; do
  opts=""; for kv in $cfg; do [ "$kv" != "-" ] && opts="$opts --opt $kv"; done
  echo "== $cfg"
  (cd $R && timeout 200 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --profile-steps 4 $opts 2>/dev/null | tail -1 | python -c '
import sys,json
b=json.loads(sys.stdin.read())
print("ms_per_step", b["ms_per_step"])
for r in b["kernels"]: print("  %-22s n=%5.1f avg %7.2fus %6.3f ms/step %5.1f%% %8.1f %s"%(r["name"],r["launches_per_step"],r["avg_launch_us"],r["ms_per_step"],100*r["time_share"],r["achieved"],r["unit"]))
')
done
