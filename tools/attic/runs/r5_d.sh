#!/bin/bash
# call D: forward launch boundary (two clocks), write-through epilogue stores A/B, re-run of the adjusted adam test
cd "$(dirname "$0")/../../.."; R=$(pwd); O=$R/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_model.py -q -x -k "adam_in_wgrad" 2>&1 | tail -3 | tee $O/r5_d_tests.txt
{ ITERS=23 python tools/fwd_boundary.py 2>/dev/null | grep FWD_WALL
  cd /tmp && export TMPDIR=/tmp
  ITERS=20 timeout 300 rocprofv3 --kernel-trace --stats -d $O/fb -- python $R/tools/fwd_boundary.py > /dev/null 2>&1
  python $R/tools/prof_summary.py $O/fb $O/fwd_kernel_stats.txt > /dev/null
  python - <<PY
import re
tot=calls=0
for l in open("$O/fwd_kernel_stats.txt"):
    m=re.match(r"^(.*\S)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", l.rstrip())
    if m: calls+=int(m.group(2)); tot+=float(m.group(3))
print("ROCPROF: %d launches, %.1f us of kernel time over 23 forwards (3 warm-up + 20) = %.2f us per forward, %.1f launches per forward" % (calls, tot, tot/23, calls/23))
PY
  rm -rf $O/fb; } 2>&1 | tee $O/r5_fwd_boundary.txt
cd $R
cp mint_amd/lib/libfact_hip_dbg.so /tmp/cur.so
run() { ms=$(FACT_LIB=$1 timeout 200 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --profile-steps 1 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms  loss", d["final_loss"])'); echo "$(basename $1) : $ms" | tee -a $O/r5_ab_wt.txt; }
rm -f $O/r5_ab_wt.txt
for r in 1 2; do run /tmp/cur.so; run tools/bin/libfact_wt1.so; run tools/bin/libfact_wt2.so; done
for l in /tmp/cur.so tools/bin/libfact_wt1.so tools/bin/libfact_wt2.so; do echo "== $l" >> $O/r5_wt_standalone.txt; FACT_LIB=$l timeout 300 python tools/bench_r2.py nt 2>/dev/null | grep -v amdgpu >> $O/r5_wt_standalone.txt; done
cat $O/r5_wt_standalone.txt | cut -c1-200
