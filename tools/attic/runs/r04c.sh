#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r04c; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
(cd $R && SHAPES=16x10x360x80 VARIANTS=3,5,7 ITERS=20 timeout 300 rocprofv3 --kernel-trace --stats -d $O/ks -- python tools/attn_bench.py 2>&1 | grep "^B[0-9]" | cut -c1-160 | tee $O/attn_bench.txt)
python $R/tools/prof_summary.py $O/ks $O/attn_kernel_stats.txt | grep -i attn | cut -c1-200; rm -rf $O/ks
cd $R
ROUNDS=3 tools/runs/abk.sh "--opt attn_variant=3" "--opt attn_variant=5" "--opt attn_variant=7" | tee $O/step_ab.txt
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --breakdown --opt attn_variant=5 2> $O/bench.err | tail -1 > $O/bench_v5.json; tail -2 $O/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04c/bench_v5.json"))
print(d["ms_per_step"])
for r in d["kernels"]:
    print("%-24s launches %5.1f avg %7.2f us  ms/step %6.3f share %5.3f  %s %s" % (r["name"], r["launches_per_step"], r["avg_launch_us"], r["ms_per_step"], r["time_share"], r.get("achieved"), r.get("unit")))
PY
