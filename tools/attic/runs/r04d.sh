#!/bin/bash
# round 4, call D: full GPU suite after the ABI split / option plumbing, bench line with engine-generated kernel names next to
# rocprofv3's own names, world-2 gloo dry runs of --mode ar / scaled / train
R=$(pwd); O=$R/gpurun_out/r04d; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
(cd $R && timeout 1200 python -m pytest tests -m gpu -q -x > $O/gputest.log 2>&1; tail -5 $O/gputest.log)
(cd $R && timeout 400 python bench.py --steps 20 --warmup 5 2> $O/bench.err | tail -1 > $O/bench.json; tail -2 $O/bench.err)
(cd $R && timeout 400 rocprofv3 --kernel-trace --stats -d $O/ks -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1)
python $R/tools/prof_summary.py $O/ks $O/kernel_stats.txt | head -24 | cut -c1-150; rm -rf $O/ks
cd $R
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04d/bench.json"))
print(d["ms_per_step"], d["value"], d.get("cpu_baseline"))
stats=open("gpurun_out/r04d/kernel_stats.txt").read()
for r in d["kernels"]:
    hit = all(part in stats for part in r["kernel"].split(" + "))
    print("%-22s cu_share %-6s grid %-6s %s | verbatim in rocprof stats: %s" % (r["name"], r.get("cu_share"), r.get("grid"), r["kernel"][:110], hit))
print(json.dumps(d["roofline"])[:600])
PY
for mode in ar scaled train; do
  extra=""; [ $mode = ar ] && extra="--steps 6 --warmup 2 --batch 4"; [ $mode = scaled ] && extra="--steps 2 --warmup 1 --batch 1"; [ $mode = train ] && extra="--steps 3 --warmup 1"
  timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --mode $mode $extra --no-cpu-baseline --dist-backend gloo > $O/dry2_$mode.json 2> $O/dry2_$mode.err; echo "dry run $mode rc=$? lines=$(grep -c '^{' $O/dry2_$mode.json)"; grep '^{' $O/dry2_$mode.json | cut -c1-700
done
timeout 200 python bench.py --mode ar --steps 6 --warmup 2 --batch 8 > $O/ar1.json 2>/dev/null; cut -c1-400 $O/ar1.json
