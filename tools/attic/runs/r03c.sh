#!/bin/bash
# Round 3, GPU call C: encoder dgrad split-K A/B + full step dump
cd "$(dirname "$0")/../.."
R=$(pwd); export TMPDIR=/tmp
O=$R/gpurun_out/r03c; mkdir -p $O
for opts in "" "--opt bwd_splitk=2" "--opt bwd_splitk=1" "" "--opt bwd_splitk=2"; do
  echo "== step: $opts"
  timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline $opts 2>>$O/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['final_loss'])"
done
cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $O/kt -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline > $O/bench_prof.json 2> $O/bench_prof.err
cd $R
python tools/step_dump.py $O/kt $O/step_base.tsv
rm -rf $O/kt
