#!/bin/bash
# sample power / clocks while the headline bench runs (is the step power-limited?)
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
(cd $R && python bench.py --steps 1500 --warmup 10 --no-cpu-baseline --profile-steps 0 $BENCH_OPTS > $O/power_bench.json 2>/dev/null) &
BP=$!
sleep 9
for i in 1 2 3 4 5 6; do
  /opt/rocm/bin/rocm-smi --showpower --showclocks --showuse 2>/dev/null | grep -i "power\|sclk\|mclk\|busy\|fclk" | tr -s " " | head -8
  echo --
  sleep 1.2
done
wait $BP
python -c "import json; print('ms_per_step', json.load(open('$O/power_bench.json'))['ms_per_step'])"
/opt/rocm/bin/rocm-smi --showmaxpower 2>/dev/null | grep -i power | head -3
