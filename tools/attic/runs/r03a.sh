#!/bin/bash
# Round 3, GPU call A: the prepared W12 runbook, baseline bench, half-batch-chain probe.
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
O=gpurun_out/r03a; mkdir -p $O
W=tools/bin/libfact_w12.so
echo "== baseline bench" ; timeout 300 python bench.py --steps 20 --warmup 5 --breakdown > $O/bench_base.json 2> $O/bench_base.err; tail -c 600 $O/bench_base.err
echo "== w12 parity"; FACT_LIB=$W FACT_EXPERIMENTAL_W12=1 timeout 400 python -m pytest tests/test_gpu_ops.py -q -k "w12" > $O/w12_parity.log 2>&1; tail -5 $O/w12_parity.log
echo "== w12 bench"; FACT_LIB=$W timeout 300 python tools/bench_r2.py w12 > $O/w12_bench.log 2>&1; cat $O/w12_bench.log
echo "== tn bench"; FACT_LIB=$W TN_LOOPS=0,2,12 timeout 300 python tools/bench_r2.py tn > $O/w12_tn.log 2>&1; cat $O/w12_tn.log
for opts in "" "--opt w12_auto=1" "--opt tn_loop=12" "--opt w12_auto=1 --opt tn_loop=12" "--opt tn_loop=12 --opt wgrad_parts=1"; do
  echo "== step with W12 lib: $opts"
  FACT_LIB=$W timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline $opts 2>>$O/bench_w12.err | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['final_loss'])"
done
echo "== halfbatch probe"
timeout 200 python tools/halfbatch_probe.py fwd 0 0 2>&1 | tail -1
timeout 200 python tools/halfbatch_probe.py fwd 1 0 2>&1 | tail -1
GPU_MAX_HW_QUEUES=7 timeout 200 python tools/halfbatch_probe.py train 0 1 2>&1 | tail -1
timeout 200 python tools/halfbatch_probe.py train 0 0 2>&1 | tail -1
echo done
