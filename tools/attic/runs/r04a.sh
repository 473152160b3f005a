#!/bin/bash
# round 4, call A: lean attention backward - parity, per-kernel times old vs lean (one rocprofv3 run), MFMA shape rates,
# same-box step A/B of attn_variant 3 / 5 / 7.  Run from the repo root on the GPU box.
R=$(pwd); O=$R/gpurun_out/r04a; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
$R/tools/bin/mfma_rate_probe > $O/mfma_rate.txt 2>&1; cat $O/mfma_rate.txt
(cd $R && timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "attention" -x > $O/attn_tests.log 2>&1; tail -4 $O/attn_tests.log)
(cd $R && SHAPES=16x10x360x80 VARIANTS=3,5,7 ITERS=20 timeout 300 rocprofv3 --kernel-trace --stats -d $O/ks -- python tools/attn_bench.py 2>&1 | grep "^B[0-9]" | cut -c1-160 | tee $O/attn_bench.txt)
python $R/tools/prof_summary.py $O/ks $O/attn_kernel_stats.txt | grep -i attn | cut -c1-200; rm -rf $O/ks
cd $R
for rnd in 1 2; do
for opts in "--opt attn_variant=3" "--opt attn_variant=5" "--opt attn_variant=7"; do
  printf "%-30s " "[$opts]"
  timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline $opts 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline())
k={r['name']:r for r in d['kernels']}
print(d['ms_per_step'], d['final_loss'], 'attn_fwd %.1f us attn_bwd %.1f us' % (k['attention_fwd']['avg_launch_us'], k['attention_bwd']['avg_launch_us']))"
done; done | tee $O/step_ab.txt
