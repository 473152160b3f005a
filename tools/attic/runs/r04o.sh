#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r04o; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "tile192" 2>&1 | tail -2
for o in 0 1 0 1; do
timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --opt tile192=$o 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline())
k={r['name']:r for r in d['kernels']}
print($o, d['ms_per_step'], 'ffn1_dgrad %.1f grid %s %s' % (k['ffn1_dgrad']['avg_launch_us'], k['ffn1_dgrad']['grid'], k['ffn1_dgrad']['kernel'][60:120]), 'qkv_dgrad %.1f' % k['qkv_dgrad']['avg_launch_us'])"
done | tee $O/ab2.txt
