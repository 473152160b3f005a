#!/bin/bash
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "attention" 2>&1 | grep -a "passed\|failed\|rror\|rel err" | tail -6
timeout 600 python bench.py --mode scaled --steps 3 --warmup 1 --parity 2>/dev/null | tail -1 > gpurun_out/scaled_st.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/scaled_st.json')); print(d['ms_per_step'], d['step_tflops'], d['parity_vs_oracle'])
for k in d['kernels'][:6]: print(k)
PY
