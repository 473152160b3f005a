#!/bin/bash
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -x 2>&1 | tail -4
bash tools/runs/ab.sh "--opt skinny_fuse=0" "--opt skinny_fuse=1"
for f in 0 1; do for B in 1 32; do
  echo -n "AR B=$B skinny_fuse=$f: "; timeout 200 python bench.py --mode ar --steps 64 --warmup 4 --batch $B --opt skinny_fuse=$f 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], 'frames/s', d['ms_per_step'], 'ms/step')"
done; done
