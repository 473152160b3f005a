#!/bin/bash
# end-of-round checks: the GPU suite with its log, and the N > 1 control flow of bench.py as a gloo dry run on one GPU
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
mkdir -p gpurun_out/final
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/final/gputest.log 2>&1; grep -a "passed\|failed" gpurun_out/final/gputest.log | tail -2
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 2 --steps 3 --warmup 1 --dist-backend gloo > gpurun_out/final/dry2.json 2> gpurun_out/final/dry2.err; echo "dry run rc=$?"; tail -1 gpurun_out/final/dry2.json | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
