#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r04p; mkdir -p $O
PYTHONPATH=. python tools/attic/host_enqueue2.py 2>/dev/null | tee $O/host.txt
