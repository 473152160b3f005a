#!/bin/bash
cd "$(dirname "$0")/../../.."; mkdir -p gpurun_out
timeout 600 python tools/attic/runs/r5_adam_diag.py > gpurun_out/r5_adam_diag.txt 2>&1; tail -60 gpurun_out/r5_adam_diag.txt
