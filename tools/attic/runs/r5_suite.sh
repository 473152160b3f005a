#!/bin/bash
# full GPU suite (hidden-visibility build, two libraries) + the diagnostics of the adam-in-wgrad model test
cd "$(dirname "$0")/../../.."; mkdir -p gpurun_out
timeout 1700 python -m pytest tests -m gpu -q -x 2>&1 | tail -30 > gpurun_out/r5_suite.txt
cat gpurun_out/r5_suite.txt
