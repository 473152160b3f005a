#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r04e; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
(cd $R && timeout 1500 python -m pytest tests -m gpu -q > $O/gputest.log 2>&1; grep -E "passed|failed" $O/gputest.log | tail -2; grep -E "^FAILED|^ERROR" $O/gputest.log | head)
