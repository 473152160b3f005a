#!/bin/bash
# round 4, call B: fixed MFMA probe, counter list, lean-backward parity again, PMC passes over old / lean attention kernels
R=$(pwd); O=$R/gpurun_out/r04b; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
$R/tools/bin/mfma_rate_probe > $O/mfma_rate.txt 2>&1; cat $O/mfma_rate.txt
rocprofv3 -L > $O/counters.txt 2>&1; grep -c . $O/counters.txt
(cd $R && timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "attention" > $O/attn_tests.log 2>&1; tail -4 $O/attn_tests.log)
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1)); d=$O/pmc$i
  (cd $R && SHAPES=16x10x360x80 VARIANTS=3,5,7 ITERS=3 timeout 300 rocprofv3 --kernel-trace --pmc $set -d $d -- python tools/attn_bench.py > /dev/null 2>&1)
  python $R/tools/pmc_summary.py $d attn > $O/pmc$i.txt 2>&1; rm -rf $d
done
cat $O/pmc1.txt $O/pmc2.txt | head -150
