#!/bin/bash
# round 5, first GPU call: launch-boundary probe + kernarg placement A/B on the headline step
cd "$(dirname "$0")/../../.."; R=$(pwd); mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/lf2 tools/launch_floor2.hip
for v in unset 0 1; do
  echo "== HIP_FORCE_DEV_KERNARG=$v" | tee -a gpurun_out/r5_launch_floor.txt
  if [ $v = unset ]; then /tmp/lf2; else HIP_FORCE_DEV_KERNARG=$v /tmp/lf2; fi 2>&1 | tee -a gpurun_out/r5_launch_floor.txt
done
rm -f gpurun_out/ab_env.txt
bash tools/attic/ab_env.sh "-" "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0"
cp gpurun_out/ab_env.txt gpurun_out/r5_ab_kernarg.txt
