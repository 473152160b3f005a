#!/bin/bash
# Round-6 evidence on the GPU box (run from the repo root): gpurun_out/r6/{gputest.log, bench.json, kernel_stats.txt,
# step_pmc.txt, roofline_traffic.json, standalone.txt, attn_kernel_stats.txt, attn_pmc.txt, ar.json, scaled.json,
# dry2_*.json}.  Stages are independent.  The summaries that are to be judged are copied to profiles/r06_* by hand.
# usage: tools/round6_evidence.sh [stages]   stages = subset of "test bench stats pmc standalone attn ar scaled dry"
R=$(pwd); O=$R/gpurun_out/r6; mkdir -p $O
ST=${*:-test bench stats pmc standalone attn ar scaled dry}
has() { case " $ST " in *" $1 "*) return 0;; esac; return 1; }
cd /tmp && export TMPDIR=/tmp
if has test; then (cd $R && timeout 1500 python -m pytest tests -m gpu -q > $O/gputest.log 2>&1; tail -3 $O/gputest.log); fi
if has bench; then (cd $R && timeout 600 python bench.py --steps 30 --warmup 5 --breakdown 2> $O/bench.err | tail -1 > $O/bench.json; tail -2 $O/bench.err; cut -c1-400 $O/bench.json); fi
if has stats; then
  (cd $R && timeout 400 rocprofv3 --kernel-trace --stats -d $O/ks -- python bench.py --steps 30 --warmup 5 --no-cpu-baseline --profile-steps 0 > /dev/null 2>&1)
  python $R/tools/prof_summary.py $O/ks $O/kernel_stats.txt | head -12
  # dependent-launch boundary IN the step, from the same trace: idle gaps between consecutive kernels per hardware queue
  # (the forward runs on one queue with no co-runner: its median gap is what a launch boundary costs on the GPU side)
  db=$(find $O/ks -name '*.db' | head -1)
  { echo "# tools/trace_gaps.py on the rocprofv3 --kernel-trace database of: python bench.py --steps 30 --warmup 5 --no-cpu-baseline --profile-steps 0"; python $R/tools/trace_gaps.py $db; } > $O/launch_boundary.txt 2>&1; head -5 $O/launch_boundary.txt
  rm -rf $O/ks
fi
if has pmc; then
  echo "# rocprofv3 --kernel-trace --pmc <set> -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --profile-steps 0 (MI355X); tools/step_pmc.py" > $O/step_pmc.txt
  i=0; dirs=""
  for set in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1)); d=$O/pmc$i; dirs="$dirs $d"
    (cd $R && timeout 300 rocprofv3 --kernel-trace --pmc $set -d $d -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --profile-steps 0 > /dev/null 2>&1)
  done
  python $R/tools/step_pmc.py $O/step_pmc.txt $dirs --traffic-json $O/roofline_traffic.json | head -16; rm -rf $dirs
fi
if has standalone; then
  [ -x $R/tools/bin/mfma_rate_probe ] || (mkdir -p $R/tools/bin && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-result $R/tools/mfma_rate_probe.hip -o $R/tools/bin/mfma_rate_probe)
  (cd $R && { timeout 300 python tools/bench_r2.py nt; timeout 200 python tools/bench_r2.py k64; timeout 200 python tools/bench_r2.py m32; TN_LOOPS=0,2 timeout 200 python tools/bench_r2.py tn; timeout 200 python tools/rowops_bench.py; $R/tools/bin/mfma_rate_probe; } > $O/standalone.txt 2>&1; grep -v amdgpu.ids $O/standalone.txt | head -40)
fi
if has attn; then
  # the default pairing (5: streaming forward + lean resident backward) and the streaming family (2), one profiler run, then two PMC passes
  (cd $R && SHAPES=16x10x360x80 VARIANTS=5,2 ITERS=20 timeout 300 rocprofv3 --kernel-trace --stats -d $O/ks -- python tools/attn_bench.py 2>&1 | grep "^B[0-9]" | cut -c1-160 > $O/attn_bench.txt)
  python $R/tools/prof_summary.py $O/ks $O/attn_kernel_stats.txt | grep -i attn | cut -c1-200; rm -rf $O/ks
  : > $O/attn_pmc.txt
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    (cd $R && SHAPES=16x10x360x80 VARIANTS=5 ITERS=3 timeout 300 rocprofv3 --kernel-trace --pmc $set -d $O/pm -- python tools/attn_bench.py > /dev/null 2>&1)
    python $R/tools/pmc_summary.py $O/pm attn >> $O/attn_pmc.txt 2>&1; rm -rf $O/pm
  done
fi
if has ar; then (cd $R && timeout 300 python bench.py --mode ar --steps 64 --warmup 4 2> $O/ar.err | tail -1 > $O/ar.json; cut -c1-600 $O/ar.json); fi
if has scaled; then (cd $R && timeout 600 python bench.py --mode scaled --steps 3 --warmup 1 --parity 2> $O/scaled.err | tail -1 > $O/scaled.json; cut -c1-900 $O/scaled.json; tail -2 $O/scaled.err); fi
if has dry; then
  # world-2 gloo dry runs on the one-GPU box (control flow of the N > 1 paths only, not measurements): bench.py --gpus 2 is
  # its OWN launcher (round 5) - no torch.distributed.run in the command
  for mode in ar scaled train; do
    extra="--steps 3 --warmup 1"; [ $mode = ar ] && extra="--steps 6 --warmup 2 --batch 4"; [ $mode = scaled ] && extra="--steps 2 --warmup 1 --batch 1"
    (cd $R && timeout 400 python bench.py --gpus 2 --mode $mode $extra --no-cpu-baseline --dist-backend gloo 2> $O/dry2_$mode.err | grep '^{' > $O/dry2_$mode.json; echo "dry run $mode: $(wc -l < $O/dry2_$mode.json) JSON line(s), n_gpus $(python -c "import json;print(json.loads(open('$O/dry2_$mode.json').readline())['n_gpus'])" 2>/dev/null)")
  done
fi
true
