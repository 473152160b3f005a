#!/bin/bash
# Round-3 evidence on the GPU box (run from the repo root): gpurun_out/r3/{gputest.log, bench.json, kernel_stats.txt,
# step_pmc.txt, roofline_traffic.json, standalone.txt, ar.json, scaled.json}.  Stages are independent.
# usage: tools/round3_evidence.sh [stages]   stages = subset of "test bench stats pmc standalone ar scaled" (default all)
R=$(pwd); O=$R/gpurun_out/r3; mkdir -p $O
ST=${*:-test bench stats pmc standalone ar scaled}
has() { case " $ST " in *" $1 "*) return 0;; esac; return 1; }
cd /tmp && export TMPDIR=/tmp
if has test; then (cd $R && timeout 900 python -m pytest tests -m gpu -q > $O/gputest.log 2>&1; tail -3 $O/gputest.log); fi
if has bench; then (cd $R && timeout 600 python bench.py --steps 30 --warmup 5 --breakdown 2> $O/bench.err | tail -1 > $O/bench.json; tail -2 $O/bench.err; cut -c1-400 $O/bench.json); fi
if has stats; then
  (cd $R && timeout 400 rocprofv3 --kernel-trace --stats -d $O/ks -- python bench.py --steps 30 --warmup 5 --no-cpu-baseline > /dev/null 2>&1)
  python $R/tools/prof_summary.py $O/ks $O/kernel_stats.txt | head -12; rm -rf $O/ks
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
  (cd $R && { timeout 300 python tools/bench_r2.py nt; timeout 200 python tools/bench_r2.py k64; TN_LOOPS=0,2 timeout 200 python tools/bench_r2.py tn; timeout 200 python tools/rowops_bench.py; } > $O/standalone.txt 2>&1; grep -v amdgpu.ids $O/standalone.txt | head -40)
fi
if has ar; then (cd $R && timeout 300 python bench.py --mode ar --steps 64 --warmup 4 2> $O/ar.err | tail -1 > $O/ar.json; cut -c1-600 $O/ar.json); fi
if has scaled; then (cd $R && timeout 600 python bench.py --mode scaled --steps 3 --warmup 1 --parity 2> $O/scaled.err | tail -1 > $O/scaled.json; cut -c1-900 $O/scaled.json; tail -2 $O/scaled.err); fi
true
