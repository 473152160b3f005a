cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_model.py -q -x -k "autoregressive or reference_code_vectors or evaluator or all_ones" 2>&1 | tail -3
for b in 32 1 4; do for sr in 1 0; do
  echo -n "B=$b sr_rows=$sr : "; timeout 120 python bench.py --mode ar --steps 64 --warmup 4 --batch $b --opt sr_rows=$sr 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms/step", d["value"], "frames/s")'
done; done
