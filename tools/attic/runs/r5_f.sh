#!/bin/bash
# call F: symmetric split-K finish - op parity, stand-alone FFN2 A/B, step A/B
cd "$(dirname "$0")/../../.."; R=$(pwd); O=$R/gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "splitk" 2>&1 | tail -4 | tee $O/r5_f_tests.txt
python - <<'PY' 2>&1 | grep -v amdgpu | tee $O/r5_sksym_standalone.txt
import os; os.environ.setdefault("FACT_DEBUG_ABI", "1")
import sys; sys.path.insert(0, os.getcwd())
import torch
from mint_amd import _lib as L
lib = L.lib(); dev = "cuda"
g = torch.Generator(device=dev).manual_seed(1)
def bench(M, N, K, sym, iters=50):
    A = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16); B = (torch.randn(N, K, device=dev, generator=g) * 0.1).to(torch.bfloat16)
    resid = torch.randn(M, N, device=dev, generator=g); bias = torch.randn(N, device=dev, generator=g); out = torch.empty(M, N, device=dev)
    lib.fact_debug_gemm_sk_sym(sym)
    run = lambda: L.check(lib.fact_op_gemm_nt(L.EPI_F32_BIAS_RESID, L.ptr(A), A.stride(0), L.ptr(B), B.stride(0), M, N, K, L.ptr(out), N, None, 0, L.ptr(bias), None, 0, L.ptr(resid), N, None, 0, L.cur_stream()))
    for _ in range(5): run()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): run()
    b.record(); b.synchronize()
    lib.fact_debug_gemm_sk_sym(1)
    return a.elapsed_time(b) * 1e3 / iters
for rnd in range(3):
    for (M, N, K) in [(5760, 800, 3072)]:
        t0, t1 = bench(M, N, K, 0), bench(M, N, K, 1)
        print("FFN2+resid M%d N%d K%d: exiting finish %.1f us (%.0f TF)  symmetric finish %.1f us (%.0f TF)" % (M, N, K, t0, 2e-6*M*N*K/t0, t1, 2e-6*M*N*K/t1))
PY
run() { local label=$1; shift
  ms=$(timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --profile-steps 1 "$@" 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); k={r["name"]:r for r in d["kernels"]}; print(d["ms_per_step"], "ms  ffn2+resid", k["ffn2+resid"]["avg_launch_us"], "us  loss", d["final_loss"])')
  echo "$label : $ms" | tee -a $O/r5_ab_sksym.txt; }
rm -f $O/r5_ab_sksym.txt
for round in 1 2 3; do run "sk_sym=0" --opt sk_sym=0; run "sk_sym=1" --opt sk_sym=1; done
