import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mint_amd import _lib as L
lib = L.lib(); dev = "cuda"
def t(fn, it=30):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
for (M, N, K) in ((5760, 800, 3072), (5760, 800, 2400), (5760, 800, 800), (5760, 3072, 800), (5760, 2400, 800)):
    g = torch.Generator(device=dev).manual_seed(0)
    A = torch.randn(M, K, device=dev, generator=g).bfloat16(); B = (torch.randn(N, K, device=dev, generator=g) * .05).bfloat16()
    bias = torch.zeros(N, device=dev); res = torch.randn(M, N, device=dev); out = torch.zeros(M, N, device=dev)
    line = "M%d N%d K%d: fused-epilogue sk1 %.1fus |" % (M, N, K, t(lambda: lib.fact_op_gemm_nt(
        L.EPI_F32_BIAS_RESID, L.ptr(A), K, L.ptr(B), K, M, N, K, L.ptr(out), N, None, 0, L.ptr(bias), None, 0, L.ptr(res), N, None, 0, L.cur_stream())))
    for sk in (1, 2, 3, 4):
        us = t(lambda: lib.fact_op_gemm_nt(L.EPI_ATOMIC_F32, L.ptr(A), K, L.ptr(B), K, M, N, K, L.ptr(out), N, None, 0, None, None, sk,
                                           None, 0, None, 0, L.cur_stream()))
        line += " atomic sk%d %.1fus" % (sk, us)
    print(line, flush=True)
