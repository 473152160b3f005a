"""Does splitting a GEMM chain into two half-batch chains on two streams beat one full-batch chain?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mint_amd import _lib as L
lib = L.lib(); dev = "cuda"

def mk(M, N, K):
    g = torch.Generator(device=dev).manual_seed(0)
    return (torch.randn(M, K, device=dev, generator=g).bfloat16(), (torch.randn(N, K, device=dev, generator=g) * .05).bfloat16(),
            torch.zeros(N, device=dev), torch.empty(M, N, device=dev, dtype=torch.bfloat16), torch.empty(M, N, device=dev, dtype=torch.bfloat16),
            torch.empty(M, N, device=dev), torch.randn(M, N, device=dev))

def chain(M, stream):
    """QKV-like, FFN1, FFN2 chain (shapes of one layer) on `stream`."""
    s = C.c_void_p(stream.cuda_stream)
    for (N, K, epi, key) in ((2400, 800, L.EPI_BF16, "a"), (3072, 800, L.EPI_BIAS_GELU, "b"), (800, 3072, L.EPI_F32_BIAS_RESID, "c")):
        A, B, bias, o0, o1, o32, res = bufs[(M, N, K)]
        out = o32 if epi == L.EPI_F32_BIAS_RESID else o0
        L.check(lib.fact_op_gemm_nt(epi, L.ptr(A), K, L.ptr(B), K, M, N, K, L.ptr(out), N, L.ptr(o1), N, L.ptr(bias), None, 0,
                                    L.ptr(res), N, None, 0, s))

import ctypes as C
bufs = {}
for M in (5760, 2880, 1920):
    for (N, K) in ((2400, 800), (3072, 800), (800, 3072)):
        bufs[(M, N, K)] = mk(M, N, K)
s0, s1, s2 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()

def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

def one():
    cur = torch.cuda.current_stream()
    chain(5760, cur)
def two():
    cur = torch.cuda.current_stream()
    s0.wait_stream(cur); s1.wait_stream(cur)
    chain(2880, s0); chain(2880, s1)
    cur.wait_stream(s0); cur.wait_stream(s1)
def three():
    cur = torch.cuda.current_stream()
    for s in (s0, s1, s2): s.wait_stream(cur)
    for s in (s0, s1, s2): chain(1920, s)
    for s in (s0, s1, s2): cur.wait_stream(s)
print("full batch, 1 stream : %.1f us" % timeit(one))
print("2 x half batch, 2 str: %.1f us" % timeit(two))
print("3 x third batch, 3 st: %.1f us" % timeit(three))
