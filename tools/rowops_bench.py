"""Stand-alone timing of the HBM-bound row kernels at FACT sizes (HIP-event timing)."""
import os as _os; _os.environ.setdefault("FACT_DEBUG_ABI", "1")  # these tools drive the test / bench surface (mint_amd/_lib.py)
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mint_amd import _lib as L

lib = L.lib()
dev = "cuda"


def timeit(f, iters=30):
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        f()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for M in (5760, 3840, 1920):
    C = 800
    x = torch.randn(M, C, device=dev)
    dh = torch.randn(M, C, device=dev).to(torch.bfloat16)
    gamma, beta = torch.randn(C, device=dev), torch.randn(C, device=dev)
    h = torch.empty(M, C, device=dev, dtype=torch.bfloat16)
    mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
    dres = torch.randn(M, C, device=dev)
    dx = torch.empty(M, C, device=dev); dx16 = torch.empty(M, C, device=dev, dtype=torch.bfloat16)
    dg, db, dbp = torch.zeros(C, device=dev), torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    us = timeit(lambda: L.check(lib.fact_op_ln_fwd(L.ptr(x), L.ptr(gamma), L.ptr(beta), L.ptr(h), L.ptr(mean), L.ptr(rstd), M, C, 1e-5, L.cur_stream())))
    print("ln_fwd M%d: %.1f us (%.0f GB/s)" % (M, us, M * C * 6 / us / 1e3))
    for rows in (8, 16, 24, 32, 48):
        for ws in (0, 1):
            lib.fact_debug_ln_bwd(rows, ws)
            us = timeit(lambda: L.check(lib.fact_op_ln_bwd(L.ptr(dh), L.ptr(x), L.ptr(mean), L.ptr(rstd), L.ptr(gamma), L.ptr(dres), L.ptr(dx), L.ptr(dx16), L.ptr(dg), L.ptr(db), L.ptr(dbp), M, C, L.cur_stream())))
            print("ln_bwd M%d rows %d ws %d: %.1f us (%.0f GB/s)" % (M, rows, ws, us, M * C * 18 / us / 1e3))
    for mode, name in ((5, "dx kernel alone"), (3, "dx + partials (4 rows/wave) + reduce")):
        lib.fact_debug_ln_bwd(8, mode)
        us = timeit(lambda: L.check(lib.fact_op_ln_bwd(L.ptr(dh), L.ptr(x), L.ptr(mean), L.ptr(rstd), L.ptr(gamma), L.ptr(dres), L.ptr(dx), L.ptr(dx16), L.ptr(dg), L.ptr(db), L.ptr(dbp), M, C, L.cur_stream())))
        print("ln_bwd M%d %s: %.1f us (%.0f GB/s)" % (M, name, us, M * C * 16 / us / 1e3))
    lib.fact_debug_ln_bwd(8, 0)
n = 120406977 // 4 * 4
p, m, v, g = (torch.randn(n, device=dev) for _ in range(4))
v.abs_()
us = timeit(lambda: L.check(lib.fact_op_adam(L.ptr(p), L.ptr(m), L.ptr(v), L.ptr(g), n, 1e-4, 0.9, 0.999, 1e-7, L.cur_stream())), 10)
print("adam %d params: %.1f us (%.0f GB/s)" % (n, us, n * 32 / us / 1e3))
