"""The split-K atomic GEMM of the small-batch AR sampler (M = 360 rows) as a function of the K split: what does the
12.7 us launch consist of?  Weights rotated over 16 copies (a frame streams 16 layers of weights: no L2 reuse).
python tools/skinny_bench.py"""
import os as _os; _os.environ.setdefault("FACT_DEBUG_ABI", "1")  # these tools drive the test / bench surface (mint_amd/_lib.py)
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mint_amd import _lib as L

lib = L.lib(); dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)


def case(M, N, K, label, copies=16):
    ld = (K + 63) // 64 * 64
    A = torch.zeros(M, ld, device=dev, dtype=torch.bfloat16); A[:, :K] = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
    Bs = [torch.zeros(N, ld, device=dev, dtype=torch.bfloat16) for _ in range(copies)]
    for b in Bs:
        b[:, :K] = (torch.randn(N, K, device=dev, generator=g) * 0.05).to(torch.bfloat16)
    acc = torch.zeros(M, (N + 3) // 4 * 4, device=dev)
    line = "%-10s M%4d N%5d K%5d:" % (label, M, N, K)
    for sk in (1, 2, 3, 4, 6, 8, 12):
        if sk > (K + 63) // 64 // 2 and sk > 1:
            continue
        it = [0]

        def launch():
            b = Bs[it[0] % copies]; it[0] += 1
            L.check(lib.fact_op_gemm_nt(L.EPI_ATOMIC_F32, L.ptr(A), ld, L.ptr(b), ld, M, N, K, L.ptr(acc), acc.stride(0), None, 0,
                                        None, None, sk, None, 0, None, 0, L.cur_stream()))
        for _ in range(4): launch()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(64): launch()
        e1.record(); e1.synchronize()
        tiles = ((M + 127) // 128) * ((N + 127) // 128)
        line += "  sk%-2d %5.1fus (%3d wg)" % (sk, e0.elapsed_time(e1) / 64 * 1e3, tiles * sk)
    print(line, flush=True)


for M in (360,):
    case(M, 2400, 800, "QKV")
    case(M, 800, 800, "out-proj")
    case(M, 3072, 800, "FFN1")
    case(M, 800, 3072, "FFN2")
