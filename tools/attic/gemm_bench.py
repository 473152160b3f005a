"""Micro-benchmark of the NT GEMM kernel variants at the FACT shapes (HIP-event timing)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mint_amd import _lib as L

lib = L.lib()
dev = "cuda"
SHAPES = [  # M, N, K, epi
    (5760, 3072, 800, L.EPI_BIAS_GELU), (5760, 800, 3072, L.EPI_F32_BIAS_RESID), (5760, 2400, 800, L.EPI_BF16),
    (5760, 800, 800, L.EPI_F32_BIAS_RESID), (5760, 800, 2400, L.EPI_BF16), (5760, 3072, 800, L.EPI_GELU_BWD),
    (5760, 3072, 800, L.EPI_BF16), (5760, 3072, 3072, L.EPI_BF16), (8192, 8192, 8192, L.EPI_BF16),
    (3840, 3072, 800, L.EPI_BIAS_GELU), (1920, 3072, 800, L.EPI_BIAS_GELU), (1920, 800, 3072, L.EPI_F32_BIAS_RESID),
]


def run(M, N, K, epi, variant, iters=30, pitch=None):
    g = torch.Generator(device=dev).manual_seed(0)
    ld = pitch or K
    A = torch.randn(M, ld, device=dev, generator=g).to(torch.bfloat16)
    B = (torch.randn(N, ld, device=dev, generator=g) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device=dev, generator=g)
    resid = torch.randn(M, N, device=dev, generator=g)
    pre = torch.randn(M, N, device=dev, generator=g).to(torch.bfloat16)
    f32 = epi in (L.EPI_F32_BIAS_RESID,)
    o0 = torch.empty(M, N, device=dev, dtype=torch.float32 if f32 else torch.bfloat16)
    o1 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    lib.fact_debug_gemm_nt_variant(variant)

    def launch():
        L.check(lib.fact_op_gemm_nt(epi, L.ptr(A), ld, L.ptr(B), ld, M, N, K, L.ptr(o0), N, L.ptr(o1), N, L.ptr(bias),
                                    None, 0, L.ptr(resid), N, L.ptr(pre), N, L.cur_stream()))
    for _ in range(3):
        launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        launch()
    e1.record()
    e1.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    ref = A[:, :K].float() @ B[:, :K].float().t()
    if epi == L.EPI_BF16:
        err = ((o0.float() - ref).norm() / ref.norm()).item()
    elif epi == L.EPI_F32_BIAS_RESID:
        err = ((o0 - (ref + bias + resid)).norm() / ref.norm()).item()
    elif epi == L.EPI_BIAS_GELU:
        err = ((o0.float() - (ref + bias)).norm() / ref.norm()).item()
    else:
        err = float("nan")
    lib.fact_debug_gemm_nt_variant(0)
    return us, 2.0 * M * N * K / us / 1e6, err


if len(sys.argv) > 1 and sys.argv[1] == "shape":  # ad-hoc shapes: shape M N K [M N K ...]
    a = list(map(int, sys.argv[2:]))
    for i in range(0, len(a), 3):
        M, N, K = a[i:i + 3]
        for v in map(int, os.environ.get("NT_VARIANTS", "1").split(",")):
            lib.fact_debug_gemm_nt_band(int(os.environ.get("NT_BAND", "8")))
            us, tf, err = run(M, N, K, L.EPI_BF16, v)
            tiles = ((M + 127) // 128) * ((N + 127) // 128)
            print("M%d N%d K%d v%d: %.1f us %.0f TF  tiles %d  us/k64-iter/round %.3f  err %.1e" % (
                M, N, K, v, us, tf, tiles, us / (K / 64) / -(-tiles // 512), err))
    sys.exit(0)
if len(sys.argv) > 1:  # single shape/variant (for rocprofv3 --pmc runs): idx variant iters
    M, N, K, epi = SHAPES[int(sys.argv[1])]
    lib.fact_debug_gemm_nt_band(int(os.environ.get("NT_BAND", "8")))
    us, tf, err = run(M, N, K, epi, int(sys.argv[2]), int(sys.argv[3]), pitch=int(os.environ.get("PITCH", "0")) or None)
    print("M%d N%d K%d epi%d v%s: %.1f us %.0f TF" % (M, N, K, epi, sys.argv[2], us, tf))
    sys.exit(0)
for (M, N, K, epi) in ([] if os.environ.get("TN_ONLY") else SHAPES):
    line = "M%5d N%5d K%5d epi%d:" % (M, N, K, epi)
    for (band, pitch) in ((1, None), (8, None), (4, None), (8, (K + 63) // 64 * 64)):
        lib.fact_debug_gemm_nt_band(band)
        us, tf, err = run(M, N, K, epi, 1, pitch=pitch)
        line += "  b%d ld%s %6.1fus %4.0fTF %s" % (band, pitch or K, us, tf, "" if (err < 5e-3 or err != err) else "ERR %.1e" % err)
    lib.fact_debug_gemm_nt_band(8)
    print(line, flush=True)

print("--- TN (wgrad) ---")
TN_SHAPES = [tuple(map(int, os.environ["TN_SHAPE"].split(",")))] if os.environ.get("TN_SHAPE") else [(5760, 800, 3072), (5760, 3072, 800), (5760, 2400, 800), (5760, 800, 800), (1920, 3072, 800), (3840, 2400, 800)]
for (K, Mo, No) in TN_SHAPES:
    g = torch.Generator(device=dev).manual_seed(0)
    A = torch.randn(K, Mo, device=dev, generator=g).to(torch.bfloat16)
    B = torch.randn(K, No, device=dev, generator=g).to(torch.bfloat16)
    ref = A.float().t() @ B.float()
    slab = torch.empty(6 * Mo * No, device=dev)
    for tv in (1, 2):  # 1 = atomics, 2 = slabs + reduce
        line = "K%5d Mo%5d No%5d mode%d:" % (K, Mo, No, tv)
        for splitk in (1, 2, 3, 4, 6):
            out = torch.zeros(Mo, No, device=dev)

            def launch():
                L.check(lib.fact_op_gemm_tn(L.ptr(A), Mo, L.ptr(B), No, Mo, No, K, L.ptr(out), No, splitk, tv,
                                            L.ptr(slab), L.cur_stream()))
            launch()
            torch.cuda.synchronize()
            err = ((out - ref).norm() / ref.norm()).item()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                launch()
            e1.record(); e1.synchronize()
            us = e0.elapsed_time(e1) / 20 * 1e3
            line += "  sk%d %6.1fus %5.0fTF e%.0e" % (splitk, us, 2.0 * K * Mo * No / us / 1e6, err)
        print(line, flush=True)
    lib.fact_debug_gemm_nt_variant(0)
