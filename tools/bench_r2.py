"""Round-2 GEMM micro-benchmarks (HIP-event timing, one process, interleaved variants):
  * NT shapes of a FACT cross-modal layer at B = 16: 128x128 kernel vs the round-1 big-tile kernel vs the
    big-tile family of gemm_big.hip (288x256 / 256x256 / 256x160), each with the epilogue the engine uses;
  * a layer's four weight gradients: round-1 path (128x128 TN kernel, split-K slabs + reduce, 4 + 4 launches)
    vs the grouped whole-K launch (160x256 tiles).
Prints one line per (shape, variant): microseconds, TFLOP/s, relative error vs torch fp32."""
import os as _os; _os.environ.setdefault("FACT_DEBUG_ABI", "1")  # these tools drive the test / bench surface (mint_amd/_lib.py)
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mint_amd import _lib as L

lib = L.lib()
dev = "cuda"
ITERS = int(os.environ.get("ITERS", "30"))


def time_us(fn, iters=None):
    iters = iters or ITERS
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def nt_case(M, N, K, epi, variants, label):
    g = torch.Generator(device=dev).manual_seed(0)
    ld = (K + 63) // 64 * 64
    A = torch.zeros(M, ld, device=dev, dtype=torch.bfloat16)
    B = torch.zeros(N, ld, device=dev, dtype=torch.bfloat16)
    A[:, :K] = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
    B[:, :K] = (torch.randn(N, K, device=dev, generator=g) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device=dev, generator=g)
    resid = torch.randn(M, N, device=dev, generator=g)
    ldn = (N + 63) // 64 * 64
    pre = torch.randn(M, ldn, device=dev, generator=g).to(torch.bfloat16)
    f32 = epi in (L.EPI_F32_BIAS_RESID,)
    o0 = torch.empty(M, N if f32 else ldn, device=dev, dtype=torch.float32 if f32 else torch.bfloat16)
    o1 = torch.empty(M, ldn, device=dev, dtype=torch.bfloat16)
    ref = A[:, :K].float() @ B[:, :K].float().t()
    line = "%-14s M%5d N%5d K%5d:" % (label, M, N, K)
    for v in variants:
        lib.fact_debug_gemm_splitk_max(1 if v in (112, 117) else 4)   # 112 / 117 = 256x160 tiles WITHOUT the in-kernel split-K
        lib.fact_debug_gemm_nt_variant(12 if v == 112 else 17 if v == 117 else v)

        def launch():
            L.check(lib.fact_op_gemm_nt(epi, L.ptr(A), ld, L.ptr(B), ld, M, N, K, L.ptr(o0), o0.stride(0), L.ptr(o1),
                                        ldn, L.ptr(bias), None, 0, L.ptr(resid), N, L.ptr(pre), ldn, L.cur_stream()))
        us = time_us(launch)
        if epi == L.EPI_BF16:
            err = ((o0[:, :N].float() - ref).norm() / ref.norm()).item()
        elif epi == L.EPI_F32_BIAS_RESID:
            err = ((o0 - (ref + bias + resid)).norm() / ref.norm()).item()
        elif epi == L.EPI_BIAS_GELU:
            err = ((o0[:, :N].float() - (ref + bias)).norm() / ref.norm()).item()
        else:
            err = float("nan")
        line += "  v%-2d %6.1fus %5.0fTF%s" % (v, us, 2.0 * M * N * K / us / 1e6,
                                              "" if (err < 5e-3 or err != err) else " ERR%.1e" % err)
    lib.fact_debug_gemm_nt_variant(0)
    lib.fact_debug_gemm_splitk_max(4)
    print(line, flush=True)


def wgrad_layer(K, d=800, ff=3072):
    g = torch.Generator(device=dev).manual_seed(1)
    dp, fp, qp = (d + 63) // 64 * 64, (ff + 63) // 64 * 64, (3 * d + 63) // 64 * 64

    def mk(cols, ld):
        t = torch.zeros(K, ld, device=dev, dtype=torch.bfloat16)
        t[:, :cols] = torch.randn(K, cols, device=dev, generator=g).to(torch.bfloat16)
        return t
    xin, gact, h2, dpre, att, xmid, h1, dqkv = (mk(d, dp), mk(ff, fp), mk(d, dp), mk(ff, fp), mk(d, dp), mk(d, dp),
                                                mk(d, dp), mk(3 * d, qp))
    probs_old = [(gact, ff, xin, d), (h2, d, dpre, ff), (att, d, xmid, d), (h1, d, dqkv, 3 * d)]  # out [Mo][No]
    outs_old = [torch.zeros(p[1], p[3], device=dev) for p in probs_old]
    slab = torch.empty(6 * d * max(ff, 3 * d) + 64, device=dev)

    def splitk_for(Mo, No):  # engine rule (engine.hip wgrad())
        tiles = ((Mo + 127) // 128) * ((No + 127) // 128)
        sk = max(1, min(6, 560 // tiles, (K + 63) // 64 // 4))
        return sk

    def old():
        for (A, Mo, Bm, No), o in zip(probs_old, outs_old):
            L.check(lib.fact_op_gemm_tn(L.ptr(A), A.stride(0), L.ptr(Bm), Bm.stride(0), Mo, No, K, L.ptr(o), No,
                                        splitk_for(Mo, No), 2, L.ptr(slab), L.cur_stream()))
    outs_new = [torch.zeros(ff, d, device=dev), torch.zeros(d, ff, device=dev), torch.zeros(d, d, device=dev),
                torch.zeros(d, 3 * d, device=dev)]
    probs_new = [(xin, d, gact, ff, outs_new[0], 1), (h2, d, dpre, ff, outs_new[1], 0),
                 (att, d, xmid, d, outs_new[2], 0), (h1, d, dqkv, 3 * d, outs_new[3], 0)]
    n = 4
    VP, IA = C.c_void_p * n, C.c_int * n
    a_ = VP(*[p[0].data_ptr() for p in probs_new]); lda = IA(*[p[0].stride(0) for p in probs_new])
    b_ = VP(*[p[2].data_ptr() for p in probs_new]); ldb = IA(*[p[2].stride(0) for p in probs_new])
    o_ = VP(*[p[4].data_ptr() for p in probs_new]); ldo = IA(*[p[4].stride(0) for p in probs_new])
    mo = IA(*[p[1] for p in probs_new]); no = IA(*[p[3] for p in probs_new]); tr = IA(*[p[5] for p in probs_new])

    def new():
        L.check(lib.fact_op_gemm_tn_group(n, a_, lda, b_, ldb, o_, ldo, mo, no, tr, K, L.cur_stream()))
    flops = 2.0 * K * (d * ff * 2 + d * d + d * 3 * d)
    for o in outs_old + outs_new:
        o.zero_()
    old(); new()
    torch.cuda.synchronize()
    errs = [((outs_new[0] - outs_old[0]).norm() / outs_old[0].norm()).item()]
    errs += [((outs_new[i] - outs_old[i]).norm() / outs_old[i].norm()).item() for i in (1, 2, 3)]
    t_old, t_new = time_us(old), time_us(new)
    line = ""
    for cfg in [int(x) for x in os.environ.get('TN_LOOPS', '1,2,0').split(',')]:
        lib.fact_debug_gemm_tn_cfg(cfg + 256 * int(os.environ.get('TN_PARTS', '1')))
        for o in outs_new:
            o.zero_()
        new()
        torch.cuda.synchronize()
        refs = [gact[:, :ff].float().t() @ xin[:, :d].float(), h2[:, :d].float().t() @ dpre[:, :ff].float(),
                att[:, :d].float().t() @ xmid[:, :d].float(), h1[:, :d].float().t() @ dqkv[:, :3 * d].float()]
        e = max(((o - r).norm() / r.norm()).item() for o, r in zip(outs_new, refs))  # all four gradients of the group
        line += " | loop%d %6.1f us %4.0f TF err %.1e" % (cfg, time_us(new), flops / time_us(new) / 1e6, e)
    lib.fact_debug_gemm_tn_cfg(0)
    print("wgrad layer K%5d: round-1 %6.1f us %4.0f TF%s" % (K, t_old, flops / t_old / 1e6, line), flush=True)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("all", "nt"):
        M = 5760
        nt_case(M, 3072, 800, L.EPI_BIAS_GELU, [1, 10, 11, 14], "FFN1+gelu")
        nt_case(M, 3072, 800, L.EPI_GELU_BWD, [1, 10, 11, 14], "dgrad gelu'")
        nt_case(M, 2400, 800, L.EPI_BF16, [1, 10, 11, 14], "QKV (plain)")
        nt_case(M, 800, 800, L.EPI_F32_BIAS_RESID, [1, 112, 14], "out-proj+resid")
        nt_case(M, 800, 3072, L.EPI_BF16, [1, 112, 12, 14], "dgrad FFN1")
        nt_case(M, 800, 3072, L.EPI_F32_BIAS_RESID, [1, 112, 12], "FFN2+resid")
        nt_case(M, 800, 800, L.EPI_F32_BIAS_RESID, [1, 112, 12], "out-proj+resid")
        nt_case(M, 800, 3072, L.EPI_BF16, [1, 112, 12], "dgrad FFN1")
        nt_case(M, 800, 2400, L.EPI_BF16, [1, 112, 12], "dgrad QKV")
        nt_case(M, 800, 800, L.EPI_BF16, [1, 112, 12], "dgrad out-proj")
        for Me in (3840, 1920):
            nt_case(Me, 800, 3072, L.EPI_F32_BIAS_RESID, [1, 112, 12], "enc FFN2")
            nt_case(Me, 800, 800, L.EPI_F32_BIAS_RESID, [1, 112, 12], "enc out-proj")
            nt_case(Me, 3072, 800, L.EPI_BIAS_GELU, [1, 10, 11, 14], "enc FFN1")
        nt_case(8192, 8192, 8192, L.EPI_BF16, [1, 11, 19], "8192^3")
    if what == "m32":  # round 6: 32x32x16 MFMA tiles (v22 = 256x256, v23 = 384x192) vs the 16x16x32 tiles on 64-deep slots (v19 / v18)
        M = 5760
        for rep in range(2):
            nt_case(8192, 8192, 8192, L.EPI_BF16, [19, 22, 23], "8192^3")
            nt_case(M, 3072, 3072, L.EPI_BF16, [18, 19, 22, 23], "long K")
            nt_case(M, 3072, 800, L.EPI_BF16, [18, 22, 23], "plain N3072")
            nt_case(M, 3072, 800, L.EPI_BIAS_GELU, [18, 22, 23], "FFN1+gelu")
            nt_case(M, 3072, 800, L.EPI_GELU_BWD, [18, 22, 23], "dgrad gelu'")
            nt_case(M, 2400, 800, L.EPI_BF16, [19, 18, 22, 23], "QKV (plain)")
        for Me in (3840, 1920):
            nt_case(Me, 3072, 800, L.EPI_BIAS_GELU, [0, 22, 23], "enc FFN1")
            nt_case(Me, 2400, 800, L.EPI_BF16, [0, 22, 23], "enc QKV")
    if what == "m32pmc":  # few launches for rocprofv3 --pmc: where does the 32x32x16 loop lose?
        globals()["ITERS"] = 3
        nt_case(8192, 8192, 8192, L.EPI_BF16, [19, 22, 23], "8192^3")
        nt_case(5760, 3072, 3072, L.EPI_BF16, [18, 23], "long K")
    if what == "t192":  # whole-K N = 800 dgrads: 256x160 (v117: 64-deep, no split-K) vs 192x160 (v20), alone on the chip
        for rep in range(2):
            for Me in (5760, 3840, 1920):
                nt_case(Me, 800, 3072, L.EPI_BF16, [117, 20], "dgrad FFN1")
                nt_case(Me, 800, 2400, L.EPI_BF16, [117, 20], "dgrad QKV")
    if what == "small":
        for Me in (5760, 3840, 1920):
            nt_case(Me, 800, 800, L.EPI_BF16, [1, 112, 14], "N800 K800 bf16")
            nt_case(Me, 800, 800, L.EPI_F32_BIAS_RESID, [1, 112, 14], "N800 K800 resid")
        for Me in (3840, 1920):
            nt_case(Me, 2400, 800, L.EPI_BF16, [1, 10, 11, 14], "enc QKV")
            nt_case(Me, 3072, 800, L.EPI_GELU_BWD, [1, 10, 11, 14], "enc gelu'")
            nt_case(Me, 800, 3072, L.EPI_BF16, [1, 12, 14], "enc dFFN1")
            nt_case(Me, 800, 2400, L.EPI_BF16, [1, 12, 14], "enc dQKV")
    if what == "t128":  # short-K N = 800 GEMMs: 256x128 x 2 per CU (v14, shipped) vs 128x160 x 2 per CU (v21, round-5 probe)
        for M in (5760, 3840, 1920):
            nt_case(M, 800, 800, L.EPI_F32_BIAS_RESID, [14, 21, 14, 21], "out-proj+resid")
            nt_case(M, 800, 800, L.EPI_BF16, [14, 21, 14, 21], "N800 K800 bf16")
    if what == "t128long":  # long-K N = 800 GEMMs: 256x160 with the (symmetric) in-kernel split-K (v17) vs 128x160 whole-K, 2 per CU (v21)
        for M in (5760, 3840):
            nt_case(M, 800, 3072, L.EPI_F32_BIAS_RESID, [17, 21, 17, 21], "FFN2+resid")
            nt_case(M, 800, 3072, L.EPI_BF16, [117, 21, 117, 21], "dgrad FFN1 (whole K)")
            nt_case(M, 800, 2400, L.EPI_BF16, [117, 21], "dgrad QKV (whole K)")
    if what == "k64":  # 256x160 tiles: 32-deep (v12, v112 = without split-K) vs 64-deep ring slots (v17, v117 = without split-K)
        M = 5760
        for rep in range(2):
            nt_case(M, 800, 3072, L.EPI_BF16, [112, 117, 12, 17], "dgrad FFN1")
            nt_case(M, 800, 3072, L.EPI_F32_BIAS_RESID, [112, 117, 12, 17], "FFN2+resid")
            nt_case(M, 800, 2400, L.EPI_BF16, [112, 117, 12, 17], "dgrad QKV")
            nt_case(M, 800, 800, L.EPI_BF16, [112, 117, 14], "N800 K800")
        for Me in (3840, 1920):
            nt_case(Me, 800, 3072, L.EPI_BF16, [112, 117, 12, 17], "enc dFFN1")
        for rep in range(2):
            nt_case(M, 3072, 800, L.EPI_BF16, [10, 18], "plain 288x256")
            nt_case(M, 3072, 800, L.EPI_BIAS_GELU, [10, 18], "FFN1+gelu")
            nt_case(M, 3072, 800, L.EPI_GELU_BWD, [10, 18], "dgrad gelu'")
            nt_case(M, 2400, 800, L.EPI_BF16, [11, 19, 10, 18], "QKV (plain)")
            nt_case(M, 3072, 3072, L.EPI_BF16, [10, 18], "long K")
    if what == "k64small":  # short-K N = 800 GEMMs: 256x128 x 2 per CU (v14) vs 256x160 on 64-deep slots (v117)
        for rep in range(2):
            for Me in (5760, 3840, 1920):
                nt_case(Me, 800, 800, L.EPI_F32_BIAS_RESID, [14, 117, 112], "out-proj+resid")
                nt_case(Me, 800, 800, L.EPI_BF16, [14, 117, 112], "N800 K800 bf16")
            nt_case(3840, 2400, 800, L.EPI_BF16, [14, 11, 19], "enc QKV")
            nt_case(3840, 3072, 800, L.EPI_BIAS_GELU, [14, 11, 19], "enc FFN1")
            nt_case(1920, 2400, 800, L.EPI_BF16, [14, 11, 19], "enc QKV")
            nt_case(1920, 3072, 800, L.EPI_BIAS_GELU, [14, 11, 19], "enc FFN1")
    if what == "wide160":  # the wide (N = 2400 / 3072) short-K GEMMs on 256x160 tiles (2 rounds of small tiles vs 1 round of big ones)
        for rep in range(2):
            for Me in (5760, 3840):
                nt_case(Me, 3072, 800, L.EPI_BF16, [10, 18, 112, 117], "plain N3072")
                nt_case(Me, 3072, 800, L.EPI_BIAS_GELU, [10, 18, 112, 117], "FFN1+gelu")
                nt_case(Me, 3072, 800, L.EPI_GELU_BWD, [10, 18, 112, 117], "dgrad gelu'")
                nt_case(Me, 2400, 800, L.EPI_BF16, [11, 19, 112, 117], "QKV (plain)")
    if what == "ar32":  # the AR sampler's forward GEMMs at 32 sequences (M = 11520)
        M = 11520
        for rep in range(2):
            nt_case(M, 800, 800, L.EPI_F32_BIAS_RESID, [0, 14, 117, 112, 1], "out-proj+resid")
            nt_case(M, 800, 3072, L.EPI_F32_BIAS_RESID, [0, 117, 17, 14], "FFN2+resid")
            nt_case(M, 2400, 800, L.EPI_BF16, [0, 18, 19, 117], "QKV (plain)")
            nt_case(M, 3072, 800, L.EPI_BIAS_GELU, [0, 18, 19, 117], "FFN1+gelu")
    if what == "enc32":  # encoder GEMMs of the AR sampler at 32 sequences (audio stack: M = 7680)
        for rep in range(2):
            nt_case(7680, 2400, 800, L.EPI_BF16, [0, 14, 11, 19, 1], "audio QKV")
            nt_case(7680, 3072, 800, L.EPI_BIAS_GELU, [0, 14, 18, 1], "audio FFN1")
            nt_case(7680, 800, 3072, L.EPI_F32_BIAS_RESID, [0, 17, 117], "audio FFN2")
            nt_case(7680, 800, 800, L.EPI_F32_BIAS_RESID, [0, 14, 117], "audio out-proj")
    if what == "pmcreq":  # L2 request counts of the plain FFN1-shaped GEMM: 128x128 (64-deep stages) vs 288x256 (32-deep stages)
        globals()["ITERS"] = 2
        nt_case(5760, 3072, 800, L.EPI_BF16, [1], "plain 128x128")
        nt_case(5760, 3072, 800, L.EPI_BF16, [10], "plain 288x256")
    if what == "pmcloop":  # long-K plain GEMMs: the main loop dominates (PMC breakdown of the loop itself)
        globals()["ITERS"] = 3
        nt_case(5760, 3072, 3072, L.EPI_BF16, [10], "long K 288x256")
        nt_case(5632, 3072, 3072, L.EPI_BF16, [11], "long K 256x256")
        nt_case(5760, 800, 3072, L.EPI_BF16, [112], "long K 256x160")
        wgrad_layer(5760)
    if what == "pmc":  # few launches of the kernels of interest (for rocprofv3 --pmc passes)
        ITERS = 4
        globals()["ITERS"] = 4
        M = 5760
        nt_case(M, 3072, 800, L.EPI_BIAS_GELU, [10], "FFN1+gelu")
        nt_case(M, 800, 3072, L.EPI_BF16, [1, 12], "dgrad FFN1")
        wgrad_layer(5760)
    if what in ("all", "tn"):
        for K in (5760, 3840, 1920):
            wgrad_layer(K)
