"""What the vendor BLAS reaches at the FACT GEMM shapes (reference point for our kernels only)."""
import torch, sys
dev = "cuda"
SH = [(5760, 3072, 800), (5760, 800, 3072), (5760, 2400, 800), (5760, 800, 800), (8192, 8192, 8192),
      ("tn", 800, 3072, 5760), ("tn", 3072, 800, 5760), ("tn", 800, 800, 5760), ("tn", 800, 2400, 5760)]
for s in SH:
    if s[0] == "tn":
        _, Mo, No, K = s
        A = torch.randn(K, Mo, device=dev).to(torch.bfloat16)
        B = torch.randn(K, No, device=dev).to(torch.bfloat16)
        f = lambda: torch.matmul(A.t(), B)
        fl = 2.0 * Mo * No * K
    else:
        M, N, K = s
        A = torch.randn(M, K, device=dev).to(torch.bfloat16)
        B = torch.randn(N, K, device=dev).to(torch.bfloat16)
        f = lambda: torch.matmul(A, B.t())
        fl = 2.0 * M * N * K
    for _ in range(5):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        f()
    e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) / 30 * 1e3
    print(s, "%.1f us %.0f TF" % (us, fl / us / 1e6), flush=True)
