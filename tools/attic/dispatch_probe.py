"""Does a kernel running on ANOTHER stream slow down kernel-to-kernel dispatch on this stream?  Stream B launches a chain
of `n` tiny dependent kernels (torch add_ on 1 KiB); stream A meanwhile runs nothing / one long spin kernel holding `ch`
CUs.  Reports microseconds per tiny kernel on B."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mint_amd import _lib as L

lib = L.lib()
torch.cuda.set_device(0)
x = torch.zeros(256, device="cuda")
big = torch.zeros(64 << 20, device="cuda", dtype=torch.uint8)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
n = 400


def chain(kind):
    with torch.cuda.stream(sb):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            if kind == "tiny":
                x.add_(1.0)
            else:
                big.add_(1)   # ~64 MiB read+write: a ~25 us bandwidth kernel
        e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for kind in ("tiny", "25us"):
    res = {}
    for rnd in range(3):
        for ch in (0, 1, 8, 64):
            torch.cuda.synchronize()
            if ch:
                L.check(lib.fact_debug_cu_hog(ch, 40000, C.c_void_p(sa.cuda_stream)))
            res.setdefault(ch, []).append(chain(kind))
            torch.cuda.synchronize()
    print("dispatch_probe %s kernels, queues=%s: " % (kind, os.environ.get("GPU_MAX_HW_QUEUES", "dflt"))
          + "  ".join("hog%d %s us" % (k, "/".join("%.2f" % v for v in vs)) for k, vs in res.items()), flush=True)
