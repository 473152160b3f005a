"""Per-wave cycle stamps of the LDS-resident forward attention kernel (fact_debug_attn_timestamps)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mint_amd import _lib as L
lib = L.lib()
B, H, n, dh = 16, 10, 360, 80
hid = H * dh
g = torch.Generator(device="cuda").manual_seed(5)
qkv = (torch.randn(B * n, 3 * hid, device="cuda", generator=g) * 3).to(torch.bfloat16)
out = torch.empty(B * n, hid, device="cuda", dtype=torch.bfloat16)
scratch = torch.empty(lib.fact_op_attention_scratch(B, H, n, dh), device="cuda", dtype=torch.uint8)
nw = 12
ts = torch.zeros(B * H * nw * 8, device="cuda", dtype=torch.int64)
lib.fact_debug_attn_variant(1)
def run():
    L.check(lib.fact_op_attention(L.ptr(qkv), B, H, n, dh, hid ** -0.5, L.ptr(out), None, None, L.ptr(scratch), L.cur_stream()))
for _ in range(3): run()
lib.fact_debug_attn_timestamps(L.ptr(ts))
run(); torch.cuda.synchronize()
lib.fact_debug_attn_timestamps(None)
t = ts.view(B * H, nw, 8).cpu().double()
t0 = t[:, :, 0].min()
names = ["start", "loads issued", "own loads landed", "barrier passed", "tile 1", "tile 6", "loop done", "stores retired"]
print("stamps relative to the first wave start of the launch (cycles): min / median / max over all %d waves" % (B * H * nw))
for i, nm in enumerate(names):
    x = (t[:, :, i] - t0).flatten()
    print("  %-18s %9.0f %9.0f %9.0f" % (nm, x.min(), x.median(), x.max()))
d = t[:, :, 1:] - t[:, :, :-1]
print("per-wave phase durations (cycles), median / max:")
for i in range(7):
    x = d[:, :, i].flatten()
    print("  %-18s -> %-18s %9.0f %9.0f" % (names[i], names[i + 1], x.median(), x.max()))
wg = (t[:, :, 7].max(dim=1).values - t[:, :, 0].min(dim=1).values)
print("workgroup residency: median %.0f max %.0f cycles; launch span %.0f cycles" % (wg.median(), wg.max(), (t[:, :, 7].max() - t0)))
