"""Run-to-run determinism of the attention op per kernel family: the same inputs twice (fresh scratch filled with
different garbage each time) must give bit-identical out / dqkv."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mint_amd import _lib as L

lib = L.lib()
DEV = "cuda"


def run(B, H, n, dh, variant, fill):
    hid = H * dh
    g = torch.Generator(device=DEV).manual_seed(5)
    qkv = (torch.randn(B * n, 3 * hid, device=DEV, generator=g) * 3.0).to(torch.bfloat16)
    dout = torch.randn(B * n, hid, device=DEV, generator=g).to(torch.bfloat16)
    out = torch.full((B * n, hid), float("nan"), device=DEV, dtype=torch.bfloat16)
    dqkv = torch.full((B * n, 3 * hid), float("nan"), device=DEV, dtype=torch.bfloat16)
    nbytes = lib.fact_op_attention_scratch(B, H, n, dh)
    scratch = torch.full((nbytes,), fill, device=DEV, dtype=torch.uint8)
    lib.fact_debug_attn_variant(variant)
    L.check(lib.fact_op_attention(L.ptr(qkv), B, H, n, dh, hid ** -0.5, L.ptr(out), L.ptr(dout), L.ptr(dqkv),
                                  L.ptr(scratch), L.cur_stream()))
    torch.cuda.synchronize()
    return out.clone(), dqkv.clone()


for (B, H, n, dh) in [(4, 4, 32, 32), (4, 4, 64, 32), (4, 4, 96, 32), (2, 10, 360, 80), (2, 10, 120, 80), (16, 10, 360, 80)]:
    for variant in (1, 2, 3, 4):
        a = run(B, H, n, dh, variant, 0)
        res = []
        for fill in (0, 0x7F, 0xFF, 0x3C):
            b = run(B, H, n, dh, variant, fill)
            res.append((torch.equal(a[0], b[0]), torch.equal(a[1], b[1]),
                        float((a[0].float() - b[0].float()).abs().max()), float((a[1].float() - b[1].float()).abs().max())))
        print("B%d H%d n%d dh%d variant %d: " % (B, H, n, dh, variant) + "  ".join("out %s dqkv %s (%.2e %.2e)" % r for r in res), flush=True)
lib.fact_debug_attn_variant(3)
