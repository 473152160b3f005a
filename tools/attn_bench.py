"""Attention micro-benchmark: runs the attention op (forward + backward) at FACT's shapes for each kernel family
and prints the relative error vs torch fp32; per-kernel times come from rocprofv3 around this script
(tools/attn_prof.sh) - HIP-event wall time of the whole op (incl. the head-scatter GEMM of the op) is printed too."""
import os as _os; _os.environ.setdefault("FACT_DEBUG_ABI", "1")  # these tools drive the test / bench surface (mint_amd/_lib.py)
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mint_amd import _lib as L

lib = L.lib()
dev = "cuda"
ITERS = int(os.environ.get("ITERS", "20"))
VARIANTS = [int(v) for v in os.environ.get("VARIANTS", "5,2").split(",")]  # 5 = default pairing, 2 = streaming everywhere


def ref(qkv, B, H, n, dh, scale, dout):
    hid = H * dh
    x = qkv.float().clone().requires_grad_(True)
    t = x.view(B, n, 3, H, dh).permute(2, 0, 3, 1, 4)
    q, k, v = t[0], t[1], t[2]
    a = torch.softmax(torch.einsum("bhid,bhjd->bhij", q, k) * scale, dim=-1)
    out = torch.einsum("bhij,bhjd->bhid", a, v).permute(0, 2, 1, 3).reshape(B * n, hid)
    out.backward(dout.float())
    return out.detach(), x.grad


def case(B, H, n, dh, std=3.0):
    hid = H * dh
    scale = hid ** -0.5
    g = torch.Generator(device=dev).manual_seed(5)
    qkv = (torch.randn(B * n, 3 * hid, device=dev, generator=g) * std).to(torch.bfloat16)
    dout = torch.randn(B * n, hid, device=dev, generator=g).to(torch.bfloat16)
    out = torch.empty(B * n, hid, device=dev, dtype=torch.bfloat16)
    dqkv = torch.empty(B * n, 3 * hid, device=dev, dtype=torch.bfloat16)
    scratch = torch.empty(lib.fact_op_attention_scratch(B, H, n, dh), device=dev, dtype=torch.uint8)
    r_out, r_dqkv = ref(qkv[: 2 * n], 2, H, n, dh, scale, dout[: 2 * n])
    for v in VARIANTS:
        lib.fact_debug_attn_variant(v)

        def run():
            L.check(lib.fact_op_attention(L.ptr(qkv), B, H, n, dh, scale, L.ptr(out), L.ptr(dout), L.ptr(dqkv),
                                          L.ptr(scratch), L.cur_stream()))
        out.fill_(float("nan")); dqkv.fill_(float("nan"))
        run()
        torch.cuda.synchronize()
        e_out = ((out[: 2 * n].float() - r_out).norm() / r_out.norm()).item()
        errs = [((dqkv[: 2 * n, w * hid:(w + 1) * hid].float() - r_dqkv[:, w * hid:(w + 1) * hid]).norm()
                 / r_dqkv[:, w * hid:(w + 1) * hid].norm()).item() for w in range(3)]
        fin = bool(torch.isfinite(out.float()).all() and torch.isfinite(dqkv.float()).all())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(ITERS):
            run()
        e1.record(); e1.synchronize()
        print("B%d H%d n%d dh%d variant %d: op %.1f us  rel err out %.2e dq %.2e dk %.2e dv %.2e finite %s" % (
            B, H, n, dh, v, e0.elapsed_time(e1) / ITERS * 1e3, e_out, errs[0], errs[1], errs[2], fin), flush=True)
    lib.fact_debug_attn_variant(5)


if __name__ == "__main__":
    shapes = os.environ.get("SHAPES", "16x10x360x80,16x10x240x80,16x10x120x80")
    for sh in shapes.split(","):
        B, H, n, dh = [int(x) for x in sh.split("x")]
        case(B, H, n, dh)
