"""Stand-alone timing + equality check of the fused Adam + shadow kernels (engine option adam_variant; FACT_ADAM_TW)."""
import os as _os; _os.environ.setdefault("FACT_DEBUG_ABI", "1")  # these tools drive the test / bench surface (mint_amd/_lib.py)
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mint_amd import configs, model_builder


def main():
    torch.cuda.set_device(0)
    pipe = configs.fact_v5_deeper_t10_cm12()
    model = model_builder.build(pipe.multi_modal_model, True)
    model.build(16, 225, 35)
    n = model._arena["params"].numel()
    g = torch.Generator(device="cuda").manual_seed(0)
    grads = torch.randn(n, device="cuda", generator=g) * 1e-3
    p0 = model._arena["params"].clone()
    ref = None
    for variant in [int(x) for x in os.environ.get("VARIANTS", "0,1,2").split(",")]:
        model.debug_option("adam_variant", variant)
        for k in ("adam_m", "adam_v"):
            model._arena[k].zero_()
        model._arena["params"].copy_(p0)
        model._arena["grads"].copy_(grads)
        from mint_amd import _lib as L
        L.check(L.lib().fact_set_step(model._h, 0))   # the bias correction depends on the step counter
        model.apply_adam(1e-3)
        torch.cuda.synchronize()
        state = (model._arena["params"].clone(), model._arena["adam_m"].clone(), model._arena["adam_v"].clone())
        gz = float(model._arena["grads"].abs().max())
        if ref is None:
            ref = state
        same = all(torch.equal(a, b) for a, b in zip(state, ref))
        if not same:
            for nm, a, b in zip(("p", "m", "v"), state, ref):
                d = (a - b).abs()
                i = int(d.argmax())
                print("   %s: max abs diff %.3e at %d (%.9e vs %.9e), mismatching %d of %d" % (
                    nm, float(d.max()), i, float(a[i]), float(b[i]), int((d > 0).sum()), a.numel()))
        # time
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            model.apply_adam(1e-3)
        e0.record()
        for _ in range(20):
            model.apply_adam(1e-3)
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print("adam_variant %d tw %s: %.3f ms  %.0f GB/s (36 B/param)  identical_to_first %s  max|g| after %.1e"
              % (variant, os.environ.get("FACT_ADAM_TW", "64"), ms, n * 36 / ms / 1e6, same, gz), flush=True)
    model.debug_option("adam_variant", 0)


if __name__ == "__main__":
    main()
