"""Diagnostic: optimizer-in-wgrad vs bucket path, per tensor, after 1 and 2 steps, plus a same-path control."""
import os, sys
os.environ.setdefault("FACT_DEBUG_ABI", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))), "tests"))
import torch
import test_gpu_model as T
from mint_amd import _lib as L

L.lib().fact_debug_attn_variant(1)
cfg = T.GROUPED_CFG
for sr in (0, 1):
    for steps in (1, 2):
        runs = {name: T._train_state(cfg, 16, 8, steps, iw, sr_rows=sr) for name, iw in (("ref", 0), ("ref2", 0), ("got", 1))}
        for other in ("ref2", "got"):
            a, b = runs[other], runs["ref"]
            print("== sr=%d steps=%d  %s vs ref: losses %s | %s" % (sr, steps, other, a[0], b[0]))
            for k in ("params", "adam_m", "adam_v"):
                if k not in a[2]:
                    continue
                x, y = a[2][k], b[2][k]
                bad = ~torch.isclose(x, y, rtol=2e-5, atol=2e-7)
                if not bool(bad.any()):
                    print("   %-8s equal" % k)
                    continue
                rows = []
                for (n, o, r, c, _k) in a[4]:
                    sl = slice(o, o + r * c)
                    nb = int(bad[sl].sum())
                    if nb:
                        rows.append("%s %d/%d max %.3g" % (n.replace("cross_modal_layer/transformer/", "X/"), nb, r * c,
                                                           float((x[sl] - y[sl]).abs().max())))
                print("   %-8s %d differ: %s" % (k, int(bad.sum()), "; ".join(rows[:12])))
