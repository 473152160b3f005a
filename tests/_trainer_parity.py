"""The EXACT path bench.py times, at the configuration it times, against the oracle (round-5 review item 3):
fact_v5_deeper_t10_cm12, batch 16, `SingleTaskTrainer` with its defaults - the optimizer inside backward
(fact_adam_begin), `grad_overwrite`, the supervised-rows shortcut of the last cross-modal layer - for three optimizer steps
at lr 1e-3, against `oracle.fact_oracle.train_step` (fp32 PyTorch-CPU restatement of
mint/ctl/single_task_trainer.py:141-196 + Keras Adam, trainer.py:150) from the same weights on the same batch.

Shared by tests/test_gpu_model.py (the test / bench build, in process) and tests/test_gpu_production_lib.py (a fresh
interpreter bound to libfact_hip.so): `python tests/_trainer_parity.py` exits non-zero on the first violated bound.

Stated tolerances (bf16 MFMA operands and fp32 accumulation into fp32 Adam, vs the fp32 oracle):
  per-step loss (per-frame pose MSE)    rel. diff <= 1e-2
  first moment m after 3 steps          rel. Frobenius <= 5e-2 per tensor
  second moment v after 3 steps         rel. Frobenius <= 1e-1 per tensor (squares double the relative gradient error)
  parameter update p3 - p0              cosine >= 0.98 per tensor, all 184 tensors
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _cos(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a @ b) / (a.norm() * b.norm() + 1e-300))


def _rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-300))


def run(steps=3, lr=1e-3, B=16, verbose=True):
    from mint_amd import configs, model_builder
    from mint_amd import _lib as L
    from mint_amd.trainer import Adam, SingleTaskTrainer
    from oracle import fact_oracle as O

    cfg = O.FACT_V5_CFG
    model = model_builder.build(configs.fact_v5_deeper_t10_cm12().multi_modal_model, True)
    batch = O.synthetic_batch(cfg, B, 20, seed=0, dtype=torch.float32)
    gb = {k: v.float().cuda() for k, v in batch.items()}
    model.build(B, 225, 35)
    # non-trivial biases / LayerNorm affine so every optimizer path moves
    g = torch.Generator().manual_seed(11)
    for name, v in zip(model.variable_names, model.trainable_variables):
        if name.endswith("/bias") or name.endswith("/beta"):
            v.copy_(torch.randn(v.shape, generator=g) * 0.05)
        elif name.endswith("/gamma"):
            v.copy_(1.0 + torch.randn(v.shape, generator=g) * 0.1)
    model.sync_weights()
    names = model.variable_names
    assert len(names) == 184
    params = {n: v.detach().cpu().float().clone() for n, v in zip(names, model.trainable_variables)}
    p0 = {k: v.clone() for k, v in params.items()}
    m = {k: torch.zeros_like(v) for k, v in params.items()}
    v = {k: torch.zeros_like(x) for k, x in params.items()}
    trainer = SingleTaskTrainer([gb] * steps, "target", model, optimizer=Adam(lr))  # defaults = what bench.py builds
    trainer.train_loop_begin()
    it = iter([gb] * steps)
    torch.set_num_threads(max(1, min(len(os.sched_getaffinity(0)), 64)))
    losses, ref_losses = [], []
    for step in range(steps):
        losses.append(float(trainer.train_step(it)))
        l, _, params, m, v = O.train_step(params, m, v, step, cfg, batch, lr)
        ref_losses.append(float(l))
    torch.cuda.synchronize()
    for a, b in zip(losses, ref_losses):
        assert abs(a - b) / b <= 1e-2, ("per-step loss", losses, ref_losses)
    st = model.state_dict()
    views = lambda arena: {n: arena[off:off + r * c] for (n, off, r, c, _k) in model._table}
    pm, pv, pp = views(st["adam_m"]), views(st["adam_v"]), views(st["params"])
    worst = {"m": (0.0, ""), "v": (0.0, ""), "cos": (1.0, "")}
    for n in names:
        rm, rv = _rel(pm[n], m[n]), _rel(pv[n], v[n])
        c = _cos(pp[n].double() - p0[n].double().flatten(), params[n].double() - p0[n].double())
        worst["m"] = max(worst["m"], (rm, n))
        worst["v"] = max(worst["v"], (rv, n))
        worst["cos"] = min(worst["cos"], (c, n))
        assert rm <= 5e-2, "m %s rel %.4f" % (n, rm)
        assert rv <= 1e-1, "v %s rel %.4f" % (n, rv)
        assert c >= 0.98, "update %s cos %.4f" % (n, c)
    if verbose:
        print("trainer parity ok on %s: losses %s vs oracle %s; worst m rel %.4f (%s), v rel %.4f (%s), update cos %.5f (%s)" % (
            os.path.basename(L.LIB_PATH), ["%.5f" % x for x in losses], ["%.5f" % x for x in ref_losses],
            worst["m"][0], worst["m"][1], worst["v"][0], worst["v"][1], worst["cos"][0], worst["cos"][1]))
    return {"losses": losses, "ref_losses": ref_losses, "worst": worst, "library": os.path.basename(L.LIB_PATH)}


if __name__ == "__main__":
    run()
