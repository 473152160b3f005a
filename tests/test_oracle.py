"""CPU tests of the oracle (oracle/fact_oracle.py): shapes pinned by the reference's own tests,
an independent cross-check against torch.nn compositions, the committed golden fixtures, and the
train-step semantics (loss/R, summed grads, Keras Adam)."""
import math
import os

import numpy as np
import pytest
import torch

from oracle import fact_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tiny_fact_golden.npz")


def _golden_setup():
    cfg = O.TINY_CFG
    params = O.init_params(cfg, seed=0)
    g = torch.Generator().manual_seed(7)
    for k, v in params.items():
        if k.endswith("/bias") or k.endswith("/beta"):
            v.copy_(torch.randn(v.shape, generator=g, dtype=torch.float64) * 0.05)
        elif k.endswith("/gamma"):
            v.copy_(1.0 + torch.randn(v.shape, generator=g, dtype=torch.float64) * 0.1)
    return cfg, params, O.synthetic_batch(cfg, 2, 8, seed=11)


def test_param_count_matches_reference_model():
    # 120 406 977 trainable parameters in 184 tensors for fact_v5_deeper_t10_cm12
    assert O.num_params(O.FACT_V5_CFG) == 120406977
    assert len(O.param_shapes(O.FACT_V5_CFG)) == 184


def test_reference_shape_tests():
    # mint/core/base_models_test.py:22-29: Transformer(hidden 20, 10 heads) on ones (4,128,20)
    d, heads, ff, L = 20, 10, 64, 2
    gen = torch.Generator().manual_seed(0)
    params = {}
    for l in range(L):
        n = O.layer_names("t", l)
        shapes = {"ln1_g": (d,), "ln1_b": (d,), "wqkv": (d, 3 * d), "wo": (d, d), "bo": (d,), "ln2_g": (d,),
                  "ln2_b": (d,), "w1": (d, ff), "b1": (ff,), "w2": (ff, d), "b2": (d,)}
        for k, s in shapes.items():
            params[n[k]] = torch.randn(s, generator=gen, dtype=torch.float64) * 0.1
    out = O.transformer(torch.ones(4, 128, d, dtype=torch.float64), params, "t", L, heads)
    assert out.shape == (4, 128, d)
    # mint/core/fact_model_test.py:47-54 at the tiny config: (2, n_m + n_a, 225) from all-ones inputs
    cfg = O.TINY_CFG
    p = O.init_params(cfg)
    y = O.fact_forward(p, cfg, torch.ones(2, 32, 225, dtype=torch.float64), torch.ones(2, 64, 35, dtype=torch.float64))
    assert y.shape == (2, 96, 225) and torch.isfinite(y).all()


def test_oracle_matches_independent_torch_modules():
    """Same block built from torch.nn / F primitives (LayerNorm, F.gelu(tanh), SDPA with explicit
    scale, Linear with transposed Keras kernels) must agree with the oracle to fp64 round-off."""
    torch.manual_seed(0)
    d, heads, ff, n, b = 32, 4, 64, 10, 3
    x = torch.randn(b, n, d, dtype=torch.float64)
    names = O.layer_names("s", 0)
    shapes = {"ln1_g": (d,), "ln1_b": (d,), "wqkv": (d, 3 * d), "wo": (d, d), "bo": (d,), "ln2_g": (d,),
              "ln2_b": (d,), "w1": (d, ff), "b1": (ff,), "w2": (ff, d), "b2": (d,)}
    p = {names[k]: torch.randn(s, dtype=torch.float64) * 0.3 for k, s in shapes.items()}
    ref = O.transformer(x, p, "s", 1, heads)

    F = torch.nn.functional
    h = F.layer_norm(x, (d,), p[names["ln1_g"]], p[names["ln1_b"]], eps=1e-5)
    qkv = F.linear(h, p[names["wqkv"]].t())
    q, k, v = qkv.split(d, dim=-1)  # (qkv h d): q = first d columns, head-major inside
    sh = lambda t: t.reshape(b, n, heads, d // heads).transpose(1, 2)
    a = F.scaled_dot_product_attention(sh(q), sh(k), sh(v), scale=d ** -0.5)  # scale = dim**-0.5
    a = a.transpose(1, 2).reshape(b, n, d)
    x1 = x + F.linear(a, p[names["wo"]].t(), p[names["bo"]])
    h2 = F.layer_norm(x1, (d,), p[names["ln2_g"]], p[names["ln2_b"]], eps=1e-5)
    x2 = x1 + F.linear(F.gelu(F.linear(h2, p[names["w1"]].t(), p[names["b1"]]), approximate="tanh"),
                       p[names["w2"]].t(), p[names["b2"]])
    assert torch.allclose(ref, x2, rtol=1e-10, atol=1e-10)
    # and the quirk matters: head-dim scaling gives a different answer
    a_wrong = F.scaled_dot_product_attention(sh(q), sh(k), sh(v))
    assert not torch.allclose(a_wrong.transpose(1, 2).reshape(b, n, d), a, atol=1e-6)


def test_golden_fixture():
    cfg, params, batch = _golden_setup()
    gold = np.load(GOLDEN)
    loss, grads, pred = O.loss_and_grads(params, cfg, batch["motion_input"], batch["audio_input"], batch["target"])
    assert abs(float(loss) - float(gold["loss"])) < 1e-12
    np.testing.assert_allclose(pred.float().numpy(), gold["pred"], rtol=1e-5, atol=1e-6)
    names = [n for n, _ in O.param_shapes(cfg)]
    np.testing.assert_allclose([float(grads[n].norm()) for n in names], gold["grad_norms"], rtol=1e-9)
    np.testing.assert_allclose([float(grads[n].sum()) for n in names], gold["grad_sums"], rtol=1e-7, atol=1e-12)
    m = {k: torch.zeros_like(v) for k, v in params.items()}
    v = {k: torch.zeros_like(x) for k, x in params.items()}
    p1, _, _ = O.adam_update(params, grads, m, v, 0, 1e-3)
    np.testing.assert_allclose([float((p1[n] - params[n]).norm()) for n in names], gold["adam_delta_norms"],
                               rtol=1e-9)
    ar = O.infer_auto_regressive(params, cfg, batch["motion_input"],
                                 torch.cat([batch["audio_input"], batch["audio_input"][:, :3]], dim=1), steps=4)
    np.testing.assert_allclose(ar.float().numpy(), gold["ar"], rtol=1e-5, atol=1e-6)


def test_loss_is_mean_over_first_target_frames_only():
    cfg, params, batch = _golden_setup()
    pred = O.fact_forward(params, cfg, batch["motion_input"], batch["audio_input"])
    t = batch["target"]
    expect = ((t - pred[:, :8]) ** 2).sum() / (2 * 8 * 225)
    assert torch.allclose(O.motion_loss(t, pred), expect)
    # rows >= 8 of the prediction receive zero gradient
    _, grads, _ = O.loss_and_grads(params, cfg, batch["motion_input"], batch["audio_input"], t)
    assert float(grads["cross_modal_layer/output/bias"].abs().sum()) > 0


def test_data_parallel_semantics_sum_of_scaled_grads():
    """single_task_trainer.py:157-158,186-187: each replica differentiates loss/R and gradients are
    summed -> equals the gradient of the global-batch mean loss."""
    cfg, params, _ = _golden_setup()
    full = O.synthetic_batch(cfg, 4, 8, seed=5)
    _, g_full, _ = O.loss_and_grads(params, cfg, full["motion_input"], full["audio_input"], full["target"])
    acc = None
    for r in range(2):
        sl = slice(2 * r, 2 * r + 2)
        _, g, _ = O.loss_and_grads(params, cfg, full["motion_input"][sl], full["audio_input"][sl],
                                   full["target"][sl], num_replicas=2)
        acc = g if acc is None else {k: acc[k] + g[k] for k in g}
    for k in g_full:
        assert torch.allclose(acc[k], g_full[k], rtol=1e-9, atol=1e-12), k


def test_keras_adam_first_step_is_lr_sized():
    p = {"w": torch.tensor([1.0, -2.0], dtype=torch.float64)}
    g = {"w": torch.tensor([0.5, -0.25], dtype=torch.float64)}
    z = {"w": torch.zeros(2, dtype=torch.float64)}
    p1, m1, v1 = O.adam_update(p, g, z, z, 0, 1e-3)
    # t=1: m_hat/sqrt(v_hat) = sign(g); epsilon sits outside the bias correction
    lr_t = 1e-3 * math.sqrt(1 - 0.999) / (1 - 0.9)
    exp = p["w"] - lr_t * (0.1 * g["w"]) / (torch.sqrt(0.001 * g["w"] ** 2) + 1e-7)
    assert torch.allclose(p1["w"], exp)
    # clip_by_global_norm
    p2, _, _ = O.adam_update(p, g, z, z, 0, 1e-3, clip_norm=0.1)
    assert torch.isfinite(p2["w"]).all()


def test_autoregressive_stops_when_audio_runs_out():
    cfg, params, batch = _golden_setup()
    out = O.infer_auto_regressive(params, cfg, batch["motion_input"], batch["audio_input"], steps=5)
    assert out.shape == (2, 1, 225)  # exactly one full 64-frame window
    short = O.infer_auto_regressive(params, cfg, batch["motion_input"], batch["audio_input"][:, :10], steps=5)
    assert short.shape == (2, 0, 225)


def test_keras_adam_algebra_against_torch_optim_adam():
    """The oracle's Keras-Adam restatement against an INDEPENDENT implementation.  Keras applies epsilon to the
    un-corrected sqrt(v) (`lr_t = lr*sqrt(1-b2^t)/(1-b1^t); p -= lr_t*m/(sqrt(v)+eps)`, the formulation just before
    section 2.1 of Kingma & Ba), torch.optim.Adam to the bias-corrected one (`p -= lr/(1-b1^t) * m/(sqrt(v)/sqrt(1-b2^t)+eps')`);
    the two coincide exactly when eps' = eps/sqrt(1-b2^t), which is set per step here.  Checks the update algebra, the moment
    recursions and the t = step+1 convention over 6 steps in fp64 (what Keras itself does cannot be run here)."""
    import math
    g = torch.Generator().manual_seed(11)
    p0 = {"a": torch.randn(7, 5, generator=g, dtype=torch.float64), "b": torch.randn(9, generator=g, dtype=torch.float64)}
    grads = [{k: torch.randn(v.shape, generator=g, dtype=torch.float64) * (10.0 ** (-i)) for k, v in p0.items()}
             for i in range(6)]
    lr, b1, b2, eps = 3e-3, 0.9, 0.999, 1e-7
    p = {k: v.clone() for k, v in p0.items()}
    m = {k: torch.zeros_like(v) for k, v in p0.items()}
    v = {k: torch.zeros_like(x) for k, x in p0.items()}
    tp = {k: torch.nn.Parameter(x.clone()) for k, x in p0.items()}
    opt = torch.optim.Adam(list(tp.values()), lr=lr, betas=(b1, b2), eps=eps)
    for step, gr in enumerate(grads):
        p, m, v = O.adam_update(p, gr, m, v, step, lr, b1, b2, eps)
        for k in tp:
            tp[k].grad = gr[k].clone()
        opt.param_groups[0]["eps"] = eps / math.sqrt(1.0 - b2 ** (step + 1))
        opt.step()
        for k in tp:
            assert torch.allclose(p[k], tp[k].detach(), rtol=1e-12, atol=1e-14), (step, k)
            st = opt.state[tp[k]]
            assert torch.allclose(m[k], st["exp_avg"], rtol=1e-12, atol=0) and torch.allclose(v[k], st["exp_avg_sq"], rtol=1e-12, atol=0)
