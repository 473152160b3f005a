"""CPU oracle for the FACT hot path — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product path (mint_amd/) never does and has no CPU fallback.

A plain PyTorch-CPU restatement (fp64 by default for parity, fp32 for the timed CPU baseline) of
the reference algorithm, one function per reference symbol, honouring the reference's quirks:

  layer_norm          mint/core/base_models.py:22-31      (eps 1e-5, biased variance)
  gelu                mint/core/base_model_util.py:94-107 (tanh approximation)
  attention           mint/core/base_models.py:60-88      (no qkv bias; (qkv h d) split; scale = dim**-0.5)
  mlp                 mint/core/base_models.py:45-57
  transformer         mint/core/base_models.py:91-110     (pre-LN residual blocks, no final LN)
  fact_forward        mint/core/fact_model.py:72-101 + base_models.py:130-202 (embed+pos, concat
                      [motion; audio], cross stack, linear head)
  motion_loss         mint/core/fact_model.py:143-148
  infer_auto_regressive  mint/core/fact_model.py:103-132
  train_step          mint/ctl/single_task_trainer.py:141-196 (loss/R, summed grads, optional
                      clip_by_global_norm) + Keras Adam (epsilon outside the bias correction)

PARITY PIN: the reference as shipped cannot run here (TensorFlow/Keras/Orbit are absent; its model tests
pin shapes only - fact_model_test.py:47-54, base_models_test.py:22-40).  What does run is the reference's
own MODEL CODE: tests/golden/make_reference_golden.py imports /root/reference/mint/core/*.py unchanged on
top of a stand-in for the few TensorFlow/Keras primitives it calls (tests/golden/ref_shim: Dense,
LayerNormalization, softmax, einsum, concat, reduce_mean ... on PyTorch-CPU float64; einops supplies
Rearrange) and records FACTModel.call / .loss / .infer_auto_regressive and the autograd gradients of that
loss.  tests/test_oracle_vs_reference.py holds this oracle to those vectors at 1e-12 (they agree to the last
bit), and re-runs the reference code live where the checkout exists.  So the composition - layer order,
pre-LN residuals, hidden**-0.5 scale, "(qkv h d)" split, tanh-GELU, [motion; audio] concat, loss slice, AR
window shift / early break - is pinned on the reference; the primitives themselves (and Keras Adam, which
the reference only instantiates) are restated from the Keras documentation and remain UNPINNED against a
real TensorFlow build.  Further pins: the reference's only numeric KAT (learning_schedules_test.py:22-40,
against mint_amd.learning_schedules) and an independent torch.nn cross-check (tests/test_oracle.py).
tests/golden/tiny_fact_golden.npz is produced BY this oracle (make_golden.py) and guards against drift.

Third-party arithmetic restated here (not under /root/reference): TensorFlow/Keras Dense,
LayerNormalization, softmax, einsum, Adam (README.md:21 `pip install tensorflow`, unpinned, TF 2.4-2.6
era) and einops Rearrange (README.md:27-28, unpinned).
"""
import math

import torch

# parameter names follow the engine's table (include/fact_hip.h FactParamDesc, Keras order)


def layer_names(prefix, l):
    b = "%s/layer_%d" % (prefix, l)
    return {
        "ln1_g": b + "/attn_norm/gamma", "ln1_b": b + "/attn_norm/beta",
        "wqkv": b + "/attn/to_qkv/kernel", "wo": b + "/attn/to_out/kernel", "bo": b + "/attn/to_out/bias",
        "ln2_g": b + "/mlp_norm/gamma", "ln2_b": b + "/mlp_norm/beta",
        "w1": b + "/mlp/dense_1/kernel", "b1": b + "/mlp/dense_1/bias",
        "w2": b + "/mlp/dense_2/kernel", "b2": b + "/mlp/dense_2/bias",
    }


def param_shapes(cfg):
    """Ordered (name, shape) list in Keras trainable_variables order."""
    out = []

    def stack(prefix, c):
        d, ff = c["hidden"], c["ff"]
        for l in range(c["layers"]):
            n = layer_names(prefix, l)
            out.extend([(n["ln1_g"], (d,)), (n["ln1_b"], (d,)), (n["wqkv"], (d, 3 * d)), (n["wo"], (d, d)),
                        (n["bo"], (d,)), (n["ln2_g"], (d,)), (n["ln2_b"], (d,)), (n["w1"], (d, ff)),
                        (n["b1"], (ff,)), (n["w2"], (ff, d)), (n["b2"], (d,))])

    stack("cross_modal_layer/transformer", cfg["cross"])
    out.append(("cross_modal_layer/output/kernel", (cfg["cross"]["hidden"], cfg["out_dim"])))
    out.append(("cross_modal_layer/output/bias", (cfg["out_dim"],)))
    for name in ("motion", "audio"):
        c = cfg[name]
        stack(name + "_transformer", c)
        out.append((name + "_pos_embedding/position_embedding", (c["seq_len"], c["hidden"])))
        out.append((name + "_linear_embedding/kernel", (c["feature_dim"], c["hidden"])))
        out.append((name + "_linear_embedding/bias", (c["hidden"],)))
    return out


def init_params(cfg, seed=0, dtype=torch.float64):
    """Reference initialisers: glorot_uniform kernels / zero biases in blocks and embeddings,
    TruncatedNormal(0.02) position tables and head kernel (base_models.py:147,176-180)."""
    gen = torch.Generator().manual_seed(seed)
    params = {}
    for name, shape in param_shapes(cfg):
        if name.endswith("/gamma"):
            t = torch.ones(shape, dtype=dtype)
        elif name.endswith("position_embedding") or name == "cross_modal_layer/output/kernel":
            t = torch.empty(shape, dtype=dtype).normal_(0, 0.02, generator=gen).clamp_(-0.04, 0.04)
        elif name.endswith("/kernel"):
            limit = math.sqrt(6.0 / (shape[0] + shape[1]))
            t = torch.empty(shape, dtype=dtype).uniform_(-limit, limit, generator=gen)
        else:
            t = torch.zeros(shape, dtype=dtype)
        params[name] = t
    return params


def layer_norm(x, gamma, beta, eps=1e-5):
    mean = x.mean(-1, keepdim=True)
    var = ((x - mean) ** 2).mean(-1, keepdim=True)
    return (x - mean) / torch.sqrt(var + eps) * gamma + beta


def gelu(x):
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * x ** 3)))


def attention(x, wqkv, wo, bo, heads):
    b, n, dim = x.shape
    scale = dim ** -0.5  # full model dim (base_models.py:66), not head dim
    qkv = x @ wqkv  # no bias
    qkv = qkv.view(b, n, 3, heads, dim // heads).permute(2, 0, 3, 1, 4)  # "b n (qkv h d) -> qkv b h n d"
    q, k, v = qkv[0], qkv[1], qkv[2]
    dots = torch.einsum("bhid,bhjd->bhij", q, k) * scale
    attn = torch.softmax(dots, dim=-1)
    out = torch.einsum("bhij,bhjd->bhid", attn, v)
    out = out.permute(0, 2, 1, 3).reshape(b, n, dim)  # "b h n d -> b n (h d)"
    return out @ wo + bo


def mlp(x, w1, b1, w2, b2):
    return gelu(x @ w1 + b1) @ w2 + b2


def transformer(x, params, prefix, layers, heads):
    for l in range(layers):
        n = layer_names(prefix, l)
        p = {k: params[v] for k, v in n.items()}
        x = attention(layer_norm(x, p["ln1_g"], p["ln1_b"]), p["wqkv"], p["wo"], p["bo"], heads) + x
        x = mlp(layer_norm(x, p["ln2_g"], p["ln2_b"]), p["w1"], p["b1"], p["w2"], p["b2"]) + x
    return x


def fact_forward(params, cfg, motion_input, audio_input):
    feats = []
    for name, inp in (("motion", motion_input), ("audio", audio_input)):
        c = cfg[name]
        f = inp @ params[name + "_linear_embedding/kernel"] + params[name + "_linear_embedding/bias"]
        f = f + params[name + "_pos_embedding/position_embedding"]
        feats.append(transformer(f, params, name + "_transformer", c["layers"], c["heads"]))
    if feats[0].shape[-1] != feats[1].shape[-1]:
        raise ValueError("The modal_a hidden size (%d) should be the same with the modal_b hidden size (%d)"
                         % (feats[0].shape[-1], feats[1].shape[-1]))
    merged = torch.cat(feats, dim=1)  # [motion ; audio]
    merged = transformer(merged, params, "cross_modal_layer/transformer", cfg["cross"]["layers"],
                         cfg["cross"]["heads"])
    return merged @ params["cross_modal_layer/output/kernel"] + params["cross_modal_layer/output/bias"]


def motion_loss(target, pred):
    t = target.shape[1]
    return ((target - pred[:, :t]) ** 2).mean()


def infer_auto_regressive(params, cfg, motion_input, audio_input, steps=1200):
    n_a = cfg["audio"]["seq_len"]
    outputs = []
    motion = motion_input
    for i in range(steps):
        audio = audio_input[:, i:i + n_a]
        if audio.shape[1] < n_a:
            break
        out = fact_forward(params, cfg, motion, audio)[:, 0:1, :]
        outputs.append(out)
        motion = torch.cat([motion[:, 1:, :], out], dim=1)
    if not outputs:
        return motion_input.new_zeros(motion_input.shape[0], 0, cfg["out_dim"])
    return torch.cat(outputs, dim=1)


def loss_and_grads(params, cfg, motion_input, audio_input, target, num_replicas=1):
    """Gradient of mean-loss / num_replicas wrt every parameter (what one replica contributes)."""
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
    pred = fact_forward(leaves, cfg, motion_input, audio_input)
    loss = motion_loss(target, pred)
    (loss / num_replicas).backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaves.items()}
    return loss.detach(), grads, pred.detach()


def adam_update(params, grads, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-7, clip_norm=0.0):
    """Keras Adam (non-amsgrad): lr_t = lr*sqrt(1-b2^t)/(1-b1^t); p -= lr_t*m/(sqrt(v)+eps), t = step+1.
    Optional tf.clip_by_global_norm first (single_task_trainer.py:180-183)."""
    if clip_norm > 0:
        gn = math.sqrt(sum(float((g.double() ** 2).sum()) for g in grads.values()))
        scale = clip_norm / max(gn, clip_norm)
        grads = {k: g * scale for k, g in grads.items()}
    t = step + 1
    lr_t = lr * math.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t)
    new_p, new_m, new_v = {}, {}, {}
    for k in params:
        g = grads[k]
        new_m[k] = beta1 * m[k] + (1 - beta1) * g
        new_v[k] = beta2 * v[k] + (1 - beta2) * g * g
        new_p[k] = params[k] - lr_t * new_m[k] / (torch.sqrt(new_v[k]) + eps)
    return new_p, new_m, new_v


def train_step(params, m, v, step, cfg, batch, lr, num_replicas=1, clip_norm=0.0):
    """One optimizer step of the reference train_fn for a single replica's batch."""
    loss, grads, _ = loss_and_grads(params, cfg, batch["motion_input"], batch["audio_input"],
                                    batch["target"], num_replicas)
    p, m, v = adam_update(params, grads, m, v, step, lr, clip_norm=clip_norm)
    return loss, grads, p, m, v


def synthetic_batch(cfg, batch, target_len, seed=0, dtype=torch.float64):
    """N(0,1) motion/audio/target tensors from a seeded CPU generator."""
    gen = torch.Generator().manual_seed(seed)
    return {
        "motion_input": torch.randn(batch, cfg["motion"]["seq_len"], cfg["motion"]["feature_dim"],
                                    generator=gen).to(dtype),
        "audio_input": torch.randn(batch, cfg["audio"]["seq_len"], cfg["audio"]["feature_dim"],
                                   generator=gen).to(dtype),
        "target": torch.randn(batch, target_len, cfg["out_dim"], generator=gen).to(dtype),
    }


TINY_CFG = {  # BASELINE.json configs[0] as fixed in SURVEY section 8(d)
    "motion": {"seq_len": 32, "feature_dim": 225, "hidden": 128, "layers": 2, "heads": 4, "ff": 512},
    "audio": {"seq_len": 64, "feature_dim": 35, "hidden": 128, "layers": 2, "heads": 4, "ff": 512},
    "cross": {"hidden": 128, "layers": 2, "heads": 4, "ff": 512},
    "out_dim": 225,
}

FACT_V5_CFG = {  # configs/fact_v5_deeper_t10_cm12.config + proto defaults
    "motion": {"seq_len": 120, "feature_dim": 225, "hidden": 800, "layers": 2, "heads": 10, "ff": 3072},
    "audio": {"seq_len": 240, "feature_dim": 35, "hidden": 800, "layers": 2, "heads": 10, "ff": 3072},
    "cross": {"hidden": 800, "layers": 12, "heads": 10, "ff": 3072},
    "out_dim": 225,
}


def num_params(cfg):
    return sum(int(torch.Size(s).numel()) for _, s in param_shapes(cfg))
