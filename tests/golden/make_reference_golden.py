"""Golden vectors from the REFERENCE'S OWN model code, run in this container.

google-research/mint is TensorFlow/Keras and TensorFlow is not installed here, so the reference cannot run
as shipped.  What CAN run is its model code itself: tests/golden/ref_shim/ provides the handful of TF/Keras
primitives that code calls (Dense, LayerNormalization, softmax, einsum, concat, reduce_mean ... on
PyTorch-CPU float64) and einops supplies Rearrange, so this script imports

    /root/reference/mint/core/{model_builder,fact_model,base_models,base_model_util,...}.py   (unchanged)

builds `FACTModel` through the reference's `model_builder.build()` from a `MultiModalModel` proto, loads
seeded weights into the layers the REFERENCE constructed, and records what the REFERENCE computes:
`FACTModel.call`, `FACTModel.loss` and `FACTModel.infer_auto_regressive`.  Everything the hot path's
semantics depend on - layer order, pre-LN residual structure, the hidden**-0.5 softmax scale, the
"(qkv h d)" split, tanh-GELU, [motion; audio] concat order, the loss slice, the AR window shift - is
therefore the reference's, not a restatement; only the primitives are ours (documented in the shim).

Output: tests/golden/reference_tiny_golden.npz (inputs, checksums of the seeded weights - the tests
regenerate them with golden_params() - and the reference's outputs), consumed by
tests/test_oracle_vs_reference.py (CPU) and tests/test_gpu_model.py (HIP engine).
Run from the repo root:   python tests/golden/make_reference_golden.py          (tiny config)
                          python tests/golden/make_reference_golden.py --v5     (fact_v5 dimensions, one sample)
"""
import os
import sys

os.environ["PROTOCOL_BUFFERS_PYTHON_IMPLEMENTATION"] = "python"  # the reference's *_pb2.py are legacy

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("MINT_REFERENCE", "/root/reference")


def import_reference():
    """Put the primitive shim and the reference checkout on sys.path and import its model builder."""
    for p in (REF, os.path.join(HERE, "ref_shim")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from mint.core import model_builder  # noqa: E402  (reference code)
    from mint.protos import model_pb2    # noqa: E402
    return model_builder, model_pb2


def proto_from_cfg(model_pb2, cfg):
    """MultiModalModel proto for an oracle-style config dict (same fields as fact_v5_deeper_t10_cm12.config)."""
    from google.protobuf import text_format

    def tr(c):
        return ("transformer: { num_attention_heads: %d hidden_size: %d num_hidden_layers: %d "
                "intermediate_size: %d }" % (c["heads"], c["hidden"], c["layers"], c["ff"]))
    txt = """
    fact_model {
      modality: { feature_name: "audio" sequence_length: %d feature_dim: %d model: { %s } }
      modality: { feature_name: "motion" sequence_length: %d feature_dim: %d model: { %s } }
      cross_modal_model: { modality_a: "motion" modality_b: "audio" %s output_layer: { out_dim: %d } }
    }""" % (cfg["audio"]["seq_len"], cfg["audio"]["feature_dim"], tr(cfg["audio"]),
            cfg["motion"]["seq_len"], cfg["motion"]["feature_dim"], tr(cfg["motion"]),
            tr(cfg["cross"]), cfg["out_dim"])
    msg = model_pb2.MultiModalModel()
    text_format.Parse(txt, msg)
    return msg


def load_params(ref_model, params, clone=True):
    """Put a {oracle name: tensor} dict into the layer objects the reference constructed."""
    def t(name):
        return params[name].to(torch.float64).clone() if clone else params[name]

    def load_stack(transformer, prefix, n_layers):
        blocks = transformer.net.layers
        assert len(blocks) == 2 * n_layers
        for l in range(n_layers):
            b = "%s/layer_%d" % (prefix, l)
            attn_block, mlp_block = blocks[2 * l].fn, blocks[2 * l + 1].fn   # Residual.fn = Norm
            attn_block.norm.gamma, attn_block.norm.beta = t(b + "/attn_norm/gamma"), t(b + "/attn_norm/beta")
            attn = attn_block.fn                                              # Norm.fn = Attention
            attn.to_qkv.kernel = t(b + "/attn/to_qkv/kernel")
            attn.to_out.kernel, attn.to_out.bias = t(b + "/attn/to_out/kernel"), t(b + "/attn/to_out/bias")
            mlp_block.norm.gamma, mlp_block.norm.beta = t(b + "/mlp_norm/gamma"), t(b + "/mlp_norm/beta")
            d1, d2 = mlp_block.fn.net.layers                                  # Norm.fn = MLP
            d1.kernel, d1.bias = t(b + "/mlp/dense_1/kernel"), t(b + "/mlp/dense_1/bias")
            d2.kernel, d2.bias = t(b + "/mlp/dense_2/kernel"), t(b + "/mlp/dense_2/bias")

    cm = ref_model.cross_modal_layer
    load_stack(cm.transformer_layer, "cross_modal_layer/transformer", len(cm.transformer_layer.net.layers) // 2)
    cm.cross_output_layer.kernel = t("cross_modal_layer/output/kernel")
    cm.cross_output_layer.bias = t("cross_modal_layer/output/bias")
    for mod in ("motion", "audio"):
        tr = getattr(ref_model, mod + "_transformer")
        load_stack(tr, mod + "_transformer", len(tr.net.layers) // 2)
        getattr(ref_model, mod + "_pos_embedding").pos_embedding = t(mod + "_pos_embedding/position_embedding")
        emb = getattr(ref_model, mod + "_linear_embedding").net
        emb.kernel, emb.bias = t(mod + "_linear_embedding/kernel"), t(mod + "_linear_embedding/bias")


def run_reference(cfg, params, motion, audio, target=None, ar_audio=None, ar_steps=0, want_grads=False):
    """Build the reference FACTModel for `cfg`, load `params`, return its outputs as float64 tensors.
    want_grads: also d(reference loss)/d(weights) by PyTorch autograd THROUGH the reference's forward code
    (the reference's own tape / Orbit train loop needs TensorFlow; the differentiated function is the
    reference's)."""
    model_builder, model_pb2 = import_reference()
    import tensorflow as tf  # the shim (import_reference put it on sys.path)
    ref_model = model_builder.build(proto_from_cfg(model_pb2, cfg), True)
    inputs = {"motion_input": tf.constant(motion), "audio_input": tf.constant(audio)}
    ref_model(inputs)             # first call builds every layer (Keras semantics), then overwrite the weights
    if want_grads:
        params = {k: v.detach().clone().to(torch.float64).requires_grad_(True) for k, v in params.items()}
        load_params(ref_model, params, clone=False)
    else:
        load_params(ref_model, params)
    out = {"pred": ref_model(inputs).as_subclass(torch.Tensor).detach()}
    if target is not None:
        loss = ref_model.loss(tf.constant(target), ref_model(inputs)).as_subclass(torch.Tensor)
        out["loss"] = loss.detach()
        if want_grads:
            loss.backward()
            out["grads"] = {k: v.grad.detach() for k, v in params.items()}
    if want_grads:
        load_params(ref_model, {k: v.detach() for k, v in params.items()})
    if ar_steps:
        ar_in = {"motion_input": tf.constant(motion), "audio_input": tf.constant(ar_audio)}
        out["ar"] = ref_model.infer_auto_regressive(ar_in, steps=ar_steps).as_subclass(torch.Tensor)
    return out


def golden_params(O, cfg):
    """Seeded weights of the fixture (regenerated by the tests; the fixture stores only their checksum):
    the oracle's reference-style initialisation plus non-trivial biases / LayerNorm affine parameters so that
    nothing is hidden behind zeros and ones."""
    params = O.init_params(cfg, seed=0)
    g = torch.Generator().manual_seed(7)
    for k, v in params.items():
        if k.endswith("/bias") or k.endswith("/beta"):
            v.copy_(torch.randn(v.shape, generator=g, dtype=torch.float64) * 0.05)
        elif k.endswith("/gamma"):
            v.copy_(1.0 + torch.randn(v.shape, generator=g, dtype=torch.float64) * 0.1)
    return params


def main():
    sys.path.insert(0, ROOT)
    from oracle import fact_oracle as O
    cfg = O.TINY_CFG
    params = golden_params(O, cfg)
    batch = O.synthetic_batch(cfg, 2, 8, seed=11)
    ar_audio = torch.cat([batch["audio_input"], batch["audio_input"][:, :3]], dim=1)  # 67 frames: 4 steps fit, the 5th does not
    ref = run_reference(cfg, params, batch["motion_input"], batch["audio_input"], batch["target"], ar_audio, 6,
                        want_grads=True)
    ones = run_reference(cfg, params, torch.ones(1, 32, 225, dtype=torch.float64),
                         torch.ones(1, 64, 35, dtype=torch.float64))  # the inputs of fact_model_test.py:47-52
    flat = torch.cat([params[n].reshape(-1) for n, _ in O.param_shapes(cfg)])
    out = {
        "params_sum": np.float64(flat.sum()), "params_abs_sum": np.float64(flat.abs().sum()),
        "params_probe": flat[::9973].numpy(), "motion_input": batch["motion_input"].numpy(),
        "audio_input": batch["audio_input"].numpy(), "target": batch["target"].numpy(), "ar_audio": ar_audio.numpy(),
        "ref_pred": ref["pred"].numpy(), "ref_loss": np.float64(ref["loss"]), "ref_ar": ref["ar"].numpy(),
        "ref_all_ones_pred": ones["pred"].numpy(),
        "ref_grad_norms": np.array([float(ref["grads"][n].norm()) for n, _ in O.param_shapes(cfg)]),
        "ref_grad_sums": np.array([float(ref["grads"][n].sum()) for n, _ in O.param_shapes(cfg)]),
    }
    path = os.path.join(HERE, "reference_tiny_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes; reference loss", float(ref["loss"]),
          "AR frames", tuple(ref["ar"].shape))


def main_v5():
    """The same, at the REAL configuration (fact_v5_deeper_t10_cm12: d = 800, 10 heads of 80, ff = 3072, 2 + 2 + 12 layers,
    120 + 240 tokens), one sample, float64: forward, loss, gradients by autograd through the reference's forward, and a
    2-step auto-regressive rollout.  Written to tests/golden/reference_v5_golden.npz (predictions as float64; per-tensor
    gradient norms and sums; the 120 M weights are regenerated from the seed by the tests and pinned by checksums)."""
    sys.path.insert(0, ROOT)
    from oracle import fact_oracle as O
    cfg = O.FACT_V5_CFG
    params = golden_params(O, cfg)
    batch = O.synthetic_batch(cfg, 1, 20, seed=12)
    ar_audio = torch.cat([batch["audio_input"], batch["audio_input"][:, :1]], dim=1)  # 241 frames: exactly 2 windows
    ref = run_reference(cfg, params, batch["motion_input"], batch["audio_input"], batch["target"], ar_audio, 3,
                        want_grads=True)
    flat = torch.cat([params[n].reshape(-1) for n, _ in O.param_shapes(cfg)])
    out = {
        "params_sum": np.float64(flat.sum()), "params_abs_sum": np.float64(flat.abs().sum()),
        "params_probe": flat[::999983].numpy(), "motion_input": batch["motion_input"].numpy(),
        "audio_input": batch["audio_input"].numpy(), "target": batch["target"].numpy(), "ar_audio": ar_audio.numpy(),
        "ref_pred": ref["pred"].numpy(), "ref_loss": np.float64(ref["loss"]), "ref_ar": ref["ar"].numpy(),
        "ref_grad_norms": np.array([float(ref["grads"][n].norm()) for n, _ in O.param_shapes(cfg)]),
        "ref_grad_sums": np.array([float(ref["grads"][n].sum()) for n, _ in O.param_shapes(cfg)]),
    }
    path = os.path.join(HERE, "reference_v5_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes; reference loss", float(ref["loss"]),
          "AR frames", tuple(ref["ar"].shape))


if __name__ == "__main__":
    if "--v5" in sys.argv:
        main_v5()
    else:
        main()
