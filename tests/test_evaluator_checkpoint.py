"""Evaluator output contract (single_task_evaluator.py:67-86) and checkpoint cadence
(trainer.py:168-173) on a CPU stand-in model."""
import os

import numpy as np
import pytest
import torch

from mint_amd.checkpoint import CheckpointManager
from mint_amd.evaluator import SingleTaskEvaluator
from mint_amd.trainer import Adam


class _FakeModel:
    def __init__(self):
        self.w = torch.zeros(3)
        self.global_step = 0

    def infer_auto_regressive(self, inputs, steps=1200):
        b = inputs["motion_input"].shape[0]
        n = min(steps, inputs["audio_input"].shape[1] - 4 + 1)
        return torch.arange(b * n * 5, dtype=torch.float32).view(b, n, 5)

    def state_dict(self):
        return {"params": self.w.clone(), "global_step": self.global_step}

    def load_state_dict(self, s):
        self.w = s["params"].clone()
        self.global_step = s["global_step"]


def test_evaluator_saves_seed_plus_generated(tmp_path):
    batch = {"motion_input": torch.ones(2, 3, 5), "audio_input": torch.zeros(2, 9, 2),
             "motion_name": ["gBR_sBM_c01", "gPO_sFM_c02"], "audio_name": ["mBR0", "mPO1"]}
    ev = SingleTaskEvaluator([batch], _FakeModel(), [], output_dir=str(tmp_path / "out"), steps=1200)
    assert ev.evaluate() == {}
    a = np.load(tmp_path / "out" / "gBR_sBM_c01_mBR0.npy")
    assert a.shape == (3 + 6, 5) and (a[:3] == 1).all() and a[3, 1] == 1.0
    assert os.path.exists(tmp_path / "out" / "gPO_sFM_c02_mPO1.npy")


def test_checkpoint_interval_keep_and_resume(tmp_path):
    model, opt = _FakeModel(), Adam(1e-3)
    mgr = CheckpointManager(model, opt, str(tmp_path), checkpoint_interval=10, max_to_keep=2)
    assert mgr.restore_or_initialize() is None
    for step in range(1, 41):
        opt.iterations = step
        model.global_step = step
        model.w += 1
        mgr.save()
    names = sorted(os.listdir(tmp_path))
    assert names == ["ckpt-30.pt", "ckpt-40.pt"]
    m2, o2 = _FakeModel(), Adam(1e-3)
    assert CheckpointManager(m2, o2, str(tmp_path)).restore_or_initialize().endswith("ckpt-40.pt")
    assert o2.iterations == 40 and float(m2.w[0]) == 40.0


@pytest.mark.gpu
def test_tf_object_graph_checkpoint_export_import_on_engine(tmp_path):
    """Row f3 on the engine: a trained model (weights, Adam m / v, iteration counter) written as a TensorFlow
    object-graph checkpoint with the reference's variable paths (mint_amd/tf_checkpoint.py) and imported into a fresh
    model reproduces the forward pass exactly and continues training identically."""
    import torch
    from mint_amd import checkpoint, configs, model_builder
    from mint_amd.trainer import Adam, SingleTaskTrainer
    from oracle import fact_oracle as O
    cfg = O.TINY_CFG
    batch = {k: v.float().cuda() for k, v in O.synthetic_batch(cfg, 4, 8, seed=2).items()}

    def make():
        m = model_builder.build(configs.tiny_fact(), True)
        m.build(4, 225, 35)
        return m
    a = make()
    opt_a = Adam(1e-3)
    tr_a = SingleTaskTrainer([batch], "target", a, optimizer=opt_a)
    for _ in range(3):
        tr_a.train_step(iter([batch]))
    prefix = checkpoint.export_tf_checkpoint(a, str(tmp_path / "ckpt-3"), opt_a)
    b = make()
    opt_b = Adam(1e-3)
    assert checkpoint.import_tf_checkpoint(b, str(tmp_path), opt_b, verify_crc=True) == prefix
    assert opt_b.iterations == opt_a.iterations == 3
    inp = {k: v for k, v in batch.items() if k != "target"}
    assert torch.equal(a(inp), b(inp))
    for arena in ("params", "adam_m", "adam_v"):
        assert torch.equal(a._arena[arena], b._arena[arena]), arena
    tr_b = SingleTaskTrainer([batch], "target", b, optimizer=opt_b)
    la, lb = float(tr_a.train_step(iter([batch]))), float(tr_b.train_step(iter([batch])))
    # (the training step sums split-K partials with fp32 atomics: equal up to summation order, not bit for bit)
    assert abs(la - lb) < 1e-5 * abs(la), (la, lb)
    d = (a._arena["params"] - b._arena["params"]).norm() / a._arena["params"].norm()
    assert float(d) < 1e-5, float(d)
