"""Evaluator output contract (single_task_evaluator.py:67-86) and checkpoint cadence
(trainer.py:168-173) on a CPU stand-in model."""
import os

import numpy as np
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
