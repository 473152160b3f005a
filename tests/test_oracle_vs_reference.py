"""Pins the CPU oracle on the REFERENCE'S OWN model code.

tests/golden/reference_tiny_golden.npz was produced by importing /root/reference/mint/core/*.py unchanged and
running `model_builder.build() -> FACTModel.call / .loss / .infer_auto_regressive` on top of a small stand-in for
the TensorFlow/Keras primitives (tests/golden/ref_shim, generator tests/golden/make_reference_golden.py).
The oracle restates the same algorithm independently; in float64 the two agree to the last bit, so the
tolerance here is 1e-12.  Where the reference checkout is present (this container, not the GPU box) the
reference code is also re-run live and compared with the committed fixture.
"""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.join(HERE, "golden"))
from oracle import fact_oracle as O  # noqa: E402
import make_reference_golden as G    # noqa: E402

FIX = os.path.join(HERE, "golden", "reference_tiny_golden.npz")
TOL = 1e-12


@pytest.fixture(scope="module")
def fx():
    d = np.load(FIX)
    params = G.golden_params(O, O.TINY_CFG)
    flat = torch.cat([params[n].reshape(-1) for n, _ in O.param_shapes(O.TINY_CFG)])
    # the regenerated weights are the ones the reference ran with
    assert abs(float(flat.sum()) - float(d["params_sum"])) < 1e-9
    assert abs(float(flat.abs().sum()) - float(d["params_abs_sum"])) < 1e-9
    assert np.array_equal(flat[::9973].numpy(), d["params_probe"])
    return d, params


def _t(a):
    return torch.from_numpy(np.asarray(a))


def test_forward_matches_reference_code(fx):
    d, params = fx
    pred = O.fact_forward(params, O.TINY_CFG, _t(d["motion_input"]), _t(d["audio_input"]))
    assert pred.shape == (2, 96, 225)
    assert float((pred - _t(d["ref_pred"])).abs().max()) <= TOL


def test_all_ones_inputs_of_the_reference_test(fx):
    """Inputs of mint/core/fact_model_test.py:47-52 (all ones); the reference test pins only the shape."""
    d, params = fx
    pred = O.fact_forward(params, O.TINY_CFG, torch.ones(1, 32, 225, dtype=torch.float64),
                          torch.ones(1, 64, 35, dtype=torch.float64))
    assert float((pred - _t(d["ref_all_ones_pred"])).abs().max()) <= TOL


def test_loss_matches_reference_code(fx):
    d, params = fx
    pred = O.fact_forward(params, O.TINY_CFG, _t(d["motion_input"]), _t(d["audio_input"]))
    assert abs(float(O.motion_loss(_t(d["target"]), pred)) - float(d["ref_loss"])) <= TOL


def test_auto_regressive_matches_reference_code(fx):
    """6 steps requested, the 67-frame audio track admits 4 windows: the early break is the reference's."""
    d, params = fx
    ar = O.infer_auto_regressive(params, O.TINY_CFG, _t(d["motion_input"]), _t(d["ar_audio"]), steps=6)
    assert ar.shape == (2, 4, 225)
    assert float((ar - _t(d["ref_ar"])).abs().max()) <= TOL


def test_gradients_match_autograd_through_reference_forward(fx):
    d, params = fx
    _, grads, _ = O.loss_and_grads(params, O.TINY_CFG, _t(d["motion_input"]), _t(d["audio_input"]), _t(d["target"]))
    names = [n for n, _ in O.param_shapes(O.TINY_CFG)]
    norms = np.array([float(grads[n].norm()) for n in names])
    sums = np.array([float(grads[n].sum()) for n in names])
    assert np.allclose(norms, d["ref_grad_norms"], rtol=1e-9, atol=1e-14)
    assert np.allclose(sums, d["ref_grad_sums"], rtol=1e-7, atol=1e-12)


@pytest.mark.skipif(not os.path.isdir(os.path.join(G.REF, "mint", "core")), reason="reference checkout not present")
def test_reference_code_rerun_reproduces_fixture(fx):
    """Re-imports the reference and re-runs it (needs /root/reference; skipped on the GPU box)."""
    d, params = fx
    ref = G.run_reference(O.TINY_CFG, params, _t(d["motion_input"]), _t(d["audio_input"]), _t(d["target"]),
                          _t(d["ar_audio"]), 6)
    assert float((ref["pred"] - _t(d["ref_pred"])).abs().max()) <= TOL
    assert abs(float(ref["loss"]) - float(d["ref_loss"])) <= TOL
    assert float((ref["ar"] - _t(d["ref_ar"])).abs().max()) <= TOL


@pytest.mark.skipif(not os.path.isdir(os.path.join(G.REF, "mint", "core")), reason="reference checkout not present")
def test_reference_width_mismatch_error_is_the_one_we_mirror():
    """base_models.py:184-189: different modal widths raise ValueError (the engine returns -2 for it)."""
    cfg = {k: (dict(v) if isinstance(v, dict) else v) for k, v in O.TINY_CFG.items()}
    cfg["audio"]["hidden"] = 64
    model_builder, model_pb2 = G.import_reference()
    import tensorflow as tf
    m = model_builder.build(G.proto_from_cfg(model_pb2, cfg), True)
    with pytest.raises(ValueError, match="should be the same"):
        m({"motion_input": tf.constant(torch.zeros(1, 32, 225)), "audio_input": tf.constant(torch.zeros(1, 64, 35))})


# ---- the callers either side of the model, reference code run live under the same shim -------
_HAS_REF = os.path.isdir(os.path.join(G.REF, "mint", "core"))


@pytest.mark.skipif(not _HAS_REF, reason="reference checkout not present")
@pytest.mark.parametrize("warmup", [False, True])
def test_manual_stepping_matches_reference_class(warmup):
    """mint/core/learning_schedules.py:19-67 (the schedule fact_v5_deeper_t10_cm12.config selects) against
    mint_amd.learning_schedules.ManualStepping at boundaries, between them and far beyond."""
    G.import_reference()
    from mint.core import learning_schedules as ref_ls
    from mint_amd import learning_schedules as our_ls
    bounds, rates = [100, 1500, 4000], [1e-4, 1e-5, 1e-6, 1e-7]
    ref = ref_ls.ManualStepping(list(bounds), list(rates), warmup)
    ours = our_ls.ManualStepping(list(bounds), list(rates), warmup)
    for step in [0, 1, 37, 99, 100, 101, 1499, 1500, 3999, 4000, 4001, 10 ** 6]:
        assert float(ours(step)) == pytest.approx(float(ref(step)), rel=1e-12, abs=0.0), step


@pytest.mark.skipif(not _HAS_REF, reason="reference checkout not present")
@pytest.mark.parametrize("is_training,start", [(True, 0), (True, 17), (True, 60), (False, 0)])
def test_fact_preprocessing_matches_reference_function(is_training, start):
    """mint/utils/inputs_util.py:59-107 run under the shim (its random window start forced to `start`)
    against mint_amd.inputs_util.fact_preprocessing on the same example: 6-column left pad, window slices,
    target shift, whole audio track in eval."""
    G.import_reference()
    import tensorflow as tf
    from mint.utils import inputs_util as ref_iu
    from mint_amd import inputs_util as our_iu
    rs = np.random.RandomState(5)
    motion = rs.randn(200, 219).astype(np.float32)
    audio = rs.randn(200, 35).astype(np.float32)
    params = {"motion": {"input_length": 120, "target_length": 20, "target_shift": 120, "feature_dim": 219},
              "audio": {"input_length": 140, "target_length": 0, "target_shift": 0, "feature_dim": 35}}
    # window = max(120, 120 + 20, 140) = 140 -> start in [0, 61)
    tf.random.forced = start
    try:
        ref = ref_iu.fact_preprocessing({"motion_sequence": tf.constant(motion), "audio_sequence": tf.constant(audio)},
                                        params, is_training)
    finally:
        tf.random.forced = None

    class Forced:
        def randint(self, lo, hi):
            assert lo <= start < hi
            return start
    ours = our_iu.fact_preprocessing({"motion_sequence": motion, "audio_sequence": audio}, params, is_training,
                                     rng=Forced())
    assert set(ours.keys()) == set(ref.keys())
    for k in ref:
        r = np.asarray(ref[k].as_subclass(torch.Tensor), dtype=np.float64)
        assert ours[k].shape == r.shape, k
        assert np.array_equal(np.asarray(ours[k], dtype=np.float64), r), k


# ---------------------------------------------------------------------------------------------------------------
# The same pin at the REAL configuration (fact_v5_deeper_t10_cm12 dimensions, one sample): the oracle's restatement
# and the reference's model code agree where the GPU parity tests use the oracle (tests/test_gpu_model.py:
# test_fact_v5_forward_and_grads_vs_oracle, test_fact_v5_autoregressive_vs_oracle).  float64; the two sum in different
# orders at d = 800 / n = 360, hence a relative tolerance instead of the tiny config's 1e-12.
# ---------------------------------------------------------------------------------------------------------------
FIX_V5 = os.path.join(HERE, "golden", "reference_v5_golden.npz")
RTOL_V5 = 1e-9


@pytest.fixture(scope="module")
def fx5():
    d = np.load(FIX_V5)
    params = G.golden_params(O, O.FACT_V5_CFG)
    flat = torch.cat([params[n].reshape(-1) for n, _ in O.param_shapes(O.FACT_V5_CFG)])
    assert flat.numel() == 120406977  # SURVEY 8a parameter inventory
    assert abs(float(flat.sum()) - float(d["params_sum"])) < 1e-6
    assert abs(float(flat.abs().sum()) - float(d["params_abs_sum"])) < 1e-6
    assert np.array_equal(flat[::999983].numpy(), d["params_probe"])
    return d, params


def _rel(a, b):
    return float((a - b).norm() / b.norm())


def test_v5_forward_and_loss_match_reference_code(fx5):
    d, params = fx5
    pred = O.fact_forward(params, O.FACT_V5_CFG, _t(d["motion_input"]), _t(d["audio_input"]))
    assert pred.shape == (1, 360, 225)
    assert _rel(pred, _t(d["ref_pred"])) <= RTOL_V5
    loss = float(O.motion_loss(_t(d["target"]), pred))
    assert abs(loss - float(d["ref_loss"])) <= RTOL_V5 * abs(float(d["ref_loss"]))


def test_v5_auto_regressive_matches_reference_code(fx5):
    """3 steps requested, the 241-frame audio track admits 2 windows (early break of fact_model.py:124-126)."""
    d, params = fx5
    ar = O.infer_auto_regressive(params, O.FACT_V5_CFG, _t(d["motion_input"]), _t(d["ar_audio"]), steps=3)
    assert ar.shape == (1, 2, 225)
    assert _rel(ar, _t(d["ref_ar"])) <= RTOL_V5


def test_v5_gradients_match_autograd_through_reference_forward(fx5):
    d, params = fx5
    _, grads, _ = O.loss_and_grads(params, O.FACT_V5_CFG, _t(d["motion_input"]), _t(d["audio_input"]), _t(d["target"]))
    names = [n for n, _ in O.param_shapes(O.FACT_V5_CFG)]
    assert len(names) == 184
    norms = np.array([float(grads[n].norm()) for n in names])
    sums = np.array([float(grads[n].sum()) for n in names])
    assert np.allclose(norms, d["ref_grad_norms"], rtol=1e-8, atol=1e-14)
    assert np.allclose(sums, d["ref_grad_sums"], rtol=1e-6, atol=1e-10)


@pytest.mark.skipif(not os.path.isdir(os.path.join(G.REF, "mint", "core")), reason="reference checkout not present")
def test_v5_reference_code_rerun_reproduces_fixture(fx5):
    """Re-imports the reference and re-runs it at the fact_v5 dimensions (needs /root/reference; skipped on the GPU box)."""
    d, params = fx5
    ref = G.run_reference(O.FACT_V5_CFG, params, _t(d["motion_input"]), _t(d["audio_input"]), _t(d["target"]),
                          _t(d["ar_audio"]), 3)
    assert _rel(ref["pred"], _t(d["ref_pred"])) <= 1e-12
    assert abs(float(ref["loss"]) - float(d["ref_loss"])) <= 1e-12 * abs(float(d["ref_loss"]))
    assert _rel(ref["ar"], _t(d["ref_ar"])) <= 1e-12
