"""The path a user of the reference walks, on the engine: TFRecord files -> `create_input` (FACT windows, device staging) ->
`SingleTaskTrainer` (trainer.py:138-178) and `SingleTaskEvaluator` (evaluator.py:44-71, single_task_evaluator.py:67-86),
with the sequence lengths of the shipped config (120 / 240 frames, 20 target frames) on a narrow model."""
import numpy as np
import pytest
import torch

from mint_amd import inputs, model_builder, protos, tfrecord
from mint_amd.evaluator import SingleTaskEvaluator
from mint_amd.trainer import Adam, SingleTaskTrainer
from tests.test_gpu_model import make_config

pytestmark = pytest.mark.gpu

CFG = {"motion": {"seq_len": 120, "feature_dim": 225, "hidden": 128, "layers": 1, "heads": 4, "ff": 256},
       "audio": {"seq_len": 240, "feature_dim": 35, "hidden": 128, "layers": 1, "heads": 4, "ff": 256},
       "cross": {"hidden": 128, "layers": 2, "heads": 4, "ff": 256}, "out_dim": 225}


def _dataset(tmp_path, name, n_tracks, seed):
    d = protos.Dataset()
    d.name = name
    d.input_length_sec, d.target_length_sec, d.target_shift_sec = 120.0, 20, 120
    for mod, dim, rate in (("motion", 219, 1), ("audio", 35, 2)):
        g = d.modality.add().general_modality
        g.feature_name, g.dimension, g.sample_rate = mod, dim, rate
    d.data_augmentation_options.add().fact_preprocessor.CopyFrom(protos.FACTPreprocessor())
    rng = np.random.RandomState(seed)
    recs, tracks = [], []
    for i in range(n_tracks):
        n = 280 + 9 * i
        m, a = rng.randn(n, 219).astype(np.float32), rng.randn(n + 3 * i, 35).astype(np.float32)
        tracks.append((m, a))
        recs.append(tfrecord.make_example({
            "motion_name": "gBR_sBM_c%02d" % i, "motion_sequence": m.flatten(), "motion_sequence_shape": np.array(m.shape),
            "audio_name": "mBR%d" % i, "audio_sequence": a.flatten(), "audio_sequence_shape": np.array(a.shape)}))
    tfrecord.write_records(str(tmp_path / ("aist_tfrecord-%s-0" % name)), recs)
    d.data_files = str(tmp_path / ("*_tfrecord-%s*" % name))
    return d, tracks


def test_training_from_files_equals_training_from_the_same_tensors(tmp_path):
    ds, _ = _dataset(tmp_path, "train", 6, 0)
    tc = protos.TrainConfig()
    tc.batch_size = 4
    fed = inputs.create_input(tc, ds, is_training=True, seed=3, device="cuda")
    host = inputs.create_input(tc, ds, is_training=True, seed=3, prefetch_batches=0)   # the same stream, host tensors
    strip = lambda b: {k: v for k, v in b.items() if not k.endswith("_name")}
    losses = []
    for source in ((strip(b) for b in fed), ({k: v.cuda() for k, v in strip(b).items()} for b in host)):
        model = model_builder.build(make_config(CFG), True)
        tr = SingleTaskTrainer(source, "target", model, optimizer=Adam(1e-3))
        tr.train_loop_begin()
        it = iter(source)
        losses.append([float(tr.train_step(it)) for _ in range(5)])
        assert model.global_step == 5
    assert all(np.isfinite(losses[0]))
    assert losses[0] == pytest.approx(losses[1], rel=1e-4)


def test_evaluator_from_files_writes_seed_plus_generated_frames(tmp_path):
    ds, tracks = _dataset(tmp_path, "val", 3, 1)
    ec = protos.EvalConfig()
    ec.batch_size = 1   # the shipped eval_config
    data = inputs.create_input(ec, ds, is_training=False, device="cuda")
    model = model_builder.build(make_config(CFG), True)
    ev = SingleTaskEvaluator(data, model, model.get_metrics(ec), output_dir=str(tmp_path / "out"), steps=1200)
    assert ev.evaluate() == {}
    for i, (m, a) in enumerate(tracks):
        got = np.load(tmp_path / "out" / ("gBR_sBM_c%02d_mBR%d.npy" % (i, i)))
        n_gen = min(1200, a.shape[0] - 240 + 1)   # the sampler stops when the audio window runs short
        assert got.shape == (120 + n_gen, 225)
        np.testing.assert_array_equal(got[:120, 6:], m[:120])            # seed = first 120 frames, 6 zero columns in front
        assert float(np.abs(got[:120, :6]).sum()) == 0.0 and np.isfinite(got).all()
        # the generated frames are the engine's AR rollout on exactly this track
        ref = model.infer_auto_regressive({"motion_input": torch.from_numpy(got[None, :120]).cuda(),
                                           "audio_input": torch.from_numpy(a[None]).cuda()}, steps=1200)
        np.testing.assert_allclose(got[120:], ref[0].cpu().numpy(), rtol=2e-2, atol=2e-3)
