"""Input contract without tf.data: TFRecord/Example round trip in the writer format of
tools/preprocessing.py:54-69, windowing semantics of mint/utils/inputs_util.py:59-107, batching of
mint/core/inputs.py:118-123, and the reference's own numeric check of get_modality_to_param_dict
(mint/utils/inputs_util_test.py:22-49)."""
import numpy as np
import pytest

from mint_amd import inputs, inputs_util, protos, tfrecord


def _dataset_cfg(files=""):
    d = protos.Dataset()
    d.name, d.data_files = "train", files
    d.input_length_sec, d.target_length_sec, d.target_shift_sec = 120.0, 20, 120
    for name, dim, rate in (("motion", 219, 1), ("audio", 35, 2)):
        g = d.modality.add().general_modality
        g.feature_name, g.dimension, g.sample_rate = name, dim, rate
    d.data_augmentation_options.add().fact_preprocessor.CopyFrom(protos.FACTPreprocessor())
    return d


def test_crc32c_known_answers():
    assert tfrecord.crc32c(b"123456789") == 0xE3069283          # RFC 3720 check value
    assert tfrecord.crc32c(b"\x00" * 32) == 0x8A9136AA


def test_modality_params_like_reference_test():
    p = inputs_util.get_modality_to_param_dict(_dataset_cfg())
    assert p["motion"] == {"feature_dim": 219, "input_length": 120, "target_length": 20, "target_shift": 120,
                           "sample_rate": 1, "resize": 0, "crop_size": 0}
    assert p["audio"]["input_length"] == 240 and p["audio"]["target_length"] == 40 and p["audio"]["feature_dim"] == 35


def test_example_roundtrip_and_windowing(tmp_path):
    rng = np.random.RandomState(0)
    recs = []
    for i in range(5):
        n = 300 + 10 * i
        motion = rng.randn(n, 219).astype(np.float32)
        audio = rng.randn(n + 7, 35).astype(np.float32)
        recs.append((motion, audio))
    path = str(tmp_path / "aist_tfrecord-train-0")
    tfrecord.write_records(path, [tfrecord.make_example({
        "motion_name": "m%d" % i, "motion_sequence": m.flatten(), "motion_sequence_shape": np.array(m.shape),
        "audio_name": "a%d" % i, "audio_sequence": a.flatten(), "audio_sequence_shape": np.array(a.shape)})
        for i, (m, a) in enumerate(recs)])
    got = [tfrecord.parse_example(p) for p in tfrecord.read_records(path, verify=True)]
    assert len(got) == 5 and got[2]["motion_name"] == [b"m2"]
    np.testing.assert_array_equal(got[3]["motion_sequence"].reshape(got[3]["motion_sequence_shape"]), recs[3][0])
    np.testing.assert_array_equal(got[3]["audio_sequence_shape"], [337, 35])

    cfg = _dataset_cfg(str(tmp_path / "*_tfrecord-train*"))
    tc = protos.TrainConfig()
    tc.batch_size = 2
    it = inputs.create_input(tc, cfg, is_training=True, seed=0)
    batch = next(it)
    assert batch["motion_input"].shape == (2, 120, 225) and batch["audio_input"].shape == (2, 240, 35)
    assert batch["target"].shape == (2, 20, 225) and len(batch["motion_name"]) == 2
    assert float(batch["motion_input"][:, :, :6].abs().sum()) == 0.0  # 6 zero-padded translation columns
    # the window is consistent: target = motion shifted by 120 frames from the same start
    i = int(batch["motion_name"][0][1:])
    m = np.pad(recs[i][0], [[0, 0], [6, 0]])
    mi = batch["motion_input"][0].numpy()
    starts = [s for s in range(m.shape[0] - 240 + 1) if np.array_equal(m[s:s + 120], mi)]
    assert len(starts) == 1
    np.testing.assert_array_equal(batch["target"][0].numpy(), m[starts[0] + 120:starts[0] + 140])
    np.testing.assert_array_equal(batch["audio_input"][0].numpy(), recs[i][1][starts[0]:starts[0] + 240])
    # eval: start 0, whole audio track, remainder kept, single pass
    ec = protos.EvalConfig()
    ec.batch_size = 1
    ev = list(inputs.create_input(ec, cfg, is_training=False))
    assert len(ev) == 5 and ev[0]["audio_input"].shape == (1, 307, 35) and "target" not in ev[0]
    np.testing.assert_array_equal(ev[1]["motion_input"][0].numpy()[:, 6:], recs[1][0][:120])


def test_training_window_too_short_raises():
    p = inputs_util.get_modality_to_param_dict(_dataset_cfg())
    with pytest.raises(ValueError):
        inputs_util.fact_preprocessing({"motion_sequence": np.zeros((100, 219)), "audio_sequence": np.zeros((300, 35))},
                                       p, True, np.random.RandomState(0))


def test_reference_inputs_util_test_vector():
    """mint/utils/inputs_util_test.py:22-49 verbatim expectations (int(sec * rate))."""
    d = protos.Dataset()
    d.window_type = "BEGINNING"
    d.input_length_sec, d.target_length_sec, d.target_shift_sec = 1.0, 0.5, 0.2
    motion = protos.GeneralModality()
    motion.feature_name, motion.dimension, motion.sample_rate = "motion", 34, 10
    visual = protos.GeneralModality()
    visual.feature_name, visual.dimension, visual.sample_rate = "visual", 1024, 20
    d.modality.add().general_modality.CopyFrom(motion)
    d.modality.add().general_modality.CopyFrom(visual)
    p = inputs_util.get_modality_to_param_dict(d)
    assert (p["motion"]["input_length"], p["motion"]["target_length"], p["motion"]["target_shift"]) == (10, 5, 2)
    assert (p["visual"]["input_length"], p["visual"]["target_length"], p["visual"]["target_shift"]) == (20, 10, 4)
