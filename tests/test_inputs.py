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


def test_prefetch_thread_preserves_order_and_propagates_errors():
    """inputs.prefetch = the reference's ds.prefetch: a background producer thread, same order, errors re-raised in
    the consumer, producer stopped when the consumer goes away."""
    import threading
    import time
    from mint_amd import inputs

    produced = []

    def gen(n, fail_at=None):
        for i in range(n):
            if fail_at is not None and i == fail_at:
                raise RuntimeError("boom at %d" % i)
            produced.append(i)
            yield {"i": i}

    assert [b["i"] for b in inputs.prefetch(gen(20), depth=3)] == list(range(20))
    # runs ahead of the consumer, but never more than depth (+ the item in hand) batches
    del produced[:]
    it = inputs.prefetch(gen(100), depth=2)
    assert next(it)["i"] == 0
    time.sleep(0.3)
    assert 2 <= len(produced) <= 4, produced
    it.close()  # consumer gone: the producer thread must stop
    time.sleep(0.3)
    n = len(produced)
    time.sleep(0.3)
    assert len(produced) == n and n < 100
    assert not any(t.name == "mint_amd-input-prefetch" and t.is_alive() for t in threading.enumerate())
    # producer exception surfaces in the consumer after the batches before it
    got = []
    with pytest.raises(RuntimeError, match="boom at 3"):
        for b in inputs.prefetch(gen(10, fail_at=3), depth=2):
            got.append(b["i"])
    assert got == [0, 1, 2]


def _example_classes():
    """tf.train.Example message classes built with the OFFICIAL protobuf runtime from the published schema
    (tensorflow/core/example/{feature,example}.proto) - an encoder independent of mint_amd.tfrecord's own writer."""
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    fd = descriptor_pb2.FileDescriptorProto()
    fd.name, fd.package, fd.syntax = "tf_example_fixture.proto", "tensorflow", "proto3"
    T = descriptor_pb2.FieldDescriptorProto

    def msg(name, fields):
        m = fd.message_type.add()
        m.name = name
        for (fname, num, typ, label, type_name, packed) in fields:
            f = m.field.add()
            f.name, f.number, f.type, f.label = fname, num, typ, label
            if type_name:
                f.type_name = type_name
            if packed:
                f.options.packed = True
        return m
    msg("BytesList", [("value", 1, T.TYPE_BYTES, T.LABEL_REPEATED, None, False)])
    msg("FloatList", [("value", 1, T.TYPE_FLOAT, T.LABEL_REPEATED, None, True)])
    msg("Int64List", [("value", 1, T.TYPE_INT64, T.LABEL_REPEATED, None, True)])
    feat = msg("Feature", [("bytes_list", 1, T.TYPE_MESSAGE, T.LABEL_OPTIONAL, ".tensorflow.BytesList", False),
                           ("float_list", 2, T.TYPE_MESSAGE, T.LABEL_OPTIONAL, ".tensorflow.FloatList", False),
                           ("int64_list", 3, T.TYPE_MESSAGE, T.LABEL_OPTIONAL, ".tensorflow.Int64List", False)])
    feat.oneof_decl.add().name = "kind"
    for f in feat.field:
        f.oneof_index = 0
    feats = msg("Features", [("feature", 1, T.TYPE_MESSAGE, T.LABEL_REPEATED, ".tensorflow.Features.FeatureEntry", False)])
    entry = feats.nested_type.add()
    entry.name = "FeatureEntry"
    entry.options.map_entry = True
    for (fname, num, typ, tn) in (("key", 1, T.TYPE_STRING, None), ("value", 2, T.TYPE_MESSAGE, ".tensorflow.Feature")):
        f = entry.field.add()
        f.name, f.number, f.type, f.label = fname, num, typ, T.LABEL_OPTIONAL
        if tn:
            f.type_name = tn
    msg("Example", [("features", 1, T.TYPE_MESSAGE, T.LABEL_OPTIONAL, ".tensorflow.Features", False)])
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    return message_factory.GetMessageClass(pool.FindMessageTypeByName("tensorflow.Example"))


def test_reader_parses_examples_encoded_by_the_official_protobuf_runtime(tmp_path):
    """The TFRecord / tf.train.Example reader against records that were NOT produced by mint_amd's own writer: the
    Examples are serialised by google.protobuf from the published tf.train.Example schema (packed float / int64 lists,
    map<string, Feature>), with the feature layout of the reference's tools/preprocessing.py:54-69, and framed per the
    TFRecord spec (little-endian u64 length, masked crc32c of length and payload)."""
    pytest.importorskip("google.protobuf")
    import struct
    Example = _example_classes()
    rng = np.random.RandomState(4)
    recs = []
    for i in range(3):
        motion = rng.randn(30 + i, 225).astype(np.float32)
        audio = rng.randn(40 + i, 35).astype(np.float32)
        ex = Example()
        f = ex.features.feature
        f["motion_name"].bytes_list.value.append(("gBR_sBM_c01_d04_mBR0_ch0%d" % i).encode())
        f["motion_sequence"].float_list.value.extend(motion.reshape(-1).tolist())
        f["motion_sequence_shape"].int64_list.value.extend(motion.shape)
        f["audio_name"].bytes_list.value.append(("mBR%d" % i).encode())
        f["audio_sequence"].float_list.value.extend(audio.reshape(-1).tolist())
        f["audio_sequence_shape"].int64_list.value.extend(audio.shape)
        recs.append((ex.SerializeToString(), motion, audio))
    path = tmp_path / "official.tfrecord"
    with open(path, "wb") as fh:
        for payload, _, _ in recs:
            hdr = struct.pack("<Q", len(payload))
            fh.write(hdr + struct.pack("<I", tfrecord._masked(hdr)) + payload + struct.pack("<I", tfrecord._masked(payload)))
    got = list(tfrecord.read_records(str(path)))
    assert len(got) == 3
    for payload, (_, motion, audio) in zip(got, recs):
        ex = inputs._decode(payload, ["motion", "audio"])
        assert ex["motion_name"].startswith("gBR_sBM") and ex["audio_name"].startswith("mBR")
        np.testing.assert_array_equal(ex["motion_sequence"], motion)
        np.testing.assert_array_equal(ex["audio_sequence"], audio)


def test_decoded_track_cache_does_not_change_the_stream(tmp_path):
    """create_input(cache_decoded_bytes=...) keeps decoded tracks across epochs; batches, order and random windows must be
    exactly those of the re-parse-every-epoch pipeline (same seed), over several passes of the files."""
    rng = np.random.RandomState(1)
    for f in range(2):
        recs = []
        for i in range(3):
            n = 300 + 17 * i + 5 * f
            m, a = rng.randn(n, 219).astype(np.float32), rng.randn(2 * n, 35).astype(np.float32)
            recs.append(tfrecord.make_example({
                "motion_name": "m%d_%d" % (f, i), "motion_sequence": m.flatten(), "motion_sequence_shape": np.array(m.shape),
                "audio_name": "a%d_%d" % (f, i), "audio_sequence": a.flatten(), "audio_sequence_shape": np.array(a.shape)}))
        tfrecord.write_records(str(tmp_path / ("aist_tfrecord-train-%d" % f)), recs)
    cfg = _dataset_cfg(str(tmp_path / "*_tfrecord-train*"))
    tc = protos.TrainConfig()
    tc.batch_size = 4
    streams = [inputs.create_input(tc, cfg, is_training=True, seed=5, prefetch_batches=0, cache_decoded_bytes=c)
               for c in (0, 1 << 30, 200_000)]   # off, everything cached, room for one file only
    for _ in range(9):   # 36 samples = 6 passes over the 6 tracks
        ref, full, part = [next(s) for s in streams]
        for other in (full, part):
            assert other["motion_name"] == ref["motion_name"] and other["audio_name"] == ref["audio_name"]
            for k in ("motion_input", "audio_input", "target"):
                assert np.array_equal(other[k].numpy(), ref[k].numpy()), k


@pytest.mark.gpu
def test_device_staging_ring_delivers_the_cpu_stream(tmp_path):
    """create_input(device=cuda): pre-pinned ring + copy stream + event hand-over (inputs._DeviceStager) must deliver exactly
    the batches of the host pipeline, also after the ring wrapped around several times and with the consumer lagging."""
    import torch
    rng = np.random.RandomState(2)
    recs = []
    for i in range(6):
        n = 300 + 11 * i
        m, a = rng.randn(n, 219).astype(np.float32), rng.randn(2 * n, 35).astype(np.float32)
        recs.append(tfrecord.make_example({
            "motion_name": "m%d" % i, "motion_sequence": m.flatten(), "motion_sequence_shape": np.array(m.shape),
            "audio_name": "a%d" % i, "audio_sequence": a.flatten(), "audio_sequence_shape": np.array(a.shape)}))
    tfrecord.write_records(str(tmp_path / "aist_tfrecord-train-0"), recs)
    cfg = _dataset_cfg(str(tmp_path / "*_tfrecord-train*"))
    tc = protos.TrainConfig()
    tc.batch_size = 4
    for depth in (2, 0):
        host = inputs.create_input(tc, cfg, is_training=True, seed=9, prefetch_batches=0)
        dev = inputs.create_input(tc, cfg, is_training=True, seed=9, prefetch_batches=depth, device="cuda")
        held = []
        for step in range(14):
            h, d = next(host), next(dev)
            assert d["motion_name"] == h["motion_name"] and inputs._EVENT not in d
            for k in ("motion_input", "audio_input", "target"):
                assert d[k].is_cuda and torch.equal(d[k].cpu(), h[k]), (depth, step, k)
            held.append((d, h))
            if step % 5 == 4:
                torch.cuda.synchronize()
        for d, h in held:   # earlier batches were not overwritten by later copies
            assert torch.equal(d["audio_input"].cpu(), h["audio_input"])
