"""Config surface: text-proto parsing with proto defaults, oneofs and the protobuf-like API the
reference touches (mint/utils/config_util.py:22-50, mint/protos/*.proto)."""
import copy

import pytest

from mint_amd import config_util, configs, protos

FACT_TEXT = """
multi_modal_model {
  fact_model {
    modality: { feature_name: "audio" sequence_length: 240
      model: { transformer: { num_attention_heads: 10 hidden_size: 800 num_hidden_layers: 2 } } }
    modality: { feature_name: "motion" sequence_length: 120 feature_dim: 225
      model: { transformer: { num_attention_heads: 10 hidden_size: 800 num_hidden_layers: 2 } } }
    fk_path: "/tmp/x.pkl"   # unused by the model
    cross_modal_model: { modality_a: "motion" modality_b: "audio"
      transformer: { num_hidden_layers: 12 hidden_size: 800 num_attention_heads: 10 }
      output_layer: { out_dim: 225 } }
  }
}
train_dataset { name: "train" input_length_sec: 120.0 target_length_sec: 20 target_shift_sec: 120
  modality: { general_modality: { feature_name: "motion" dimension: 219 sample_rate: 1 } }
  data_augmentation_options { fact_preprocessor: { } }
  data_files: "./data/*_tfrecord-train*" }
train_config: { batch_size: 32
  learning_rate: { manual_step_learning_rate { initial_learning_rate: 1e-4
      schedule { step: 100000 learning_rate: 1e-5 } schedule { step: 150000 learning_rate: 1e-6 } } } }
eval_config: { batch_size: 1 }
"""


def test_parse_pipeline_text(tmp_path):
    path = tmp_path / "fact.config"
    path.write_text(FACT_TEXT)
    c = config_util.get_configs_from_pipeline_file(str(path))
    assert set(c) == {"model", "train_config", "train_dataset", "eval_config", "eval_dataset"}
    m = c["model"]
    assert m.WhichOneof("model") == "fact_model"
    audio, motion = m.fact_model.modality
    t = audio.model[0].transformer
    # intermediate_size falls back to the proto default 3072, audio feature_dim is unset (0)
    assert (audio.feature_name, audio.sequence_length, audio.feature_dim) == ("audio", 240, 0)
    assert (t.hidden_size, t.num_hidden_layers, t.num_attention_heads, t.intermediate_size) == (800, 2, 10, 3072)
    assert motion.feature_dim == 225 and motion.model[0].WhichOneof("model") == "transformer"
    cm = m.fact_model.cross_modal_model
    assert cm.cross_modal_concat_dim == protos.CrossModalModel.CrossModalConcatDim.SEQUENCE_WISE
    assert cm.output_layer.out_dim == 225 and abs(cm.output_layer.initializer_range - 0.02) < 1e-12
    tc = c["train_config"]
    assert tc.batch_size == 32 and tc.learning_rate.WhichOneof("learning_rate") == "manual_step_learning_rate"
    assert [s.step for s in tc.learning_rate.manual_step_learning_rate.schedule] == [100000, 150000]
    assert c["eval_config"].batch_size == 1 and c["train_dataset"].modality[0].general_modality.dimension == 219
    # override text merges on top (config_util.py:39-40)
    c2 = config_util.get_configs_from_pipeline_file(str(path), "train_config { batch_size: 8 }")
    assert c2["train_config"].batch_size == 8
    # round trip
    pipe = config_util.create_pipeline_proto_from_configs(c)
    out = config_util.save_pipeline_config(pipe, str(tmp_path / "out"))
    c3 = config_util.get_configs_from_pipeline_file(out)
    assert c3["model"] == c["model"] and c3["train_config"] == c["train_config"]


def test_message_api_like_protobuf():
    mm = protos.ModalityModel()
    assert mm.WhichOneof("model") is None
    assert mm.transformer.hidden_size == 768 and mm.WhichOneof("model") is None  # reads do not set
    mm.transformer.num_hidden_layers = 2
    assert mm.WhichOneof("model") == "transformer" and mm.HasField("transformer")
    mm.mlp.out_dim = 3  # oneof: setting the other member clears the first
    assert mm.WhichOneof("model") == "mlp" and not mm.HasField("transformer")
    f = protos.FACTModel()
    f.cross_modal_model.output_layer.out_dim = 225  # nested auto-vivification
    assert f.cross_modal_model.output_layer.out_dim == 225 and f.HasField("cross_modal_model")
    g = copy.deepcopy(f)
    g.cross_modal_model.output_layer.out_dim = 1
    assert f.cross_modal_model.output_layer.out_dim == 225
    with pytest.raises(AttributeError):
        f.no_such_field = 1
    with pytest.raises(ValueError):
        protos.Merge('bogus_field: 1', protos.TrainConfig())
    with pytest.raises(ValueError):
        protos.Merge('cross_modal_concat_dim: NOPE', protos.CrossModalModel())


def test_programmatic_configs_equal_text():
    def vals(t):  # value equality (protobuf equality distinguishes explicitly-set defaults)
        return (t.hidden_size, t.num_hidden_layers, t.num_attention_heads, t.intermediate_size)
    pipe = configs.fact_v5_deeper_t10_cm12()
    parsed = protos.Merge(FACT_TEXT, protos.TrainEvalPipelineConfig())
    a, b = pipe.multi_modal_model.fact_model, parsed.multi_modal_model.fact_model
    assert [m.feature_name for m in a.modality] == [m.feature_name for m in b.modality]
    for x, y in zip(a.modality, b.modality):
        assert x.sequence_length == y.sequence_length and x.feature_dim == y.feature_dim
        assert vals(x.model[0].transformer) == vals(y.model[0].transformer)
    assert vals(a.cross_modal_model.transformer) == vals(b.cross_modal_model.transformer)
    assert pipe.train_config.learning_rate == parsed.train_config.learning_rate


def test_shipped_reference_config_parses():
    """The reference's own configs/fact_v5_deeper_t10_cm12.config (text proto, TrainEvalPipelineConfig) through
    mint_amd.config_util; skipped where the reference checkout is absent (the GPU box)."""
    import os
    import pytest
    path = "/root/reference/configs/fact_v5_deeper_t10_cm12.config"
    if not os.path.exists(path):
        pytest.skip("reference checkout not present")
    from mint_amd import config_util
    cfgs = config_util.get_configs_from_pipeline_file(path)
    fm = cfgs["model"].fact_model
    mods = {m.feature_name: m for m in fm.modality}
    assert mods["motion"].sequence_length == 120 and mods["motion"].feature_dim == 225
    assert mods["audio"].sequence_length == 240
    for m in mods.values():
        t = m.model[0].transformer
        assert (t.hidden_size, t.num_hidden_layers, t.num_attention_heads, t.intermediate_size) == (800, 2, 10, 3072)
    ct = fm.cross_modal_model.transformer
    assert (ct.hidden_size, ct.num_hidden_layers, ct.num_attention_heads, ct.intermediate_size) == (800, 12, 10, 3072)
    assert fm.cross_modal_model.output_layer.out_dim == 225
    assert cfgs["train_config"].batch_size == 32
    lr = cfgs["train_config"].learning_rate.manual_step_learning_rate
    assert lr.initial_learning_rate == pytest.approx(1e-4)
    assert [(s.step, s.learning_rate) for s in lr.schedule] == [(100000, pytest.approx(1e-5)), (150000, pytest.approx(1e-6))]
