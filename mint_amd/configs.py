"""Programmatic pipeline configs (same message tree a fact_*.config text proto parses into)."""
from mint_amd import protos


def fact_config(motion=(120, 225, 800, 2, 10, 3072), audio=(240, 35, 800, 2, 10, 3072),
                cross=(800, 12, 10, 3072), out_dim=225, audio_feature_dim_in_config=False):
    """MultiModalModel message; tuples are (seq_len, feature_dim, hidden, layers, heads, ff)."""
    mm = protos.MultiModalModel()
    fm = mm.fact_model
    for name, c in (("audio", audio), ("motion", motion)):
        mod = fm.modality.add()
        mod.feature_name = name
        mod.sequence_length = c[0]
        if name == "motion" or audio_feature_dim_in_config:
            mod.feature_dim = c[1]
        t = mod.model.add().transformer
        t.hidden_size, t.num_hidden_layers, t.num_attention_heads, t.intermediate_size = c[2], c[3], c[4], c[5]
    cm = fm.cross_modal_model
    cm.modality_a, cm.modality_b = "motion", "audio"
    t = cm.transformer
    t.hidden_size, t.num_hidden_layers, t.num_attention_heads, t.intermediate_size = cross
    cm.output_layer.out_dim = out_dim
    return mm


def fact_v5_deeper_t10_cm12():
    """Model + train section equivalent to configs/fact_v5_deeper_t10_cm12.config of the reference:
    d=800, 10 heads, ff=3072 (proto default), 2+2+12 layers, seq 120/240, out 225; batch 32,
    manual-step LR 1e-4 -> 1e-5 @100k -> 1e-6 @150k."""
    pipe = protos.TrainEvalPipelineConfig()
    pipe.multi_modal_model.CopyFrom(fact_config())
    pipe.train_config.batch_size = 32
    ms = pipe.train_config.learning_rate.manual_step_learning_rate
    ms.initial_learning_rate = 1e-4
    ms.schedule.add(step=100000, learning_rate=1e-5)
    ms.schedule.add(step=150000, learning_rate=1e-6)
    pipe.eval_config.batch_size = 1
    return pipe


def tiny_fact():
    """BASELINE.json configs[0]: 2+2+2 layers, d=128, seq 32/64 (heads 4, ff 512 as fixed in SURVEY 8d)."""
    return fact_config(motion=(32, 225, 128, 2, 4, 512), audio=(64, 35, 128, 2, 4, 512),
                       cross=(128, 2, 4, 512))
