"""Config surface of the FACT pipeline: the proto2 schema of mint/protos/*.proto restated as plain
Python message classes, plus a text-format parser, so `fact_*.config` files load without
protobuf/TF (the reference's generated *_pb2.py only import under the pure-python protobuf
backend, and `protoc` is not available).

Mirrors the subset of the protobuf message API the reference touches: attribute access with
proto defaults, `WhichOneof`, `HasField`, repeated `.add()` / `.append()`, `CopyFrom`.

Schema sources: mint/protos/model.proto:20-124, train.proto:20-88, dataset.proto:22-90,
eval.proto:20-44, preprocessor.proto:20-26, pipeline.proto:26-33.
"""
import copy
import re

_SCALARS = {"int32": int, "uint32": int, "int64": int, "float": float, "string": str, "bool": bool}


class _Repeated(list):
    def __init__(self, ftype):
        super().__init__()
        self._ftype = ftype

    def add(self, **kw):
        m = self._ftype(**kw)
        self.append(m)
        return m


class Message:
    """Base class; subclasses define FIELDS = {name: (type, label, default, oneof_group)}."""
    FIELDS = {}
    ENUMS = {}

    def __init__(self, **kw):
        object.__setattr__(self, "_values", {})
        for k, v in kw.items():
            setattr(self, k, v)

    # -- attribute protocol ------------------------------------------------------------------
    def __getattr__(self, name):
        fields = type(self).FIELDS
        if name not in fields:
            raise AttributeError("%s has no field %r" % (type(self).__name__, name))
        ftype, label, default, _ = fields[name]
        vals = self._values
        if name in vals:
            return vals[name]
        if label == "repeated":
            vals[name] = _Repeated(ftype) if isinstance(ftype, type) and issubclass(ftype, Message) else []
            return vals[name]
        if isinstance(ftype, type) and issubclass(ftype, Message):
            # reading an unset sub-message returns a default instance that attaches on mutation
            return _LazyChild(self, name, ftype)
        return default

    def __setattr__(self, name, value):
        fields = type(self).FIELDS
        if name not in fields:
            raise AttributeError("%s has no field %r" % (type(self).__name__, name))
        ftype, label, _, group = fields[name]
        if label != "repeated" and not (isinstance(ftype, type) and issubclass(ftype, Message)):
            value = self._coerce(name, ftype, value)
        self._set(name, value, group)

    def _set(self, name, value, group):
        if group:
            for other, (_, _, _, g) in type(self).FIELDS.items():
                if g == group and other != name:
                    self._values.pop(other, None)
        self._values[name] = value

    def _coerce(self, name, ftype, value):
        if isinstance(ftype, str) and ftype.startswith("enum:"):
            enum = type(self).ENUMS[ftype[5:]]
            if isinstance(value, str):
                if value not in enum:
                    raise ValueError("unknown enum value %s for %s" % (value, name))
                return enum[value]
            return int(value)
        py = _SCALARS[ftype]
        if py is bool and isinstance(value, str):
            return value.lower() in ("true", "1", "t")
        return py(value)

    # -- protobuf-like helpers ------------------------------------------------------------------
    def HasField(self, name):
        return name in self._values

    def WhichOneof(self, group):
        for name, (_, _, _, g) in type(self).FIELDS.items():
            if g == group and name in self._values:
                return name
        return None

    def CopyFrom(self, other):
        object.__setattr__(self, "_values", copy.deepcopy(other._values))

    def __deepcopy__(self, memo):
        m = type(self)()
        object.__setattr__(m, "_values", copy.deepcopy(self._values, memo))
        return m

    def __eq__(self, other):
        return type(self) is type(other) and self._values == other._values

    def __repr__(self):
        return "%s(%s)" % (type(self).__name__, MessageToString(self).replace("\n", " "))


class _LazyChild:
    """Default sub-message proxy: reads give defaults; the first write attaches it to the parent."""

    def __init__(self, parent, name, ftype):
        object.__setattr__(self, "_p", (parent, name, ftype))
        object.__setattr__(self, "_m", ftype())

    def _attach(self):
        parent, name, ftype = self._p
        if name not in parent._values:
            parent._set(name, self._m, type(parent).FIELDS[name][3])
        return parent._values[name]

    def __getattr__(self, k):
        parent, name, ftype = self._p
        target = parent._values.get(name, self._m)
        ftype_k = type(target).FIELDS.get(k)
        if k in ("CopyFrom", "_set") or (ftype_k and (
                ftype_k[1] == "repeated" or (isinstance(ftype_k[0], type) and issubclass(ftype_k[0], Message)))):
            target = self._attach()
        return getattr(target, k)

    def __setattr__(self, k, v):
        setattr(self._attach(), k, v)

    def __deepcopy__(self, memo):
        parent, name, ftype = self._p
        return copy.deepcopy(parent._values.get(name, self._m), memo)

    def __eq__(self, other):
        parent, name, ftype = self._p
        return parent._values.get(name, self._m) == other


def _msg(name, fields, enums=None):
    return type(name, (Message,), {"FIELDS": fields, "ENUMS": enums or {}})


# ---- preprocessor.proto ----------------------------------------------------------------------
FACTPreprocessor = _msg("FACTPreprocessor", {})
Preprocessor = _msg("Preprocessor", {"fact_preprocessor": (FACTPreprocessor, "optional", None, "preprocessor")})

# ---- model.proto -----------------------------------------------------------------------------
Transformer = _msg("Transformer", {
    "hidden_size": ("int32", "optional", 768, None),
    "num_hidden_layers": ("int32", "optional", 12, None),
    "num_attention_heads": ("int32", "optional", 12, None),
    "max_position_embeddings": ("int32", "optional", 512, None),
    "intermediate_size": ("int32", "optional", 3072, None),
    "hidden_act": ("string", "optional", "gelu", None),
    "hidden_dropout_prob": ("float", "optional", 0.1, None),
    "attention_probs_dropout_prob": ("float", "optional", 0.1, None),
    "initializer_range": ("float", "optional", 0.02, None),
    "masked_loss_type": ("string", "optional", "nce", None),
    "add_spatial_attention": ("bool", "optional", False, None),
    "sp_hidden_size": ("int32", "optional", 768, None),
    "sp_num_attention_heads": ("int32", "optional", 12, None),
    "sp_num_hidden_layers": ("int32", "optional", 12, None),
    "add_cls_token": ("bool", "optional", False, None),
    "weight_decay": ("float", "optional", 0.0, None),
})
MLP = _msg("MLP", {
    "initializer_range": ("float", "optional", 0.02, None),
    "hidden_act": ("string", "optional", "gelu", None),
    "out_dim": ("int32", "optional", 0, None),
})
ModalityInputConfig = _msg("ModalityInputConfig", {"use_look_ahead_mask": ("bool", "optional", False, None)})
ModalityPreprocessor = _msg("ModalityPreprocessor",
                            {"fact_preprocessor": (FACTPreprocessor, "optional", None, "preprocessor")})
ModalityModel = _msg("ModalityModel", {
    "transformer": (Transformer, "optional", None, "model"),
    "mlp": (MLP, "optional", None, "model"),
})
Modality = _msg("Modality", {
    "feature_name": ("string", "optional", "", None),
    "feature_dim": ("int32", "optional", 0, None),
    "sequence_length": ("int32", "optional", 0, None),
    "input_config": (ModalityInputConfig, "optional", None, None),
    "preprocessor": (ModalityPreprocessor, "repeated", None, None),
    "model": (ModalityModel, "repeated", None, None),
})
CrossModalModel = _msg("CrossModalModel", {
    "modality_a": ("string", "optional", "", None),
    "modality_b": ("string", "optional", "", None),
    "transformer": (Transformer, "optional", None, "model"),
    "mlp": (MLP, "optional", None, "model"),
    "cross_modal_concat_dim": ("enum:CrossModalConcatDim", "optional", 1, None),
    "output_layer": (MLP, "optional", None, None),
    "preprocess": ("enum:Preprocess", "optional", 0, None),
}, {"CrossModalConcatDim": {"DEFAULT_CONCAT": 0, "SEQUENCE_WISE": 1, "CHANNEL_WISE": 2},
    "Preprocess": {"DEFAULT_NONE": 0, "CONTRASTIVE": 1}})
CrossModalModel.CrossModalConcatDim = type("CrossModalConcatDim", (), dict(DEFAULT_CONCAT=0, SEQUENCE_WISE=1,
                                                                           CHANNEL_WISE=2))
FACTModel = _msg("FACTModel", {
    "modality": (Modality, "repeated", None, None),
    "cross_modal_model": (CrossModalModel, "optional", None, None),
    "fk_path": ("string", "optional", "", None),
})
MultiModalModel = _msg("MultiModalModel", {"fact_model": (FACTModel, "optional", None, "model")})

# ---- train.proto -----------------------------------------------------------------------------
ConstantLearningRate = _msg("ConstantLearningRate", {"learning_rate": ("float", "optional", 0.002, None)})
ExponentialDecayLearningRate = _msg("ExponentialDecayLearningRate", {
    "initial_learning_rate": ("float", "optional", 0.002, None),
    "decay_steps": ("uint32", "optional", 4000000, None),
    "decay_factor": ("float", "optional", 0.95, None),
    "staircase": ("bool", "optional", True, None),
    "burnin_learning_rate": ("float", "optional", 0.0, None),
    "burnin_steps": ("uint32", "optional", 0, None),
    "min_learning_rate": ("float", "optional", 0.0, None),
})
LearningRateSchedule = _msg("LearningRateSchedule", {
    "step": ("uint32", "optional", 0, None),
    "learning_rate": ("float", "optional", 0.002, None),
})
ManualStepLearningRate = _msg("ManualStepLearningRate", {
    "initial_learning_rate": ("float", "optional", 0.002, None),
    "schedule": (LearningRateSchedule, "repeated", None, None),
    "warmup": ("bool", "optional", False, None),
})
CosineDecayLearningRate = _msg("CosineDecayLearningRate", {
    "learning_rate_base": ("float", "optional", 0.002, None),
    "total_steps": ("uint32", "optional", 4000000, None),
    "warmup_learning_rate": ("float", "optional", 0.0002, None),
    "warmup_steps": ("uint32", "optional", 10000, None),
    "hold_base_rate_steps": ("uint32", "optional", 0, None),
})
LearningRate = _msg("LearningRate", {
    "constant_learning_rate": (ConstantLearningRate, "optional", None, "learning_rate"),
    "exponential_decay_learning_rate": (ExponentialDecayLearningRate, "optional", None, "learning_rate"),
    "manual_step_learning_rate": (ManualStepLearningRate, "optional", None, "learning_rate"),
    "cosine_decay_learning_rate": (CosineDecayLearningRate, "optional", None, "learning_rate"),
})
TrainConfig = _msg("TrainConfig", {
    "num_steps": ("int32", "optional", 10000, None),
    "batch_size": ("int32", "optional", 4, None),
    "use_bfloat16": ("bool", "optional", False, None),
    "learning_rate": (LearningRate, "optional", None, None),
    "grad_clip_norm": ("float", "optional", 1.0, None),
    "fine_tune_checkpoint": ("string", "optional", "", None),
    "fine_tune_checkpoint_type": ("enum:CheckpointType", "optional", 0, None),
}, {"CheckpointType": {"DEFAULT": 0}})

# ---- dataset.proto ---------------------------------------------------------------------------
GeneralModality = _msg("GeneralModality", {
    "feature_name": ("string", "optional", "", None),
    "dimension": ("int32", "optional", 0, None),
    "sample_rate": ("int32", "optional", 0, None),
    "resize": ("int32", "optional", 0, None),
    "crop_size": ("int32", "optional", 0, None),
})
DataModality = _msg("DataModality", {"general_modality": (GeneralModality, "optional", None, "modality")})
Dataset = _msg("Dataset", {
    "name": ("string", "optional", "", None),
    "data_files": ("string", "optional", "", None),
    "window_type": ("enum:WindowType", "optional", 0, None),
    "data_target_field": ("string", "optional", "", None),
    "create_bert_masks": ("bool", "optional", False, None),
    "bert_mask_type": ("enum:BERTMaskType", "optional", 0, None),
    "data_augmentation_options": (Preprocessor, "repeated", None, None),
    "sample_window": ("bool", "optional", True, None),
    "target_num_categories": ("int32", "optional", 0, None),
    "modality": (DataModality, "repeated", None, None),
    "input_length_sec": ("float", "optional", 0.0, None),
    "target_length_sec": ("float", "optional", 0.0, None),
    "target_shift_sec": ("float", "optional", 0.0, None),
    "length_threshold_sec": ("float", "optional", 0.0, None),
}, {"WindowType": {"DEFAULT_WINDOW": 0, "BEGINNING": 1, "CENTER": 2, "RANDOM": 3},
    "BERTMaskType": {"DEFAULT_MASK": 0, "CONTIGUOUS": 1}})

# ---- eval.proto ------------------------------------------------------------------------------
MotionPredictionMetrics = _msg("MotionPredictionMetrics", {
    "add_positional_metrics": ("bool", "optional", False, None),
    "pck_thresholds": ("float", "repeated", None, None),
})
MotionGenerationMetrics = _msg("MotionGenerationMetrics", {
    "pck_thresholds": ("float", "repeated", None, None),
    "num_joints": ("int32", "optional", 24, None),
})
EvalMetric = _msg("EvalMetric", {
    "motion_prediction_metrics": (MotionPredictionMetrics, "optional", None, "metric_oneof"),
    "motion_generation_metrics": (MotionGenerationMetrics, "optional", None, "metric_oneof"),
})
EvalConfig = _msg("EvalConfig", {
    "batch_size": ("int32", "optional", 4, None),
    "eval_metric": (EvalMetric, "optional", None, None),
})

# ---- pipeline.proto --------------------------------------------------------------------------
TrainEvalPipelineConfig = _msg("TrainEvalPipelineConfig", {
    "multi_modal_model": (MultiModalModel, "optional", None, None),
    "train_config": (TrainConfig, "optional", None, None),
    "train_dataset": (Dataset, "optional", None, None),
    "eval_config": (EvalConfig, "optional", None, None),
    "eval_dataset": (Dataset, "optional", None, None),
})


# ---- text format -----------------------------------------------------------------------------
_TOKEN = re.compile(r"""\s*(?:(\#[^\n]*)|([A-Za-z_][A-Za-z0-9_]*)|("(?:[^"\\]|\\.)*"|'(?:[^'\\]|\\.)*')|"""
                    r"""([-+]?(?:\d+\.?\d*(?:[eE][-+]?\d+)?|\.\d+(?:[eE][-+]?\d+)?)f?)|([{}:<>\[\],;]))""")


def _tokenize(text):
    pos, out = 0, []
    text = text.rstrip()
    while pos < len(text):
        m = _TOKEN.match(text, pos)
        if not m:
            if text[pos:].strip() == "":
                break
            raise ValueError("text-format parse error near %r" % text[pos:pos + 30])
        pos = m.end()
        if m.group(1):
            continue
        if m.group(2):
            out.append(("id", m.group(2)))
        elif m.group(3):
            s = m.group(3)[1:-1]
            out.append(("str", bytes(s, "utf-8").decode("unicode_escape")))
        elif m.group(4):
            out.append(("num", m.group(4).rstrip("f")))
        else:
            out.append(("sym", m.group(5)))
    return out


def _parse_fields(msg, toks, i, closer):
    fields = type(msg).FIELDS
    while i < len(toks):
        kind, val = toks[i]
        if kind == "sym" and val == closer:
            return i + 1
        if kind == "sym" and val in (",", ";"):
            i += 1
            continue
        if kind != "id":
            raise ValueError("expected field name, got %r" % (val,))
        name = val
        if name not in fields:
            raise ValueError('Message type "%s" has no field named "%s"' % (type(msg).__name__, name))
        ftype, label, _, group = fields[name]
        i += 1
        if i < len(toks) and toks[i] == ("sym", ":"):
            i += 1
        is_msg = isinstance(ftype, type) and issubclass(ftype, Message)
        if is_msg:
            if toks[i] not in (("sym", "{"), ("sym", "<")):
                raise ValueError("expected '{' after message field %s" % name)
            close = "}" if toks[i][1] == "{" else ">"
            if label == "repeated":
                child = getattr(msg, name).add()
            else:
                child = msg._values.get(name)
                if child is None:
                    child = ftype()
                    msg._set(name, child, group)
            i = _parse_fields(child, toks, i + 1, close)
        else:
            if toks[i] == ("sym", "["):  # repeated scalar list syntax
                i += 1
                while toks[i] != ("sym", "]"):
                    if toks[i][0] != "sym":
                        getattr(msg, name).append(msg._coerce(name, ftype, toks[i][1]))
                    i += 1
                i += 1
                continue
            v = msg._coerce(name, ftype, toks[i][1])
            i += 1
            if label == "repeated":
                getattr(msg, name).append(v)
            else:
                msg._set(name, v, group)
    if closer is not None:
        raise ValueError("unexpected end of text-format input")
    return i


def Merge(text, message):
    """google.protobuf.text_format.Merge equivalent for the messages above."""
    _parse_fields(message, _tokenize(text), 0, None)
    return message


def MessageToString(msg, indent=0):
    out = []
    pad = "  " * indent
    for name, (ftype, label, _, _) in type(msg).FIELDS.items():
        if name not in msg._values:
            continue
        vals = msg._values[name] if label == "repeated" else [msg._values[name]]
        for v in vals:
            if isinstance(v, Message):
                out.append("%s%s {\n%s%s}\n" % (pad, name, MessageToString(v, indent + 1), pad))
            elif isinstance(v, str):
                out.append('%s%s: "%s"\n' % (pad, name, v.replace("\\", "\\\\").replace('"', '\\"')))
            elif isinstance(v, bool):
                out.append("%s%s: %s\n" % (pad, name, "true" if v else "false"))
            elif isinstance(ftype, str) and ftype.startswith("enum:"):
                inv = {b: a for a, b in type(msg).ENUMS[ftype[5:]].items()}
                out.append("%s%s: %s\n" % (pad, name, inv.get(v, v)))
            else:
                out.append("%s%s: %s\n" % (pad, name, repr(v) if isinstance(v, float) else v))
    return "".join(out)
