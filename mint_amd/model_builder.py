"""Build model from model config — the drop-in boundary (mint/core/model_builder.py:19-33)."""
from mint_amd import fact_model  # noqa: E402


def _build_fact_model(model_config, is_training):
    return fact_model.FACTModel(model_config.fact_model, is_training)


MODEL_BUILDER_MAP = {
    "fact_model": _build_fact_model,
}


def build(model_config, is_training):
    """Build model based on model_config (a MultiModalModel message)."""
    model_type = model_config.WhichOneof("model")
    build_func = MODEL_BUILDER_MAP[model_type]
    return build_func(model_config, is_training)
