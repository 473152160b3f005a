"""Pipeline-config loading. Mirrors mint/utils/config_util.py:22-50 (same function name, same
returned dict keys) on top of the protobuf-free message classes in mint_amd.protos."""
from mint_amd import protos


def get_configs_from_pipeline_file(pipeline_config_path, config_override=None):
    """Reads a TrainEvalPipelineConfig text proto; returns the dict the reference returns
    (keys `model`, `train_config`, `train_dataset`, `eval_config`, `eval_dataset`)."""
    pipeline_config = protos.TrainEvalPipelineConfig()
    with open(pipeline_config_path, "r") as f:
        protos.Merge(f.read(), pipeline_config)
    if config_override:
        protos.Merge(config_override, pipeline_config)
    return {
        "model": pipeline_config.multi_modal_model,
        "train_config": pipeline_config.train_config,
        "train_dataset": pipeline_config.train_dataset,
        "eval_config": pipeline_config.eval_config,
        "eval_dataset": pipeline_config.eval_dataset,
    }


def create_pipeline_proto_from_configs(configs):
    """Inverse of get_configs_from_pipeline_file (mint/utils/config_util.py:53-72)."""
    pipeline_config = protos.TrainEvalPipelineConfig()
    pipeline_config.multi_modal_model.CopyFrom(configs["model"])
    pipeline_config.train_config.CopyFrom(configs["train_config"])
    pipeline_config.train_dataset.CopyFrom(configs["train_dataset"])
    pipeline_config.eval_config.CopyFrom(configs["eval_config"])
    pipeline_config.eval_dataset.CopyFrom(configs["eval_dataset"])
    return pipeline_config


def save_pipeline_config(pipeline_config, directory):
    """mint/utils/config_util.py:75-89 (without its TF1 logging call)."""
    import os
    os.makedirs(directory, exist_ok=True)
    path = os.path.join(directory, "pipeline.config")
    with open(path, "w") as f:
        f.write(protos.MessageToString(pipeline_config))
    return path
