"""mint_amd — MI355X-native FACT (Full-Attention Cross-modal Transformer) engine.

Drop-in for the hot path of google-research/mint: `model_builder.build(model_config, is_training)`
returns an object with the reference FACTModel surface (mint/core/fact_model.py) whose math runs in
hand-written gfx950 HIP kernels behind the C ABI of include/fact_hip.h.
"""
__version__ = "0.1.0"
