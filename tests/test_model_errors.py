"""Error behaviour of the host mirror at the drop-in boundary (no GPU needed): the reference's ValueError for modalities
of different width (mint/core/base_models.py:184-189), its NotImplementedError for anything but SEQUENCE_WISE concatenation
(:190-196), and a loud failure - not a CPU fallback - where no GPU exists."""
import pytest
import torch

from mint_amd import configs, model_builder, protos


def _mm():
    return configs.fact_v5_deeper_t10_cm12().multi_modal_model


def test_modalities_of_different_width_raise_like_the_reference():
    mm = _mm()
    audio = [m for m in mm.fact_model.modality if m.feature_name == "audio"][0]
    audio.model[0].transformer.hidden_size = 640
    model = model_builder.build(mm, True)
    with pytest.raises(ValueError, match="should be the same"):
        model.build(2, 225, 35)


@pytest.mark.parametrize("dim", ["CHANNEL_WISE", "DEFAULT_CONCAT"])
def test_only_sequence_wise_concat_is_supported_like_the_reference(dim):
    mm = _mm()
    mm.fact_model.cross_modal_model.cross_modal_concat_dim = getattr(protos.CrossModalModel.CrossModalConcatDim, dim)
    model = model_builder.build(mm, True)
    with pytest.raises(NotImplementedError, match="not supported"):
        model.build(2, 225, 35)


@pytest.mark.skipif(torch.cuda.is_available(), reason="needs a host WITHOUT a GPU")
def test_no_gpu_is_an_error_not_a_cpu_fallback():
    model = model_builder.build(_mm(), True)
    with pytest.raises(RuntimeError, match="no CPU path"):
        model.build(2, 225, 35)
    with pytest.raises(RuntimeError):  # (torch's own "No HIP GPUs are available" when the call moves its inputs first)
        model({"motion_input": torch.zeros(1, 120, 225), "audio_input": torch.zeros(1, 240, 35)})
