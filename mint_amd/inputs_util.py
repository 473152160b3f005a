"""Input windowing of the reference (mint/utils/inputs_util.py) on numpy arrays."""
import numpy as np


def get_modality_to_param_dict(dataset_config):
    """Creates a map from modality name to modality parameters (inputs_util.py:18-45)."""
    out = {}
    for modality in dataset_config.modality:
        kind = modality.WhichOneof("modality")
        if kind == "general_modality":
            m = modality.general_modality
            out[m.feature_name] = {
                "feature_dim": m.dimension,
                "input_length": int(dataset_config.input_length_sec * m.sample_rate),
                "target_length": int(dataset_config.target_length_sec * m.sample_rate),
                "target_shift": int(dataset_config.target_shift_sec * m.sample_rate),
                "sample_rate": m.sample_rate,
                "resize": m.resize,
                "crop_size": m.crop_size,
            }
        else:
            raise ValueError("Unknown modality type:", kind)
    return out


def fact_preprocessing(example, modality_to_params, is_training, rng=None):
    """Preprocess data for the FACT model (inputs_util.py:59-107): left-pad the motion features with
    6 zero columns (3-dim translation -> 9-dim), pick a random window start in training
    (start = 0 in eval), motion_input = [start, start+L_m), target = [start+shift, +L_t),
    audio_input = [start, start+L_a) in training and the WHOLE audio track in eval."""
    example = dict(example)
    motion = np.asarray(example.pop("motion_sequence"), dtype=np.float32)
    audio = np.asarray(example.pop("audio_sequence"), dtype=np.float32)
    mp, ap = modality_to_params["motion"], modality_to_params["audio"]
    # (the reference pads the whole sequence and then cuts windows; cutting first gives the same windows without
    # copying a multi-thousand-frame track per sample)
    pad = lambda w: np.pad(w, [[0, 0], [6, 0]])
    if is_training:
        window = max(mp["input_length"], mp["target_shift"] + mp["target_length"], ap["input_length"])
        hi = motion.shape[0] - window + 1
        if hi <= 0:
            raise ValueError("sequence of %d frames is shorter than the %d-frame window" % (motion.shape[0], window))
        rng = rng if rng is not None else np.random
        start = int(rng.randint(0, hi))
    else:
        start = 0
    example["motion_input"] = pad(motion[start:start + mp["input_length"]])
    if is_training:
        s = start + mp["target_shift"]
        example["target"] = pad(motion[s:s + mp["target_length"]])
        example["audio_input"] = audio[start:start + ap["input_length"]]
    else:
        example["audio_input"] = audio
    return example
