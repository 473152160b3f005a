"""Minimal stand-in for the TensorFlow / Keras primitives the reference's FACT model code uses
(mint/core/{fact_model,base_models,base_model_util,multi_modal_model}.py), implemented on PyTorch-CPU
in float64.  TEST INFRASTRUCTURE ONLY: with this package first on sys.path the reference's own model
code (layer composition, attention scale, qkv split, concat order, loss slice, auto-regressive loop)
can be imported and executed in a container that has no TensorFlow; only the primitives below are ours:

  Dense            y = x @ kernel (+ bias), optional activation      (tf.keras.layers.Dense)
  LayerNormalization  last axis, biased variance, eps inside the sqrt  (tf.keras.layers.LayerNormalization)
  softmax / einsum / concat / reduce_mean / square / tanh / pow / shape

Used by tests/golden/make_reference_golden.py and tests/test_oracle_vs_reference.py.
"""
import sys
import types

import torch

__version__ = "2.16.0-shim"
float32 = torch.float64  # everything runs in float64 here; the name is what the reference asks for
float64 = torch.float64
int32 = torch.int32


class _Shape(list):
    def as_list(self):
        return list(self)

    @property
    def rank(self):
        return len(self)


class Tensor(torch.Tensor):
    """torch.Tensor whose .shape answers .as_list() like a tf.TensorShape."""

    @property
    def shape(self):
        return _Shape(super().shape)


def convert_to_tensor(x, dtype=None):
    t = torch.as_tensor(x)
    if t.is_floating_point():
        t = t.to(torch.float64)
    return t.as_subclass(Tensor)


constant = convert_to_tensor


def _raw(x):
    return x.as_subclass(torch.Tensor) if isinstance(x, torch.Tensor) else x


def einsum(eq, *ops):
    return torch.einsum(eq, *[_raw(o) for o in ops]).as_subclass(Tensor)


def concat(values, axis):
    return torch.cat([_raw(v) for v in values], dim=axis).as_subclass(Tensor)


def shape(x):
    return list(_raw(x).shape)


def reduce_mean(x, axis=None):
    r = _raw(x)
    return (r.mean() if axis is None else r.mean(dim=axis)).as_subclass(Tensor)


def square(x):
    return (_raw(x) ** 2).as_subclass(Tensor)


def tanh(x):
    return torch.tanh(_raw(x)).as_subclass(Tensor)


def pow(x, y):  # noqa: A001  (tf.pow)
    return torch.pow(_raw(x), y).as_subclass(Tensor)


def zeros(shape, dtype=None):
    return torch.zeros(*shape, dtype=torch.float64).as_subclass(Tensor)


def ones(shape, dtype=None):
    return torch.ones(*shape, dtype=torch.float64).as_subclass(Tensor)


nn = types.ModuleType("tensorflow.nn")


def _softmax(x, axis=-1):
    return torch.softmax(_raw(x), dim=axis).as_subclass(Tensor)


nn.softmax = _softmax
nn.relu = lambda x: torch.relu(_raw(x)).as_subclass(Tensor)
sys.modules["tensorflow.nn"] = nn

from . import keras  # noqa: E402,F401

# einops picks its backend by scanning sys.modules; with a module called "tensorflow" present it would try
# its TensorFlow backends first.  Rearrange layers of the reference receive this shim's Tensor (a torch
# tensor), so pin einops' torch backend for both tensor types.
try:
    import einops._backends as _eb
    _tb = _eb.TorchBackend()
    _eb._type2backend[Tensor] = _tb
    _eb._type2backend[torch.Tensor] = _tb
except Exception:  # pragma: no cover - einops layout changed: fail at first use instead
    pass
