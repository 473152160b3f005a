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

    def __rmul__(self, other):  # `python_list * tensor`, which TensorFlow converts like any other operand
        if isinstance(other, (list, tuple)):
            other = torch.as_tensor(other, dtype=torch.float64)
        return (other * self.as_subclass(torch.Tensor)).as_subclass(Tensor)


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


# ---- primitives used by mint/utils/inputs_util.py:fact_preprocessing and mint/core/learning_schedules.py ----
def _set_shape(self, shp):
    assert len(shp) == self.dim() and all(s is None or int(s) == int(d) for s, d in zip(shp, self.size())), \
        "set_shape(%s) on a tensor of shape %s" % (list(shp), list(self.size()))


Tensor.set_shape = _set_shape


def pad(x, paddings):
    flat = []
    for lo, hi in reversed([list(p) for p in paddings]):
        flat += [int(lo), int(hi)]
    return torch.nn.functional.pad(_raw(x), flat).as_subclass(Tensor)


def maximum(a, b):
    if isinstance(a, torch.Tensor) or isinstance(b, torch.Tensor):
        return torch.maximum(torch.as_tensor(_raw(a)), torch.as_tensor(_raw(b))).as_subclass(Tensor)
    return max(a, b)


def cast(x, dtype):
    return torch.as_tensor(_raw(x)).to(dtype).as_subclass(Tensor)


def greater_equal(a, b):
    return torch.as_tensor(_raw(a)) >= torch.as_tensor(b)


def where(cond, a, b):
    return torch.where(cond, torch.as_tensor(a), torch.as_tensor(b))


def reduce_max(x, axis=None):
    return torch.as_tensor(x).max()


def reduce_sum(x, axis=None, name=None):
    return torch.as_tensor(_raw(x)).sum().as_subclass(Tensor)


def one_hot(index, depth):
    return torch.nn.functional.one_hot(torch.as_tensor(index).long(), depth).to(torch.float64).as_subclass(Tensor)


class name_scope:  # noqa: N801
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        return self.name

    def __exit__(self, *a):
        return False


random = types.ModuleType("tensorflow.random")
random.forced = None  # tests set this to make the reference's random window start reproducible


def _uniform(shape, minval=0, maxval=None, dtype=None):
    assert list(shape) == [], "shim: scalar tf.random.uniform only"
    if random.forced is not None:
        v = int(random.forced)
        assert minval <= v < int(maxval)
        return v
    return int(torch.randint(int(minval), int(maxval), ()).item())


random.uniform = _uniform
sys.modules["tensorflow.random"] = random
