"""tf.keras subset: Model / Layer base with lazy build, Sequential, initializers, activations, metrics."""
import types

import torch

from . import layers  # noqa: F401
from .layers import Layer


class Model(Layer):
    pass


class Sequential(Model):
    def __init__(self, layers=None):
        super().__init__()
        self.layers = list(layers or [])

    def call(self, x):
        for layer in self.layers:
            x = layer(x)
        return x


initializers = types.ModuleType("tensorflow.keras.initializers")


class TruncatedNormal:
    """tf.keras.initializers.TruncatedNormal: N(0, stddev) re-drawn outside +-2 stddev."""

    def __init__(self, mean=0.0, stddev=0.05, seed=None):
        self.mean, self.stddev = mean, stddev

    def __call__(self, shape, dtype=None):
        t = torch.empty(*shape, dtype=torch.float64)
        torch.nn.init.trunc_normal_(t, mean=self.mean, std=self.stddev, a=self.mean - 2 * self.stddev,
                                    b=self.mean + 2 * self.stddev)
        return t


initializers.TruncatedNormal = TruncatedNormal

activations = types.ModuleType("tensorflow.keras.activations")
activations.relu = lambda x: torch.relu(x)

metrics = types.ModuleType("tensorflow.keras.metrics")


class Metric:
    def __init__(self, name=None, **kwargs):
        self.name = name


metrics.Metric = Metric

optimizers = types.ModuleType("tensorflow.keras.optimizers")
optimizers.schedules = types.ModuleType("tensorflow.keras.optimizers.schedules")


class LearningRateSchedule:
    def __init__(self, *a, **k):
        pass


optimizers.schedules.LearningRateSchedule = LearningRateSchedule
