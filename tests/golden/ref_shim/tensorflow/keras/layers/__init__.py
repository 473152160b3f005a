"""tf.keras.layers subset (see tensorflow/__init__.py of this shim)."""
import math

import torch


def _T():
    import tensorflow as tf
    return tf.Tensor


class Layer:
    def __init__(self, *args, **kwargs):
        self.built = False
        self._weights = []

    def add_weight(self, name=None, shape=None, initializer=None, dtype=None, trainable=True):
        if callable(initializer):
            w = initializer(shape, dtype)
        else:
            w = torch.zeros(*shape, dtype=torch.float64)
        w = torch.as_tensor(w, dtype=torch.float64)
        self._weights.append((name, w))
        return w

    def build(self, input_shape):
        pass

    def __call__(self, *args, **kwargs):
        if not self.built:
            first = args[0]
            self.build(list(first.shape) if hasattr(first, "shape") else None)
            self.built = True
        out = self.call(*args, **kwargs)
        return out.as_subclass(_T()) if isinstance(out, torch.Tensor) else out

    def call(self, *args, **kwargs):
        raise NotImplementedError


class Dense(Layer):
    """y = activation(x @ kernel + bias); kernel [in, out] glorot-uniform, bias zeros (Keras defaults)."""

    def __init__(self, units, activation=None, use_bias=True, kernel_initializer=None, **kwargs):
        super().__init__()
        self.units, self.activation, self.use_bias = units, activation, use_bias
        self.kernel_initializer = kernel_initializer
        self.kernel = None
        self.bias = None

    def build(self, input_shape):
        fan_in = int(input_shape[-1])
        if callable(self.kernel_initializer):
            self.kernel = torch.as_tensor(self.kernel_initializer((fan_in, self.units), None), dtype=torch.float64)
        else:
            lim = math.sqrt(6.0 / (fan_in + self.units))
            self.kernel = (torch.rand(fan_in, self.units, dtype=torch.float64) * 2 - 1) * lim
        if self.use_bias:
            self.bias = torch.zeros(self.units, dtype=torch.float64)

    def call(self, x):
        y = x.as_subclass(torch.Tensor) @ self.kernel
        if self.use_bias:
            y = y + self.bias
        y = y.as_subclass(_T())
        if self.activation is not None:
            y = self.activation(y)
        return y


class LayerNormalization(Layer):
    """Normalises the last axis: (x - mean) / sqrt(var + epsilon) * gamma + beta, biased variance."""

    def __init__(self, axis=-1, epsilon=1e-3, **kwargs):
        super().__init__()
        self.epsilon = epsilon
        self.gamma = None
        self.beta = None

    def build(self, input_shape):
        c = int(input_shape[-1])
        self.gamma = torch.ones(c, dtype=torch.float64)
        self.beta = torch.zeros(c, dtype=torch.float64)

    def call(self, x):
        r = x.as_subclass(torch.Tensor)
        mu = r.mean(dim=-1, keepdim=True)
        var = ((r - mu) ** 2).mean(dim=-1, keepdim=True)
        return (r - mu) / torch.sqrt(var + self.epsilon) * self.gamma + self.beta
