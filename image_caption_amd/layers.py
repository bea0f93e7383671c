"""Weight holders with Keras semantics (lazy build on first call, Keras layouts and
default initialisers).  They hold torch tensors; the arithmetic is in libicap."""
from __future__ import annotations

import numpy as np
import torch

from . import ops

_seed_counter = [0]


def default_device():
    return torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() \
        else torch.device("cpu")


def _rng():
    _seed_counter[0] += 1
    return np.random.default_rng(0x1CA9 + _seed_counter[0])


def glorot_uniform(fan_in, fan_out, shape, rng=None):
    lim = np.sqrt(6.0 / (fan_in + fan_out))
    return (rng or _rng()).uniform(-lim, lim, size=shape).astype(np.float32)


class Dense:
    """tf.keras.layers.Dense: kernel [in, units] (glorot_uniform), bias [units] (zeros)."""

    def __init__(self, units, name=None, device=None):
        self.units = units
        self.name = name
        self.device = device
        self.kernel = None
        self.bias = None

    def build(self, in_dim, device=None):
        dev = self.device or device or default_device()
        self.kernel = torch.from_numpy(glorot_uniform(in_dim, self.units, (in_dim, self.units))).to(dev)
        self.bias = torch.zeros(self.units, dtype=torch.float32, device=dev)

    @property
    def built(self):
        return self.kernel is not None

    def set_weights(self, kernel, bias, device=None):
        dev = self.device or device or default_device()
        self.kernel = torch.as_tensor(np.ascontiguousarray(kernel), dtype=torch.float32).to(dev).contiguous()
        self.bias = torch.as_tensor(np.ascontiguousarray(bias), dtype=torch.float32).to(dev).contiguous()
        if self.kernel.shape[1] != self.units or self.bias.shape != (self.units,):
            raise ValueError("Dense %s: weights do not match units=%d" % (self.name, self.units))

    def __call__(self, x, act=ops.ACT_NONE):
        if not self.built:
            self.build(x.shape[-1], x.device)
        return ops.dense(x, self.kernel, self.bias, act=act)


class Embedding:
    """tf.keras.layers.Embedding: table [vocab, dim], uniform(-0.05, 0.05)."""

    def __init__(self, vocab_size, embedding_dim, device=None):
        self.vocab_size = vocab_size
        self.embedding_dim = embedding_dim
        self.device = device
        self.table = None

    def build(self, device=None):
        dev = self.device or device or default_device()
        t = _rng().uniform(-0.05, 0.05, size=(self.vocab_size, self.embedding_dim)).astype(np.float32)
        self.table = torch.from_numpy(t).to(dev)

    @property
    def built(self):
        return self.table is not None

    def set_weights(self, table, device=None):
        dev = self.device or device or default_device()
        self.table = torch.as_tensor(np.ascontiguousarray(table), dtype=torch.float32).to(dev).contiguous()
        if tuple(self.table.shape) != (self.vocab_size, self.embedding_dim):
            raise ValueError("Embedding: table shape mismatch")

    def __call__(self, ids):
        if not self.built:
            self.build(ids.device)
        flat = ids.reshape(-1).to(torch.int32)
        out = ops.embedding(self.table, flat)
        return out.reshape(*ids.shape, self.embedding_dim)
