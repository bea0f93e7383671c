import torch

from .. import ops
from ..layers import Dense


class BahdanauAttention:
    """Additive attention (reference attention/attention.py:26-65).

    score = V(tanh(W1(features) + W2(hidden))) (+ bias); weights = softmax over the
    L axis; context = sum_L weights * features.  W1(features) does not depend on the
    decode step, so it is computed once per features tensor and cached; the
    tanh / V-dot / softmax / context reduction is one HIP kernel."""

    def __init__(self, units, attention_dropout=0.0, activation="tanh",
                 name="bahdanau_attention", device=None):
        if activation not in ("tanh", None) and getattr(activation, "__name__", "") != "tanh":
            raise NotImplementedError("BahdanauAttention: only tanh (the reference default) is built")
        self.name = name
        self.units = units
        self.W1 = Dense(units, name="{}/W1".format(name), device=device)
        self.W2 = Dense(units, name="{}/W2".format(name), device=device)
        self.V = Dense(1, name="{}/V".format(name), device=device)
        self.attention_dropout = attention_dropout
        self._pf_key = None
        self._pf = None
        self._bv = None

    def _feature_projection(self, features):
        key = (features.data_ptr(), features._version, tuple(features.shape),
               self.W1.kernel.data_ptr() if self.W1.built else 0,
               self.W1.kernel._version if self.W1.built else 0)
        if key != self._pf_key or self._pf is None:
            self._pf = self.W1(features)
            self._pf_key = (features.data_ptr(), features._version, tuple(features.shape),
                            self.W1.kernel.data_ptr(), self.W1.kernel._version)
        return self._pf

    def _scalar_bias(self):
        if self._bv is None or self._bv[0] != self.V.bias._version:
            self._bv = (self.V.bias._version, float(self.V.bias.item()))
        return self._bv[1]

    def attend(self, features, hidden, bias=None, gate_state=None, gate_w=None, gate_b=0.0):
        """(context[N,D], weights[N,L]); optional sigmoid gate fused on the context."""
        if not self.V.built:
            self.V.build(self.units, features.device)
        pf = self._feature_projection(features.contiguous())
        q = self.W2(hidden)
        return ops.attention(pf, features.contiguous(), q, self.V.kernel.reshape(-1),
                             self._scalar_bias(), score_bias=bias, gate_state=gate_state,
                             gate_w=gate_w, gate_b=gate_b, rows_per_group=1, row_stride_g=1,
                             row_stride_k=0)

    def call(self, features, hidden, training=False, bias=None):
        if training and self.attention_dropout > 0.0:
            raise NotImplementedError("BahdanauAttention: dropout in training mode is outside "
                                      "the inference engine")
        ctx, weights = self.attend(features, hidden, bias=bias)
        return ctx, weights.unsqueeze(-1)

    __call__ = call
