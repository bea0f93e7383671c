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


_NEG_INF_FP32 = -1e9


def get_decoder_self_attention_bias(length, dtype=torch.float32, device=None):
    """[1, 1, length, length]: 0 on and below the diagonal, -1e9 above (reference
    attention/attention.py:160-180)."""
    valid = torch.tril(torch.ones(length, length, dtype=dtype, device=device))
    return (_NEG_INF_FP32 * (1.0 - valid)).reshape(1, 1, length, length)


class EinsumProjection:
    """The projections of MultiHeadAttention as flat matrices: kernel [in, heads*dim] (the
    EinsumDense kernel [in, heads, dim] flattened), bias [heads*dim]; glorot_uniform / zeros."""

    def __init__(self, out_dim, device=None):
        self.fc = Dense(out_dim, device=device)

    @property
    def kernel(self):
        return self.fc.kernel

    @property
    def bias(self):
        return self.fc.bias

    def build(self, in_dim, device):
        if not self.fc.built:
            self.fc.build(in_dim, device)


class MultiHeadAttention:
    """Multi-head dot-product attention (reference attention/attention.py:183-562):
    query [B,T,dim], value [B,S,dim] (key = value) -> [B,T,dim_query].

    Constructor arguments keep the reference's order (num_heads, key_dim, ...); note that the
    reference's decoder stack passes (hidden_size, num_heads) into them positionally
    (decoder/transform_decoder.py:65-72).  attention_mask: values broadcastable to [B,T,S]
    (leading singleton axes allowed), applied as (1 - mask) * -1e9 on the fp32 scores
    (`:461-474`).  The q / k|v / output projections run on the MFMA dense kernels, the
    score / softmax / combine step on one HIP kernel that reads the keys and values of a
    batch element once for all its query positions."""

    def __init__(self, num_heads, key_dim, value_dim=None, dropout=0.0, use_bias=True,
                 output_shape=None, attention_axes=None, device=None, **kwargs):
        if value_dim not in (None, key_dim):
            raise NotImplementedError("MultiHeadAttention: value_dim != key_dim is not built")
        if not use_bias or output_shape is not None or attention_axes is not None:
            raise NotImplementedError("MultiHeadAttention: only the defaults the decoder stack uses "
                                      "(use_bias, output to the query dim, one attention axis)")
        self._num_heads = num_heads
        self._key_dim = key_dim
        self._dropout = dropout
        self.device = device
        hd = num_heads * key_dim
        self._query_dense = EinsumProjection(hd, device)
        self._key_dense = EinsumProjection(hd, device)
        self._value_dense = EinsumProjection(hd, device)
        self._output_dense = None
        self._kv = None

    def _build(self, query, value):
        dev = query.device
        self._query_dense.build(query.shape[-1], dev)
        self._key_dense.build(value.shape[-1], dev)
        self._value_dense.build(value.shape[-1], dev)
        if self._output_dense is None:
            self._output_dense = Dense(query.shape[-1], device=self.device)
        if not self._output_dense.built:
            self._output_dense.build(self._num_heads * self._key_dim, dev)

    def set_weights(self, q, k, v, o, device=None):
        """Each argument is a (kernel, bias) pair in the flattened layouts."""
        for proj, (kern, bias) in ((self._query_dense, q), (self._key_dense, k), (self._value_dense, v)):
            proj.fc.set_weights(kern, bias, device)
        self._output_dense = Dense(o[0].shape[1], device=self.device)
        self._output_dense.set_weights(o[0], o[1], device)
        self._kv = None

    def _packed_kv(self):
        k, v = self._key_dense, self._value_dense
        key = tuple((t.data_ptr(), t._version) for t in (k.kernel, k.bias, v.kernel, v.bias))
        if self._kv is None or self._kv[0] != key:
            self._kv = (key, torch.cat([k.kernel, v.kernel], dim=1).contiguous(),
                        torch.cat([k.bias, v.bias]).contiguous())
        return self._kv[1], self._kv[2]

    def call(self, query, value, key=None, attention_mask=None, return_attention_scores=False,
             training=None):
        if key is not None and key is not value:
            raise NotImplementedError("MultiHeadAttention: a key tensor different from value is not built")
        if return_attention_scores:
            raise NotImplementedError("MultiHeadAttention: attention scores are not materialised")
        if training and self._dropout > 0.0:
            raise NotImplementedError("MultiHeadAttention: dropout in training mode is outside the "
                                      "inference engine")
        B, T, _ = query.shape
        S = value.shape[1]
        self._build(query, value)
        q = ops.dense(query.contiguous(), self._query_dense.kernel, self._query_dense.bias)
        wkv, bkv = self._packed_kv()
        kv = ops.dense(value.contiguous(), wkv, bkv)                       # [B,S,2*HD]
        mask = None
        if attention_mask is not None:
            m = attention_mask.to(torch.float32)
            while m.dim() > 3 and m.shape[0] == 1:
                m = m[0]
            if m.dim() == 4 and m.shape[1] == 1:                           # [B,1,T,S]
                m = m[:, 0]
            if m.dim() == 2:
                m = m.reshape(1, *m.shape)
            if m.dim() != 3 or m.shape[0] not in (1, B):
                raise ValueError("MultiHeadAttention: attention_mask must broadcast to [B,T,S] "
                                 "without a head axis")
            m = m.expand(m.shape[0], T, S).contiguous()
            mask = m[0] if m.shape[0] == 1 else m
        hd = self._num_heads * self._key_dim
        out = ops.mha_cross_step(q.reshape(B * T, hd), kv, B, T, T, 1, self._num_heads, self._key_dim,
                                 mask=mask)
        return self._output_dense(out).reshape(B, T, -1)

    __call__ = call
