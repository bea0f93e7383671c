import math

import torch

from .. import ops
from ..layers import Dense


def _activation_name(activation):
    """'tanh' | 'linear' | 'relu' | 'sigmoid' for the activations the attention kernel applies
    inside the score, None for any other callable."""
    if activation is None or activation == "linear":
        return "linear"
    if isinstance(activation, str):
        if activation in ops.ATT_ACTIVATIONS:
            return activation
        raise ValueError("BahdanauAttention: unknown activation %r" % (activation,))
    if activation in (torch.tanh, math.tanh) or getattr(activation, "__name__", "") == "tanh":
        return "tanh"
    if activation in (torch.relu, torch.nn.functional.relu) or getattr(activation, "__name__", "") == "relu":
        return "relu"
    if activation is torch.sigmoid or getattr(activation, "__name__", "") == "sigmoid":
        return "sigmoid"
    return None


class BahdanauAttention:
    """Additive attention (reference attention/attention.py:26-65).

    score = V(tanh(W1(features) + W2(hidden))) (+ bias); weights = softmax over the
    L axis; context = sum_L weights * features.  W1(features) does not depend on the
    decode step, so it is computed once per features tensor and cached; the
    tanh / V-dot / softmax / context reduction is one HIP kernel.

    activation (`:31,47`): "tanh" (default), "relu", "sigmoid", "linear" / None or the torch
    functions of those names run inside the kernel; any other callable is applied to the
    [N, L, units] sum W1(features) + W2(hidden) as the caller's own tensor function and the
    kernel then runs with the identity in its place."""

    def __init__(self, units, attention_dropout=0.0, activation="tanh",
                 name="bahdanau_attention", device=None):
        self.activation = activation
        self._act_name = _activation_name(activation)
        if self._act_name is None and not callable(activation):
            raise TypeError("BahdanauAttention: activation must be a name or a callable")
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
        act = self._act_name
        if act is None:
            # a callable of the caller's: applied here, the kernel adds nothing and applies nothing
            pf = self.activation(pf + q.unsqueeze(1)).to(torch.float32).contiguous()
            if pf.shape != (q.shape[0], features.shape[1], self.units):
                raise ValueError("BahdanauAttention: the activation changed the shape of its input")
            q = torch.zeros_like(q)
            act = "linear"
        return ops.attention(pf, features.contiguous(), q, self.V.kernel.reshape(-1),
                             self._scalar_bias(), score_bias=bias, gate_state=gate_state,
                             gate_w=gate_w, gate_b=gate_b, rows_per_group=1, row_stride_g=1,
                             row_stride_k=0, activation=act)

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
    EinsumDense kernel [in, heads, dim] flattened), bias [heads*dim] or None (use_bias=False);
    glorot_uniform / zeros."""

    def __init__(self, out_dim, device=None, use_bias=True):
        self.fc = Dense(out_dim, device=device)
        self.use_bias = use_bias

    @property
    def kernel(self):
        return self.fc.kernel

    @property
    def bias(self):
        return self.fc.bias if self.use_bias else None

    def build(self, in_dim, device):
        if not self.fc.built:
            self.fc.build(in_dim, device)

    def __call__(self, x):
        return ops.dense(x.contiguous(), self.kernel, self.bias)


def _prod(xs):
    n = 1
    for x in xs:
        n *= int(x)
    return n


class MultiHeadAttention:
    """Multi-head dot-product attention (reference attention/attention.py:183-562):
    query [B,<query dims>,dim], value [B,<key dims>,dim], optional key -> [B,<query dims>,E].

    Constructor arguments keep the reference's order (num_heads, key_dim, ...); note that the
    reference's decoder stack passes (hidden_size, num_heads) into them positionally
    (decoder/transform_decoder.py:65-72).  attention_mask: values broadcastable to
    [B,<query dims>,<key dims>], applied as (1 - mask) * -1e9 on the fp32 scores (`:461-474`).

    The form the decoder stack uses (key = value, value_dim = key_dim, one attention axis, no
    scores returned) runs the q / k|v / output projections on the MFMA dense kernels and the
    score / softmax / combine step on one HIP kernel that reads the keys and values of a batch
    element once for all its query positions.  Everything else the reference's signature
    allows - a key tensor of its own, value_dim, output_shape, use_bias=False, attention_axes
    over inputs of any rank, return_attention_scores - takes the general kernels
    (ops.mha_scores / ops.mha_combine).  With several attention axes the soft-max runs over
    the last key axis only, as the reference's Softmax(axis=-1) does (`:457-458`)."""

    def __init__(self, num_heads, key_dim, value_dim=None, dropout=0.0, use_bias=True,
                 output_shape=None, attention_axes=None, device=None, **kwargs):
        self._num_heads = num_heads
        self._key_dim = key_dim
        self._value_dim = value_dim if value_dim else key_dim
        self._dropout = dropout
        self._use_bias = use_bias
        self._output_shape = output_shape
        if attention_axes is not None and not hasattr(attention_axes, "__len__"):
            attention_axes = (attention_axes,)
        self._attention_axes = None if attention_axes is None else tuple(attention_axes)
        self.device = device
        self._query_dense = EinsumProjection(num_heads * key_dim, device, use_bias)
        self._key_dense = EinsumProjection(num_heads * key_dim, device, use_bias)
        self._value_dense = EinsumProjection(num_heads * self._value_dim, device, use_bias)
        self._output_dense = None
        self._kv = None

    def _out_dims(self, query):
        if self._output_shape:
            o = self._output_shape
            return tuple(o) if hasattr(o, "__len__") else (int(o),)
        return (query.shape[-1],)

    def _build(self, query, value, key=None):
        dev = query.device
        self._query_dense.build(query.shape[-1], dev)
        self._key_dense.build((value if key is None else key).shape[-1], dev)
        self._value_dense.build(value.shape[-1], dev)
        if self._output_dense is None:
            self._output_dense = EinsumProjection(_prod(self._out_dims(query)), self.device, self._use_bias)
        self._output_dense.build(self._num_heads * self._value_dim, dev)

    def set_weights(self, q, k, v, o, device=None):
        """Each argument is a (kernel, bias) pair in the flattened layouts (bias ignored with
        use_bias=False)."""
        for proj, (kern, bias) in ((self._query_dense, q), (self._key_dense, k), (self._value_dense, v)):
            proj.fc.set_weights(kern, bias, device)
        self._output_dense = EinsumProjection(o[0].shape[1], self.device, self._use_bias)
        self._output_dense.fc.set_weights(o[0], o[1], device)
        self._kv = None

    def _packed_kv(self):
        k, v = self._key_dense, self._value_dense
        key = tuple((t.data_ptr(), t._version) for t in (k.kernel, k.bias, v.kernel, v.bias))
        if self._kv is None or self._kv[0] != key:
            self._kv = (key, torch.cat([k.kernel, v.kernel], dim=1).contiguous(),
                        torch.cat([k.bias, v.bias]).contiguous())
        return self._kv[1], self._kv[2]

    def _step_form(self, query, value, key, attention_mask, return_attention_scores):
        """True when the call is the form of the decoder stack (served by mha_cross_step)."""
        if return_attention_scores or (key is not None and key is not value):
            return False
        if self._value_dim != self._key_dim or not self._use_bias or query.dim() != 3 or value.dim() != 3:
            return False
        if self._attention_axes not in (None, (1,), (-3,)):
            return False
        if attention_mask is not None:
            m = attention_mask
            if m.dim() == 4 and m.shape[1] != 1:
                return False                      # a head axis of its own
            if m.dim() > 4:
                return False
        return True

    def call(self, query, value, key=None, attention_mask=None, return_attention_scores=False,
             training=None):
        if training and self._dropout > 0.0:
            raise NotImplementedError("MultiHeadAttention: dropout in training mode is outside the "
                                      "inference engine")
        self._build(query, value, key)
        if not self._step_form(query, value, key, attention_mask, return_attention_scores):
            return self._call_general(query, value, key, attention_mask, return_attention_scores)
        B, T, _ = query.shape
        S = value.shape[1]
        q = self._query_dense(query)
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
                raise ValueError("MultiHeadAttention: attention_mask must broadcast to [B,T,S]")
            m = m.expand(m.shape[0], T, S).contiguous()
            mask = m[0] if m.shape[0] == 1 else m
        hd = self._num_heads * self._key_dim
        out = ops.mha_cross_step(q.reshape(B * T, hd), kv, B, T, T, 1, self._num_heads, self._key_dim,
                                 mask=mask)
        return self._output_dense(out).reshape(B, T, *self._out_dims(query))

    def _call_general(self, query, value, key, attention_mask, return_attention_scores):
        """`call` for any rank and any of the constructor's options (`:526-562` with the einsum
        equations of `_build_attention_equation`, `:84-125`): the projected tensors have rank
        r = query rank + 1, layout (batch, <dims>, heads, channels); the attention axes are
        flattened into one query axis T and one key axis S, every other axis but heads and
        channels into one batch axis."""
        if key is None:
            key = value
        N, Dk, Dv = self._num_heads, self._key_dim, self._value_dim
        rank = query.dim() + 1
        if key.dim() != query.dim() or value.dim() != query.dim():
            raise ValueError("MultiHeadAttention: query, key and value must have the same rank")
        axes = self._attention_axes
        axes = tuple(range(1, rank - 2)) if axes is None else tuple(a % rank if a < 0 else a for a in axes)
        if any(a < 0 or a >= rank - 2 for a in axes) or len(set(axes)) != len(axes):
            raise ValueError("MultiHeadAttention: attention_axes %r do not fit rank-%d inputs"
                             % (self._attention_axes, rank - 1))
        batch_axes = tuple(i for i in range(rank - 2) if i not in axes)
        perm = batch_axes + axes + (rank - 2, rank - 1)

        def project(x, dense, d):
            y = dense(x.reshape(-1, x.shape[-1])).reshape(*x.shape[:-1], N, d)
            return y.permute(*perm)

        qp = project(query, self._query_dense, Dk)
        kp = project(key, self._key_dense, Dk)
        vp = project(value, self._value_dense, Dv)
        nb, na = len(batch_axes), len(axes)
        b_dims, t_dims, s_dims = tuple(qp.shape[:nb]), tuple(qp.shape[nb:nb + na]), tuple(kp.shape[nb:nb + na])
        if tuple(kp.shape[:nb]) != b_dims or tuple(vp.shape[:nb + na]) != b_dims + s_dims:
            raise ValueError("MultiHeadAttention: key / value shapes do not match the query's batch "
                             "axes or each other")
        Bf, T, S = _prod(b_dims), _prod(t_dims), _prod(s_dims)
        qf = qp.reshape(Bf, T, N, Dk).contiguous()
        kf = kp.reshape(Bf, S, N, Dk).contiguous()
        vf = vp.reshape(Bf, S, N, Dv).contiguous()
        mask = None
        if attention_mask is not None:
            # `_masked_softmax`: singleton axes are inserted in front of the attention axes
            # (where the heads axis sits) until the mask has the rank of the scores
            scores_shape = b_dims + (N,) + t_dims + s_dims
            m = attention_mask.to(torch.float32)
            while m.dim() < len(scores_shape):
                m = m.unsqueeze(-na * 2 - 1)
            mask = m.expand(*scores_shape).reshape(Bf, N, T, S).contiguous()
        probs = ops.mha_scores(qf, kf, mask=mask, softmax_segment=s_dims[-1] if na else 1)
        out = ops.mha_combine(probs, vf)                                   # [Bf,T,N,Dv]
        out = out.reshape(*b_dims, *t_dims, N, Dv)
        inv = [0] * rank
        for pos, ax in enumerate(perm):
            inv[ax] = pos
        out = out.permute(*inv).contiguous()                               # query layout, heads, Dv
        y = self._output_dense(out.reshape(-1, N * Dv)).reshape(*query.shape[:-1], *self._out_dims(query))
        if return_attention_scores:
            return y, probs.reshape(*b_dims, N, *t_dims, *s_dims)
        return y

    __call__ = call
