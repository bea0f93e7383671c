import numpy as np
import torch

from .. import ops
from ..attention.attention import BahdanauAttention
from ..layers import Dense, default_device, glorot_uniform
from ..utils.tf_utils import get_kwargs
from .decoder import Decoder
from .embedding import EmbeddingLayer
from .init_state import FeatureProjState


class GruCellWeights:
    """Keras GRUCell variables (TF 2.3 defaults, reset_after=True): kernel [in,3H],
    recurrent_kernel [H,3H] (glorot_uniform as the decoder requests), bias [2,3H] zeros;
    gates z, r, h.  `packed()` is the four-column layout the step kernel multiplies."""

    def __init__(self, units, device=None):
        self.units = units
        self.device = device
        self.kernel = self.recurrent_kernel = self.bias = None
        self._packed = None

    @property
    def built(self):
        return self.kernel is not None

    def build(self, in_dim, device=None):
        dev = self.device or device or default_device()
        H = self.units
        self.kernel = torch.from_numpy(glorot_uniform(in_dim, 3 * H, (in_dim, 3 * H))).to(dev)
        self.recurrent_kernel = torch.from_numpy(glorot_uniform(H, 3 * H, (H, 3 * H))).to(dev)
        self.bias = torch.zeros(2, 3 * H, dtype=torch.float32, device=dev)
        self._packed = None

    def set_weights(self, kernel, recurrent_kernel, bias, device=None):
        dev = self.device or device or default_device()
        conv = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32).to(dev).contiguous()
        self.kernel, self.recurrent_kernel, self.bias = conv(kernel), conv(recurrent_kernel), conv(bias)
        self._packed = None

    def packed(self):
        key = tuple((t.data_ptr(), t._version) for t in (self.kernel, self.recurrent_kernel, self.bias))
        if self._packed is None or self._packed[0] != key:
            self._packed = (key, ops.gru_pack(self.kernel, self.recurrent_kernel, self.bias))
        return self._packed[1]


class RecurrentDecoder(Decoder):
    """One decode step of the GRU decoder (reference decoder/recurrent_decoder.py:9-135).

    call((dec_input, feat_d, hiddens), training=...) ->
        (logits[N,V], [h], attention_weights[N,L,1], context_vector[N,D])
    The state list holds one tensor h; the attention query (`:77`) and the beta gate
    sigmoid(fc3(h)) (`:90`) read it, the cell input is [context | embedding] (`:97`)."""

    def __init__(self,
                 embedding_layer,
                 embedding_dim,
                 units,
                 vocab_size,
                 reset_state_from_feat=0,
                 data_type=torch.float32,
                 attention_dropout=0.0,
                 recurrent_input_dropout=0.0,
                 recurrent_state_dropout=0.0,
                 recurrent_initializer='glorot_uniform',
                 fc1_dropout=0.0,
                 name='gru_decoder',
                 device=None):
        super(RecurrentDecoder, self).__init__(vocab_size, data_type, name=name)
        if data_type != torch.float32:
            raise NotImplementedError("RecurrentDecoder: only float32 module state is built")
        self.embedding_layer = embedding_layer
        self.embedding_dim = embedding_dim
        self.units = units
        self.vocab_size = vocab_size
        self.device = device
        self.gru = GruCellWeights(units, device=device)
        self.reset_state_from_feat = reset_state_from_feat
        self.fc1 = Dense(units, name='{}/fc1'.format(self.name), device=device)
        self.fc2 = Dense(vocab_size, name='{}/fc2'.format(self.name), device=device)
        self.fc1_dropout = fc1_dropout
        self.dropouts = (attention_dropout, recurrent_input_dropout, recurrent_state_dropout,
                         fc1_dropout)
        self.attention = BahdanauAttention(units, attention_dropout=attention_dropout,
                                           name='{}/attention'.format(self.name), device=device)
        self.fc3 = Dense(1, name='{}/fc3'.format(self.name), device=device)
        self.num_cells = 1
        self.feat_proj_state_vec = [
            FeatureProjState(units, '{}/recurrent_feat_proj_{}'.format(self.name, hidx), device=device)
            for hidx in range(self.num_cells)]
        self._fc3_b = None
        self._emb_proj = None
        self.args = get_kwargs()

    def _build(self, feat_dim, device):
        if not self.gru.built:
            self.gru.build(feat_dim + self.embedding_dim, device)
        for layer, in_dim in ((self.fc1, self.units), (self.fc2, self.units), (self.fc3, self.units),
                              (self.attention.W1, feat_dim), (self.attention.W2, self.units),
                              (self.attention.V, self.units)):
            if not layer.built:
                layer.build(in_dim, device)
        for fp in self.feat_proj_state_vec:
            if not fp.fc.built:
                fp.fc.build(feat_dim, device)

    def _fc3_bias(self):
        if self._fc3_b is None or self._fc3_b[0] != self.fc3.bias._version:
            self._fc3_b = (self.fc3.bias._version, float(self.fc3.bias.item()))
        return self._fc3_b[1]

    def _embedding_projection(self, feat_dim, kernel4):
        """[V,4H] = embedding table . packed kernel rows of the embedding input (the
        token-only part of the gate pre-activations), once per weight version."""
        table = self.embedding_layer.table
        key = (table.data_ptr(), table._version, kernel4.data_ptr())
        if self._emb_proj is None or self._emb_proj[0] != key:
            rows = kernel4[feat_dim:feat_dim + self.embedding_dim]
            self._emb_proj = (key, ops.dense(table, rows.contiguous(), None))
        return self._emb_proj[1]

    def call(self, inputs, **kwargs):
        dec_input, feat_d, hiddens = inputs
        features = feat_d['features']
        feature_bias = feat_d.get('bias')
        training = kwargs['training']          # KeyError when omitted, like the reference
        if training and any(d > 0.0 for d in self.dropouts):
            raise NotImplementedError("RecurrentDecoder: dropout in training mode is outside "
                                      "the inference engine")
        features = features.contiguous()
        self._build(features.shape[-1], features.device)
        ids = dec_input.reshape(-1)
        hidden = hiddens[0]

        context_vector, attention_weights = self.attention.attend(
            features, hidden, bias=feature_bias, gate_state=hidden,
            gate_w=self.fc3.kernel.reshape(-1), gate_b=self._fc3_bias())

        kernel4, recurrent4, bias4 = self.gru.packed()
        if isinstance(self.embedding_layer, EmbeddingLayer):
            state = ops.gru_step(context_vector, hidden, kernel4, recurrent4, bias4,
                                 table=self.embedding_layer.table, ids=ids.to(torch.int32),
                                 emb_proj=self._embedding_projection(features.shape[-1], kernel4))
        else:
            emb = self.embedding_layer(ids, **kwargs)
            emb = emb.reshape(-1, emb.shape[-1]).contiguous()
            state = ops.gru_step(context_vector, hidden, kernel4, recurrent4, bias4, emb=emb)
        output = self.fc1(state)
        output_v = self.fc2(output)
        return output_v, [state], attention_weights.unsqueeze(-1), context_vector

    def reset_state(self, target, feat_d, **kwargs):
        features = feat_d.get('features')
        feature_bias = feat_d.get('bias')
        batch_size = target.shape[0]
        if features is None:
            dev = self.device or (target.device if isinstance(target, torch.Tensor) else default_device())
            return [torch.zeros((batch_size, self.units), dtype=self.data_type, device=dev)]
        features = features.contiguous()
        self._build(features.shape[-1], features.device)
        feat_proj = self.feat_proj_state_vec[0](features, mask=feature_bias)
        # 0 + reset_state_from_feat * proj  (`:125`)
        return [ops.expand_rows(feat_proj, 1, scale=self.reset_state_from_feat)]

    def copy(self):
        return type(self)(**self.args)

    def create_variables(self):
        # (the reference's version passes a malformed input tuple, recurrent_decoder.py:132-135,
        # and cannot run; this one only builds the variables)
        dev = self.device or default_device()
        self._build(self.embedding_dim, dev)
        if isinstance(self.embedding_layer, EmbeddingLayer):
            self.embedding_layer.table  # noqa: B018  builds the table
        return None

    def load_params(self, p, device=None):
        dev = device or self.device or default_device()
        self.gru.set_weights(p["gru_kernel"], p["gru_recurrent_kernel"], p["gru_bias"], dev)
        self.fc1.set_weights(p["fc1_kernel"], p["fc1_bias"], dev)
        self.fc2.set_weights(p["fc2_kernel"], p["fc2_bias"], dev)
        self.fc3.set_weights(p["fc3_kernel"], p["fc3_bias"], dev)
        self.attention.W1.set_weights(p["att_W1_kernel"], p["att_W1_bias"], dev)
        self.attention.W2.set_weights(p["att_W2_kernel"], p["att_W2_bias"], dev)
        self.attention.V.set_weights(p["att_V_kernel"], p["att_V_bias"], dev)
        self.feat_proj_state_vec[0].fc.set_weights(p["init0_kernel"], p["init0_bias"], dev)
        if isinstance(self.embedding_layer, EmbeddingLayer) and "embedding" in p:
            self.embedding_layer.embedding.set_weights(p["embedding"], dev)
        self.attention._pf_key = None
        self._fc3_b = None
        self._emb_proj = None
        self.attention._bv = None
