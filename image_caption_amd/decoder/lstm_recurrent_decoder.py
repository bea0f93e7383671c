import torch

from .. import ops
from ..attention.attention import BahdanauAttention
from ..layers import Dense, default_device, glorot_uniform
from ..utils.tf_utils import get_kwargs
from .decoder import Decoder
from .embedding import EmbeddingLayer
from .init_state import FeatureProjState

import numpy as np


class LstmCellWeights:
    """Keras LSTMCell variables: kernel [in,4H], recurrent_kernel [H,4H] (glorot_uniform
    as the decoder requests), bias [4H] zeros with the forget slice at 1; gates i,f,g,o."""

    def __init__(self, units, device=None):
        self.units = units
        self.device = device
        self.kernel = self.recurrent_kernel = self.bias = None

    @property
    def built(self):
        return self.kernel is not None

    def build(self, in_dim, device=None):
        dev = self.device or device or default_device()
        H = self.units
        self.kernel = torch.from_numpy(glorot_uniform(in_dim, 4 * H, (in_dim, 4 * H))).to(dev)
        self.recurrent_kernel = torch.from_numpy(glorot_uniform(H, 4 * H, (H, 4 * H))).to(dev)
        b = np.zeros(4 * H, np.float32)
        b[H:2 * H] = 1.0
        self.bias = torch.from_numpy(b).to(dev)

    def set_weights(self, kernel, recurrent_kernel, bias, device=None):
        dev = self.device or device or default_device()
        conv = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32).to(dev).contiguous()
        self.kernel, self.recurrent_kernel, self.bias = conv(kernel), conv(recurrent_kernel), conv(bias)


class LstmDecoder(Decoder):
    """One decode step of the show-attend-tell LSTM decoder
    (reference decoder/lstm_recurrent_decoder.py:10-140).

    call((dec_input, feat_d, hidden), training=...) ->
        (logits[N,V], [h, c], attention_weights[N,L,1], context_vector[N,D])
    hidden is the Keras state list [h, c]; the attention query and the beta gate use
    hidden[-1], i.e. the CELL state c (`:80`)."""

    def __init__(self,
                 embedding_layer,
                 embedding_dim,
                 units,
                 vocab_size,
                 reset_state_from_feat=0,
                 data_type=torch.float32,
                 reset_state_fn=None,
                 attention_dropout=0.0,
                 recurrent_input_dropout=0.0,
                 recurrent_state_dropout=0.0,
                 recurrent_initializer='glorot_uniform',
                 fc1_dropout=0.0,
                 name='lstm_decoder',
                 device=None):
        super(LstmDecoder, self).__init__(vocab_size, data_type, name=name)
        if data_type != torch.float32:
            raise NotImplementedError("LstmDecoder: only float32 module state is built")
        self.embedding_layer = embedding_layer
        self.embedding_dim = embedding_dim
        self.units = units
        self.vocab_size = vocab_size
        self.device = device
        self.lstm = LstmCellWeights(units, device=device)
        self.reset_state_from_feat = reset_state_from_feat
        self.fc1 = Dense(units, name='{}/fc1'.format(self.name), device=device)
        self.fc2 = Dense(vocab_size, name='{}/fc2'.format(self.name), device=device)
        self.fc1_dropout = fc1_dropout
        self.dropouts = (attention_dropout, recurrent_input_dropout, recurrent_state_dropout,
                         fc1_dropout)
        self.attention = BahdanauAttention(units, attention_dropout=attention_dropout,
                                           name='{}/attention'.format(self.name), device=device)
        self.fc3 = Dense(1, name='{}/fc3'.format(self.name), device=device)
        self.num_cells = 2
        self.feat_proj_state_vec = [
            FeatureProjState(units, '{}/lstmt_feat_proj_{}'.format(self.name, hidx), device=device)
            for hidx in range(self.num_cells)]
        self._fc3_b = None
        self._emb_proj = None
        self.args = get_kwargs()

    # -- lazily built variables (Keras builds on first call) --
    def _build(self, feat_dim, device):
        if not self.lstm.built:
            self.lstm.build(feat_dim + self.embedding_dim, device)
        if not self.fc1.built:
            self.fc1.build(self.units, device)
        if not self.fc2.built:
            self.fc2.build(self.units, device)
        if not self.fc3.built:
            self.fc3.build(self.units, device)
        att = self.attention
        if not att.W1.built:
            att.W1.build(feat_dim, device)
        if not att.W2.built:
            att.W2.build(self.units, device)
        if not att.V.built:
            att.V.build(self.units, device)
        for fp in self.feat_proj_state_vec:
            if not fp.fc.built:
                fp.fc.build(feat_dim, device)

    def _fc3_bias(self):
        if self._fc3_b is None or self._fc3_b[0] != self.fc3.bias._version:
            self._fc3_b = (self.fc3.bias._version, float(self.fc3.bias.item()))
        return self._fc3_b[1]

    def _embedding_projection(self, feat_dim):
        """[V,4H] = embedding table . lstm kernel rows of the embedding input: the part of
        the gate pre-activations that depends only on the token, computed once per weight
        version and gathered by id inside the LSTM kernel."""
        table = self.embedding_layer.table
        key = (table.data_ptr(), table._version, self.lstm.kernel.data_ptr(),
               self.lstm.kernel._version)
        if self._emb_proj is None or self._emb_proj[0] != key:
            rows = self.lstm.kernel[feat_dim:feat_dim + self.embedding_dim]
            self._emb_proj = (key, ops.dense(table, rows.contiguous(), None))
        return self._emb_proj[1]

    def call(self, inputs, **kwargs):
        dec_input, feat_d, hidden = inputs
        features = feat_d['features']
        feature_bias = feat_d.get('bias')
        training = kwargs['training']          # KeyError when omitted, like the reference
        if training and any(d > 0.0 for d in self.dropouts):
            raise NotImplementedError("LstmDecoder: dropout in training mode is outside the "
                                      "inference engine")
        features = features.contiguous()
        self._build(features.shape[-1], features.device)
        ids = dec_input.reshape(-1)
        h_prev, c_prev = hidden[0], hidden[-1]
        h_state = hidden[-1]                   # Keras state order [h, c]: this is c

        # attention over L with the beta gate sigmoid(fc3(c)) fused on the context
        context_vector, attention_weights = self.attention.attend(
            features, h_state, bias=feature_bias, gate_state=h_state,
            gate_w=self.fc3.kernel.reshape(-1), gate_b=self._fc3_bias())

        # [context | embedding | h] . [kernel ; recurrent_kernel] + bias, cell update
        if isinstance(self.embedding_layer, EmbeddingLayer):
            h_new, c_new = ops.lstm_step(context_vector, h_prev, c_prev, self.lstm.kernel,
                                         self.lstm.recurrent_kernel, self.lstm.bias,
                                         table=self.embedding_layer.table,
                                         ids=ids.to(torch.int32),
                                         emb_proj=self._embedding_projection(features.shape[-1]))
        else:
            emb = self.embedding_layer(ids, **kwargs)
            emb = emb.reshape(-1, emb.shape[-1]).contiguous()
            h_new, c_new = ops.lstm_step(context_vector, h_prev, c_prev, self.lstm.kernel,
                                         self.lstm.recurrent_kernel, self.lstm.bias, emb=emb)
        output = self.fc1(h_new)
        output_v = self.fc2(output)
        return output_v, [h_new, c_new], attention_weights.unsqueeze(-1), context_vector

    def reset_state(self, target, feat_d, **kwargs):
        features = feat_d.get('features')
        feature_bias = feat_d.get('bias')
        batch_size = target.shape[0]
        if features is None:
            dev = self.device or (target.device if isinstance(target, torch.Tensor) else default_device())
            zeros = torch.zeros((batch_size, self.units), dtype=self.data_type, device=dev)
            return [zeros] * self.num_cells
        features = features.contiguous()
        self._build(features.shape[-1], features.device)
        pooled = ops.pool_features(features, feature_bias)
        out_initial_state = []
        for h_idx in range(self.num_cells):
            feat_proj = self.feat_proj_state_vec[h_idx](features, mask=feature_bias, pooled=pooled)
            # 0 + reset_state_from_feat * proj  (`:129`)
            out_initial_state.append(ops.expand_rows(feat_proj, 1, scale=self.reset_state_from_feat))
        return out_initial_state

    def copy(self):
        return type(self)(**self.args)

    def create_variables(self):
        """One step on zero inputs of the smallest shape, which builds every variable, and
        returns that step's outputs like the reference (`:136-140`): (logits[1,V], [h,c],
        weights[1,1,1], context[1,embedding_dim])."""
        dev = self.device or default_device()
        dec_input = torch.zeros((1,), dtype=torch.int32, device=dev)
        feats = torch.zeros((1, 1, self.embedding_dim), dtype=torch.float32, device=dev)
        hidden = self.reset_state(dec_input, {})
        return self.call((dec_input, {"features": feats}, hidden), training=False)

    # -- weight import / export with the .npz container's parameter names --
    def load_params(self, p, device=None):
        dev = device or self.device or default_device()
        self.lstm.set_weights(p["lstm_kernel"], p["lstm_recurrent_kernel"], p["lstm_bias"], dev)
        self.fc1.set_weights(p["fc1_kernel"], p["fc1_bias"], dev)
        self.fc2.set_weights(p["fc2_kernel"], p["fc2_bias"], dev)
        self.fc3.set_weights(p["fc3_kernel"], p["fc3_bias"], dev)
        self.attention.W1.set_weights(p["att_W1_kernel"], p["att_W1_bias"], dev)
        self.attention.W2.set_weights(p["att_W2_kernel"], p["att_W2_bias"], dev)
        self.attention.V.set_weights(p["att_V_kernel"], p["att_V_bias"], dev)
        for idx, fp in enumerate(self.feat_proj_state_vec):
            fp.fc.set_weights(p["init%d_kernel" % idx], p["init%d_bias" % idx], dev)
        if isinstance(self.embedding_layer, EmbeddingLayer) and "embedding" in p:
            self.embedding_layer.embedding.set_weights(p["embedding"], dev)
        self.attention._pf_key = None
        self._fc3_b = None
        self._emb_proj = None
        self.attention._bv = None
