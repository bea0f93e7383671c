import torch

from .. import ops
from ..attention.attention import MultiHeadAttention, get_decoder_self_attention_bias
from ..embedding.position_embedding import RelativePositionEmbedding
from ..layers import Dense, Embedding, default_device
from .decoder import Decoder
from .ffn_layer import FeedForwardNetwork


class LayerNormWeights:
    """tf.keras.layers.LayerNormalization(epsilon=1e-6) over the last axis: gamma ones, beta zeros."""

    def __init__(self, epsilon=1e-6):
        self.epsilon = epsilon
        self.gamma = self.beta = None

    def build(self, dim, device):
        if self.gamma is None:
            self.gamma = torch.ones(dim, dtype=torch.float32, device=device)
            self.beta = torch.zeros(dim, dtype=torch.float32, device=device)

    def set_weights(self, gamma, beta, device):
        self.gamma = torch.as_tensor(gamma, dtype=torch.float32).to(device).contiguous()
        self.beta = torch.as_tensor(beta, dtype=torch.float32).to(device).contiguous()

    def __call__(self, x):
        self.build(x.shape[-1], x.device)
        # (the kernel adds an optional residual in place first; none here, x is unchanged)
        return ops.residual_layernorm(x.contiguous(), None, self.gamma, self.beta, self.epsilon)


class PrePostProcessingWrapper:
    """LayerNorm on the FIRST argument only, the wrapped layer, dropout (identity at
    inference), residual (reference decoder/transform_decoder.py:9-42): further positional
    arguments reach the layer untouched - the self-attention layer therefore sees the
    normalised tensor as query and the raw one as value."""

    def __init__(self, layer, params):
        self.layer = layer
        self.params = params
        self.layer_norm = LayerNormWeights(1e-6)

    def call(self, x, *args, **kwargs):
        training = kwargs["training"]
        if training and self.params.get("layer_postprocess_dropout", 0.0) > 0.0:
            raise NotImplementedError("PrePostProcessingWrapper: dropout in training mode is "
                                      "outside the inference engine")
        y = self.layer_norm(x)
        y = self.layer(y, *args, **kwargs)
        return ops.add_(y.contiguous(), x.contiguous())        # x + y

    __call__ = call


class DecoderStack:
    """N x (self-attention, encoder-decoder attention, feed-forward) with pre-norm wrappers
    and an output LayerNorm (reference decoder/transform_decoder.py:45-151).  As there, the
    attention layers are built as MultiHeadAttention(hidden_size, num_heads) - positionally
    (num_heads=hidden_size, key_dim=num_heads) - and the additive self-attention bias is
    handed to them as `attention_mask`."""

    def __init__(self, params, device=None):
        self.params = params
        self.layers = []
        for _ in range(params["num_hidden_layers"]):
            self_attention_layer = MultiHeadAttention(params["hidden_size"], params["num_heads"],
                                                      dropout=params["attention_dropout"], device=device)
            enc_dec_attention_layer = MultiHeadAttention(params["hidden_size"], params["num_heads"],
                                                         dropout=params["attention_dropout"],
                                                         device=device)
            feed_forward_network = FeedForwardNetwork(params["hidden_size"], params["filter_size"],
                                                      params["relu_dropout"], device=device)
            self.layers.append([PrePostProcessingWrapper(self_attention_layer, params),
                                PrePostProcessingWrapper(enc_dec_attention_layer, params),
                                PrePostProcessingWrapper(feed_forward_network, params)])
        self.output_normalization = LayerNormWeights(1e-6)

    def call(self, decoder_inputs, encoder_outputs, decoder_self_attention_bias, attention_bias,
             training, decode_loop_step=None):
        for layer in self.layers:
            self_attention_layer, enc_dec_attention_layer, feed_forward_network = layer
            decoder_inputs = self_attention_layer(decoder_inputs, decoder_inputs,
                                                  attention_mask=decoder_self_attention_bias,
                                                  training=training)
            decoder_inputs = enc_dec_attention_layer(decoder_inputs, encoder_outputs,
                                                     attention_mask=attention_bias, training=training)
            decoder_inputs = feed_forward_network(decoder_inputs, training=training)
        return self.output_normalization(decoder_inputs)

    __call__ = call


class TransformDecoder(Decoder):
    """Transformer decoder step (reference decoder/transform_decoder.py:154-235).

    call((dec_input[N,1], feat_d, [prefix[N,len]]), training=...) ->
        (logits[N,V] of the last position, [prefix | dec_input], None, outputs[:, -1, :])
    Like the reference it re-runs the whole prefix on every call (no cache); the fused
    whole-sequence path with per-row key / value caches is image_caption_amd.tengine.
    params: hidden_size, num_heads, num_hidden_layers, filter_size and the three dropouts."""

    def __init__(self, params, vocab_size, data_type=torch.int32, device=None):
        super(TransformDecoder, self).__init__(vocab_size, data_type)
        hidden_size = params['hidden_size']
        self.params = params
        self.device = device
        self.embedding = Embedding(vocab_size, hidden_size, device=device)
        self.fc2 = Dense(vocab_size, device=device)
        self.decoder_stack = DecoderStack(params, device=device)
        self.position_embedding = RelativePositionEmbedding(hidden_size=hidden_size)

    def call(self, inputs, **kwargs):
        dec_input, feat_d, hidden = inputs
        features = feat_d['features']
        kwargs['training']                       # KeyError when omitted, like the reference
        prev_decoded = hidden[0]
        targets = torch.cat([prev_decoded, dec_input.reshape(-1, 1).to(prev_decoded.dtype)], dim=-1)
        return self.decode(targets, features, **kwargs)

    def decode(self, targets, encoder_outputs, attention_bias=None, **kwargs):
        training = kwargs['training']
        if training and self.params.get("layer_postprocess_dropout", 0.0) > 0.0:
            raise NotImplementedError("TransformDecoder: dropout in training mode is outside the "
                                      "inference engine")
        decoder_inputs = self.embedding(targets)
        length = decoder_inputs.shape[1]
        decoder_inputs = ops.add_(decoder_inputs.contiguous(), self.position_embedding(decoder_inputs))
        decoder_self_attention_bias = get_decoder_self_attention_bias(
            length, dtype=decoder_inputs.dtype, device=decoder_inputs.device)
        outputs = self.decoder_stack(decoder_inputs, encoder_outputs.contiguous(),
                                     decoder_self_attention_bias, attention_bias, training=training)
        logits = self.fc2(outputs)
        return logits[:, -1, :], [targets], None, outputs[:, -1, :]

    def reset_state(self, target, features):
        batch_size = target.shape[0]
        dev = self.device or (target.device if isinstance(target, torch.Tensor) else default_device())
        return [torch.zeros((batch_size, 1), dtype=self.data_type, device=dev)]

    # -- weights under the TransformerEngine's parameter names --
    def load_params(self, p, device=None):
        dev = device or self.device or default_device()
        self.embedding.set_weights(p["t_embedding"], dev)
        self.fc2.set_weights(p["t_fc2_kernel"], p["t_fc2_bias"], dev)
        self.decoder_stack.output_normalization.set_weights(p["out_ln_gamma"], p["out_ln_beta"], dev)
        for l, (sa, ca, ff) in enumerate(self.decoder_stack.layers):
            pre = "l%d_" % l
            for wrap, att in ((sa, "self_"), (ca, "cross_")):
                wrap.layer_norm.set_weights(p[pre + att + "ln_gamma"], p[pre + att + "ln_beta"], dev)
                wrap.layer.set_weights(*[(p[pre + att + n + "_kernel"], p[pre + att + n + "_bias"])
                                         for n in ("q", "k", "v", "o")], device=dev)
            ff.layer_norm.set_weights(p[pre + "ffn_ln_gamma"], p[pre + "ffn_ln_beta"], dev)
            ff.layer.filter_dense_layer.set_weights(p[pre + "ffn_filter_kernel"],
                                                    p[pre + "ffn_filter_bias"], dev)
            ff.layer.output_dense_layer.set_weights(p[pre + "ffn_output_kernel"],
                                                    p[pre + "ffn_output_bias"], dev)

    def export_params(self, encoder=None):
        """The weights under the TransformerEngine's names (after the first call has built
        them); with an MlpEncoder also `enc_kernel` / `enc_bias`."""
        p = {"t_embedding": self.embedding.table, "t_fc2_kernel": self.fc2.kernel,
             "t_fc2_bias": self.fc2.bias,
             "out_ln_gamma": self.decoder_stack.output_normalization.gamma,
             "out_ln_beta": self.decoder_stack.output_normalization.beta}
        for l, (sa, ca, ff) in enumerate(self.decoder_stack.layers):
            pre = "l%d_" % l
            for wrap, att in ((sa, "self_"), (ca, "cross_")):
                p[pre + att + "ln_gamma"], p[pre + att + "ln_beta"] = wrap.layer_norm.gamma, wrap.layer_norm.beta
                m = wrap.layer
                for n, d in (("q", m._query_dense), ("k", m._key_dense), ("v", m._value_dense),
                             ("o", m._output_dense)):
                    p[pre + att + n + "_kernel"], p[pre + att + n + "_bias"] = d.kernel, d.bias
            p[pre + "ffn_ln_gamma"], p[pre + "ffn_ln_beta"] = ff.layer_norm.gamma, ff.layer_norm.beta
            p[pre + "ffn_filter_kernel"] = ff.layer.filter_dense_layer.kernel
            p[pre + "ffn_filter_bias"] = ff.layer.filter_dense_layer.bias
            p[pre + "ffn_output_kernel"] = ff.layer.output_dense_layer.kernel
            p[pre + "ffn_output_bias"] = ff.layer.output_dense_layer.bias
        if encoder is not None:
            p["enc_kernel"], p["enc_bias"] = encoder.fc.kernel, encoder.fc.bias
        return p
