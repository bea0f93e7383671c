"""Sequence generators of the sequence-level trainers (reference utils/agent_utils.py:30-284):
the sampler and the beam-search baseline, same class names, constructor arguments and call
signatures; tensors are torch tensors on the ROCm device."""
import torch

from .. import ops
from ..beam_search.beam_search import (BeamSearchConfig, beam_search_step_batch, choose_top_k,
                                       create_initial_beam_state_batch,
                                       loc_optimal_beam_path_batch)
from .reinforce import RnnDecoderNetwork


def sequence_length_fn(sequences, sequence_mask, keepdims=False):
    """Number of tokens greater than the pad token per row (`:30-32`)."""
    flag = (sequences > sequence_mask).to(torch.int32)
    return flag.sum(dim=-1, keepdim=keepdims).to(torch.int32)


class SequenceSampler(object):
    """`:34-132`.  __call__((features, sequences)) -> int32 [B, T], start token first: one
    draw per row and step from the policy's softmax with the pad, start and eos columns
    masked, eos forced where the step reaches a row's target length and pad after it.

    The reference draws with `tf.random.categorical`; that random stream is TensorFlow's
    own, so here each step takes one uniform number per row - from `generator` (a
    torch.Generator on the device, default: torch's global one) or from
    `uniform_fn(time_step, batch_size)` -> float32 [B] in [0,1) - and inverts the cumulative
    distribution at it.  `log_prob_fn` other than log-softmax is applied first and the draw
    is made from its output."""

    def __init__(self,
                 action_network,
                 start_token,
                 eos_token,
                 pad_token=0,
                 log_prob_fn=None,
                 dtype=torch.int32,
                 generator=None,
                 uniform_fn=None):
        self.action_network = action_network
        self.start_token = start_token
        self.pad_token = pad_token
        self.eos_token = eos_token
        self.log_prob_fn = log_prob_fn
        self.data_type = dtype
        self.generator = generator
        self.uniform_fn = uniform_fn

    def _uniforms(self, time_step, batch_size, device):
        if self.uniform_fn is not None:
            return torch.as_tensor(self.uniform_fn(time_step, batch_size), dtype=torch.float32,
                                   device=device).contiguous()
        return torch.rand(batch_size, dtype=torch.float32, device=device, generator=self.generator)

    def __call__(self, inputs, **kwargs):
        features, sequences = inputs
        batch_size = sequences.shape[0]
        device = features.device
        start_col = torch.full((batch_size, 1), self.start_token, dtype=torch.int32, device=device)
        row_len = sequence_length_fn(sequences, self.pad_token).contiguous()
        samples = [start_col]
        state = self.action_network.reset_state(sequences, {"features": features}, **kwargs)
        dec_input = start_col[:, 0]
        for time_step in range(1, sequences.shape[1]):
            predictions, state = self.action_network((dec_input, {"features": features}, state),
                                                     **kwargs)
            predictions = predictions.reshape(batch_size, -1)
            if not predictions.is_contiguous() or predictions.dtype != torch.float32:
                predictions = predictions.contiguous().float()
            if self.log_prob_fn is not None:
                # a caller-supplied normalisation sees the masked logits, as at `:112-118` (the
                # draw below masks the same columns again, which leaves their zero mass at zero)
                for tok in (self.pad_token, self.start_token, self.eos_token):
                    ops.logit_masks(predictions, start_token=tok)
                predictions = self.log_prob_fn(predictions).contiguous().float()
            sample = ops.sample_categorical(predictions, self._uniforms(time_step, batch_size, device),
                                            step=time_step, row_len=row_len,
                                            start_token=self.start_token, eos_token=self.eos_token,
                                            pad_token=self.pad_token)
            samples.append(sample.reshape(-1, 1))
            dec_input = sample
        return torch.cat(samples, dim=-1).to(self.data_type)


class BaselineSequeneBeamSearcher(object):
    """`:134-262`.  __call__(features, sequences) -> int32 [B, steps + 1], start token first.

    features [B,L,D] are the encoded image features.  max_length > 0: max_length - 1 beam
    steps with the start token masked; max_length <= 0: as many steps as `sequences` has
    columns minus one, eos forced where the step reaches a row's target length and pad
    after it.  A decoder of this package behind RnnDecoderNetwork runs as one fused engine
    enqueue; any other actor_network callable runs the step loop on the operator API."""

    def __init__(self,
                 actor_network,
                 beam_width,
                 vocab_size,
                 max_length,
                 start_token,
                 eos_token,
                 pad_token=0):
        self.actor_network = actor_network
        self.beam_width = beam_width
        self.vocab_size = vocab_size
        self.max_length = max_length
        self.start_token = start_token
        self.eos_token = eos_token
        self.pad_token = pad_token
        self.use_engine = True       # False: always the operator-level loop
        self._engine = None

    # -- fused path ---------------------------------------------------------------------
    def _engine_for(self, decoder, features, steps):
        from ..engine import CaptionConfig, CaptionEngine, params_from_modules
        B, L, D = features.shape
        decoder._build(D, features.device)

        class _NoEncoder:                      # the searcher starts from encoded features
            class fc:
                kernel = torch.zeros(D, D, dtype=torch.float32, device=features.device)
                bias = torch.zeros(D, dtype=torch.float32, device=features.device)
        p = params_from_modules(_NoEncoder, decoder)
        cfg = CaptionConfig(B=B, W=self.beam_width, L=L, F=D, D=D, E=decoder.embedding_dim,
                            H=decoder.units, U=decoder.attention.units, V=self.vocab_size, T=steps,
                            start_id=self.start_token, eos_id=self.eos_token,
                            length_penalty_weight=1.0,
                            reset_state_from_feat=float(decoder.reset_state_from_feat),
                            tile_mode="reference", cell="gru" if hasattr(decoder, "gru") else "lstm")
        key = (tuple(sorted(cfg.__dict__.items())),
               tuple((k, v.data_ptr()) for k, v in sorted(p.items()) if not k.startswith("enc_")),
               str(features.device))
        if self._engine is None or self._engine[0] != key:
            if self._engine is not None:
                self._engine[1].close()
            self._engine = (key, CaptionEngine(cfg, p, device=features.device))
        return self._engine[1]

    def _fusable(self):
        from ..decoder.embedding import EmbeddingLayer
        net = self.actor_network
        dec = getattr(net, "rnn_decoder", None)
        return (self.use_engine and isinstance(net, RnnDecoderNetwork) and dec is not None and
                isinstance(getattr(dec, "embedding_layer", None), EmbeddingLayer) and
                (hasattr(dec, "lstm") or hasattr(dec, "gru")) and self.vocab_size == dec.vocab_size)

    def __call__(self, features, sequences, **kwargs):
        batch_size = features.shape[0]
        W, V = self.beam_width, self.vocab_size
        max_length = self.max_length
        forced = max_length <= 0
        if forced:
            max_length = sequences.shape[1]
        start_col = torch.full((batch_size, 1), self.start_token, dtype=torch.int32,
                               device=features.device)
        if max_length <= 1:
            return start_col
        features = features.contiguous()

        if self._fusable():
            eng = self._engine_for(self.actor_network.rnn_decoder, features, max_length - 1)
            seq_len = sequence_length_fn(sequences, self.pad_token) if forced else None
            tokens, _ = eng.search_baseline(features, seq_len=seq_len, pad_token=self.pad_token)
            return torch.cat([start_col, tokens], dim=1)

        # operator-level loop, any actor_network (`:155-262`)
        dec_input = torch.full((batch_size * W, 1), self.start_token, dtype=torch.int32,
                               device=features.device)
        features = features.repeat(W, 1, 1)                               # tf.tile, `:162`
        hidden = self.actor_network.reset_state(torch.zeros_like(dec_input),
                                                {'features': features}, **kwargs)
        bs_state = create_initial_beam_state_batch(batch_size, W, device=features.device)
        bs_config = BeamSearchConfig(beam_width=W, vocab_size=V, eos_token=self.eos_token,
                                     length_penalty_weight=1.0, choose_successors_fn=choose_top_k)
        row_len = None
        if forced:
            row_len = sequence_length_fn(sequences, self.pad_token).repeat(W).contiguous()  # `:183-184`
        bs_sequence = []
        for i in range(1, max_length):
            predictions, hidden = self.actor_network((dec_input, {'features': features}, hidden),
                                                     training=False)
            predictions = predictions.reshape(-1, V)
            if not predictions.is_contiguous() or predictions.dtype != torch.float32:
                predictions = predictions.contiguous().float()
            ops.logit_masks(predictions, start_token=self.start_token, row_len=row_len,
                            step_plus_1=i + 1, eos_token=self.eos_token, pad_token=self.pad_token)
            bs_out, bs_state = beam_search_step_batch(i, predictions.reshape(-1, W, V), bs_state,
                                                      bs_config)
            bs_sequence.append((bs_out, bs_state))
            dec_input = bs_out.predicted_ids.reshape(-1, 1)
        decode_sequence = torch.cat(list(loc_optimal_beam_path_batch(bs_sequence)), dim=-1)
        return torch.cat([start_col, decode_sequence.to(torch.int32)], dim=-1)


class SequenceBeamSearchSampler(object):
    """`:264-284`: the searcher in its target-length form (max_length 0), called with the
    (features, sequences) pair the samplers share."""

    def __init__(self,
                 actor_network,
                 beam_width,
                 vocab_size,
                 start_token,
                 eos_token,
                 pad_token=0,
                 dtype=torch.int32):
        self.eos_token = eos_token
        self.searcher = BaselineSequeneBeamSearcher(actor_network, beam_width, vocab_size, 0,
                                                    start_token, eos_token)

    def __call__(self, inputs, **kwargs):
        features, sequences = inputs
        return self.searcher(features, sequences, **kwargs)
