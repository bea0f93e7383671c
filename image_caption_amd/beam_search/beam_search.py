"""Batched beam search with the reference's step-level interface
(reference beam_search/beam_search.py), plus the fused whole-sequence entry.

State and outputs are torch tensors on the ROCm device with the reference's dtypes:
log_probs/scores float32, finished bool, lengths / predicted_ids / beam_parent_ids
int32.  With the default successor choice one step is one fused HIP op (log-softmax,
finished mask, length penalty, top-k over W*V, gathers); a custom
`choose_successors_fn` runs between a scores kernel and a gather kernel instead."""
from __future__ import annotations

from collections import namedtuple

import torch

from .. import ops
from ..layers import default_device


class BeamSearchState(namedtuple("BeamSearchState", ["log_probs", "finished", "lengths"])):
    """State for a single step of beam search (reference `:26-35`)."""
    pass


class BeamSearchStepOutput(namedtuple("BeamSearchStepOutput",
                                      ["scores", "predicted_ids", "beam_parent_ids"])):
    """Outputs for a single step of beam search (reference `:38-49`)."""
    pass


class BeamSearchConfig(namedtuple("BeamSearchConfig", [
        "beam_width", "vocab_size", "eos_token", "length_penalty_weight",
        "choose_successors_fn"])):
    """Configuration object for beam search (reference `:52-68`)."""
    pass


def create_initial_beam_state(beam_width, device=None):
    """Zero state for the unbatched step (reference `:70-83`)."""
    dev = device or default_device()
    return BeamSearchState(
        log_probs=torch.zeros([beam_width], dtype=torch.float32, device=dev),
        finished=torch.zeros([beam_width], dtype=torch.bool, device=dev),
        lengths=torch.zeros([beam_width], dtype=torch.int32, device=dev))


def create_initial_beam_state_batch(batch_size, beam_width, device=None):
    """Zero state for the batched step (reference `:85-98`)."""
    dev = device or default_device()
    return BeamSearchState(
        log_probs=torch.zeros([batch_size, beam_width], dtype=torch.float32, device=dev),
        finished=torch.zeros([batch_size, beam_width], dtype=torch.bool, device=dev),
        lengths=torch.zeros([batch_size, beam_width], dtype=torch.int32, device=dev))


def combine_beam_state_batch(beam_state_left, beam_state_right, axis=-1):
    """Concatenate two states along the beam axis (reference `:100-113`)."""
    return BeamSearchState(
        log_probs=torch.cat([beam_state_left.log_probs, beam_state_right.log_probs], dim=axis),
        finished=torch.cat([beam_state_left.finished, beam_state_right.finished], dim=axis),
        lengths=torch.cat([beam_state_left.lengths, beam_state_right.lengths], dim=axis))


def choose_top_k(scores_flat, top_k):
    """Marker for the default successor choice (tf.nn.top_k: descending, lower index
    first on ties; reference `:144-148`).  `beam_search_step(_batch)` recognises this
    function and runs the fused kernel; called directly it runs the same kernel as a
    plain top-k over the last axis."""
    x = scores_flat.reshape(-1, 1, scores_flat.shape[-1]).contiguous()
    n = x.shape[0]
    dev = x.device
    zeros_f = torch.zeros(n, 1, dtype=torch.float32, device=dev)
    # time 0, zero log-prob, weight 0 (penalty exactly 1), input taken as log-probs:
    # score = (0 + x) / 1
    xw = x.expand(n, top_k, x.shape[-1]).contiguous() if top_k > 1 else x
    scores, ids, _, _, _, _ = ops.beam_step(
        xw, zeros_f.expand(n, top_k).contiguous(),
        torch.zeros(n, top_k, dtype=torch.uint8, device=dev),
        torch.zeros(n, top_k, dtype=torch.int32, device=dev),
        0, -1, 0.0, start_offset=0, input_is_log_prob=True)
    shape = tuple(scores_flat.shape[:-1]) + (top_k,)
    return scores.reshape(shape), ids.reshape(shape)


def length_penalty(sequence_lengths, penalty_factor):
    """(5 + len)^w / 6^w as float32 (reference `:115-128`)."""
    return ops.hyp_score(None, sequence_lengths.to(torch.int32).contiguous(), penalty_factor)


def hyp_score(log_probs, sequence_lengths, length_penalty_weight):
    """log_probs / length_penalty (reference `:131-141`); the operands broadcast."""
    lp, ln = torch.broadcast_tensors(log_probs, sequence_lengths)
    return ops.hyp_score(lp.to(torch.float32).contiguous(), ln.to(torch.int32).contiguous(),
                         length_penalty_weight)


def mask_probs(probs, eos_token, finished):
    """Finished beams put all probability mass on eos (reference `:165-197`):
    probs [..., V], finished [...]."""
    return ops.mask_probs(probs.to(torch.float32).contiguous(), eos_token,
                          _as_u8(finished).contiguous())


def batch_gather(a, b):
    """Per-row gather a[i, b[i, :]] (reference `:347-358`).  Like the reference, both
    operands go through float32 and the result is float32; an index outside the row yields 0,
    which is what the reference's tf.gather stores on a GPU (no host round trip to check)."""
    a32 = a.to(torch.float32).contiguous()
    idx = b.to(torch.float32).to(torch.int32).contiguous()
    return ops.batch_gather(a32, idx)


def _as_u8(finished):
    return finished.contiguous().view(torch.uint8) if finished.dtype == torch.bool \
        else finished.to(torch.uint8)


def beam_search_step_batch(time_, logits, beam_state, config, prob_fn=None, start_offset=0):
    """One batched beam-search step (reference `:360-543`).

    logits [B, R, V]; state tensors [B, R]; config.beam_width successors per image come
    out.  R is usually the beam width; the beam-search-optimisation trainer passes the gold
    row in front of the search rows (R = beam_width + 1, `start_offset=1`,
    utils/train_utils.py:275-279).  At time 0 only beam `start_offset` is scored.
    `prob_fn` None means log-softmax (the reference default, fused in the kernel); a custom
    callable is applied to the [B*R, V] logits first and its output is taken as
    log-probabilities."""
    B, Wl, V = logits.shape
    if V != config.vocab_size:
        raise ValueError("logits last dim %d != config.vocab_size %d" % (V, config.vocab_size))
    if tuple(beam_state.log_probs.shape) != (B, Wl):
        raise ValueError("beam state of shape %s for logits with %d x %d rows"
                         % (tuple(beam_state.log_probs.shape), B, Wl))
    time_ = int(time_)
    logits = logits.contiguous()
    is_log_prob = False
    if prob_fn is not None:
        logits = prob_fn(logits.reshape(-1, V)).reshape(B, Wl, V).contiguous()
        is_log_prob = True
    if config.choose_successors_fn not in (None, choose_top_k):
        return _step_with_custom_successors(time_, logits, beam_state, config, is_log_prob,
                                            start_offset)
    scores, ids, parents, nlp, nfin, nlen = ops.beam_step(
        logits, beam_state.log_probs.contiguous(), _as_u8(beam_state.finished),
        beam_state.lengths.contiguous(), time_, config.eos_token,
        config.length_penalty_weight, start_offset=start_offset, input_is_log_prob=is_log_prob,
        beam_width=config.beam_width)
    next_state = BeamSearchState(log_probs=nlp, lengths=nlen, finished=nfin.view(torch.bool))
    output = BeamSearchStepOutput(scores=scores, predicted_ids=ids, beam_parent_ids=parents)
    return output, next_state


def _step_with_custom_successors(time_, logits, beam_state, config, is_log_prob, start_offset):
    """The unfused step: scores kernel -> caller's choice -> gather kernel.  The callable
    gets the reference's `scores_flat` ([B, W*V], or [B, V] of beam `start_offset` at time
    0) and the beam width, and returns (scores [B,K], flat indices [B,K])."""
    B, Wl, V = logits.shape
    fin = _as_u8(beam_state.finished).contiguous()
    lengths = beam_state.lengths.contiguous()
    scores, totals = ops.beam_scores(logits, beam_state.log_probs.contiguous(), fin, lengths,
                                     config.eos_token, config.length_penalty_weight,
                                     input_is_log_prob=is_log_prob)
    scores_flat = scores.reshape(B, -1) if time_ > 0 else scores[:, start_offset, :]
    next_scores, word_indices = config.choose_successors_fn(scores_flat, config.beam_width)
    word_indices = torch.as_tensor(word_indices, device=logits.device).to(torch.int32)
    word_indices = word_indices.reshape(B, -1).contiguous()
    # (indices outside [0, W*V) are handled in the kernel the way the reference's gathers
    # handle them on a GPU: zeros, not an exception - no host round trip on the step path)
    ids, parents, nlp, nfin, nlen = ops.beam_gather(totals, fin, lengths, word_indices,
                                                    config.eos_token)
    next_state = BeamSearchState(log_probs=nlp, lengths=nlen, finished=nfin.view(torch.bool))
    output = BeamSearchStepOutput(scores=next_scores, predicted_ids=ids, beam_parent_ids=parents)
    return output, next_state


def beam_search_step(time_, logits, beam_state, config):
    """Unbatched step (reference `:200-298`): logits [W, V], state vectors [W]."""
    st = BeamSearchState(beam_state.log_probs.unsqueeze(0), beam_state.finished.unsqueeze(0),
                         beam_state.lengths.unsqueeze(0))
    fn = config.choose_successors_fn
    if fn not in (None, choose_top_k):
        # the caller's function sees the unbatched [W*V] (or [V]) scores of the reference
        def batched_fn(scores_flat, k, _fn=fn):
            sc, idx = _fn(scores_flat[0], k)
            return (torch.as_tensor(sc, device=scores_flat.device).unsqueeze(0),
                    torch.as_tensor(idx, device=scores_flat.device).unsqueeze(0))
        config = config._replace(choose_successors_fn=batched_fn)
    out, nst = beam_search_step_batch(time_, logits.unsqueeze(0), st, config)
    return (BeamSearchStepOutput(out.scores[0], out.predicted_ids[0], out.beam_parent_ids[0]),
            BeamSearchState(nst.log_probs[0], nst.finished[0], nst.lengths[0]))


def _backtrack(beam_sequence, batched):
    outs = [o for o, _ in beam_sequence]
    last_out, last_state = beam_sequence[-1]
    if batched:
        ids = torch.stack([o.predicted_ids for o in outs]).to(torch.int32).contiguous()
        parents = torch.stack([o.beam_parent_ids for o in outs]).to(torch.int32).contiguous()
        scores, fin = last_out.scores.contiguous(), _as_u8(last_state.finished)
    else:
        ids = torch.stack([o.predicted_ids for o in outs]).to(torch.int32).unsqueeze(1).contiguous()
        parents = torch.stack([o.beam_parent_ids for o in outs]).to(torch.int32).unsqueeze(1).contiguous()
        scores, fin = last_out.scores.unsqueeze(0).contiguous(), _as_u8(last_state.finished.unsqueeze(0))
    tokens, _, _ = ops.backtrack(ids, parents, scores, fin)
    return tokens                                            # [B, T] int32


def loc_optimal_beam_path(beam_sequence):
    """Best path of an unbatched search (reference `:300-345`): list of T scalar ids."""
    if len(beam_sequence) == 0:
        return []
    tokens = _backtrack(beam_sequence, batched=False)
    return [tokens[0, t] for t in range(tokens.shape[1])]


def loc_optimal_beam_path_batch(beam_sequence):
    """Best path per image (reference `:545-613`): list of T tensors [B, 1], float32 like
    the reference (its batch_gather casts to float32), time-major."""
    if len(beam_sequence) == 0:
        return []
    tokens = _backtrack(beam_sequence, batched=True).to(torch.float32)
    return [tokens[:, t:t + 1] for t in range(tokens.shape[1])]


def sequence_beam_search(*args, **kwargs):
    """Fused whole-sequence decode; see image_caption_amd.engine.sequence_beam_search."""
    from ..engine import sequence_beam_search as impl
    return impl(*args, **kwargs)
