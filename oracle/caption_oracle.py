"""CPU oracle for the show-attend-tell caption-decoding path.  TEST INFRASTRUCTURE ONLY.

This file is a numpy restatement of the reference's batched inference path.  It is
the *checker* for the HIP engine: only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import it.  Nothing under `image_caption_amd/`
imports it, and the product path raises when the HIP library is missing instead of
falling back to this code.

Why a restatement: the reference is Python on TensorFlow 2.3 eager
(`requirements.txt:1`), TensorFlow is not installed in the build container and no
reference file travels to the GPU box.  The arithmetic therefore follows the
reference source line by line plus the public contracts of the TF 2.3 / Keras ops it
calls (stated in each docstring).  Operation ORDER is preserved wherever it changes
fp32 rounding (two-step length penalty, `(x.K + h.R) + b` in the LSTM cell,
`fm*probs + (1-fm)*row` masking, un-hoisted `W1(features)`).

Pinning status
  * beam search step / back-tracking: PINNED by the reference's own golden vectors
    (`beam_search/beam_search_test.py:36-59` and `:61-108`), see
    `tests/test_oracle_golden.py`.
  * encoder, embedding, init-state, attention, LSTM step, logits, end-to-end loop:
    PARITY UNPINNED - the reference holds no test, fixture or golden vector for
    them.  They are cross-checked against independent torch-CPU primitives
    (`torch.nn.functional.linear/embedding/softmax`, `torch.nn.LSTMCell`) in
    `tests/test_oracle_crosscheck.py`, which guards the restatement against
    transcription slips but is not a reference-derived pin.

All functions take and return numpy arrays.  `dtype` defaults to float32 (the
reference's `tf.float32`); passing float64 weights/inputs gives a high-precision
"truth" run used by the tests to measure top-k margins.
"""

from __future__ import annotations

import math
from collections import namedtuple

import numpy as np

F32_MIN = np.float32(np.finfo(np.float32).min)  # tf.float32.min = -3.4028235e38


# ----------------------------------------------------------------------------------
# Keras layer contracts
# ----------------------------------------------------------------------------------

def dense(x, kernel, bias):
    """`tf.keras.layers.Dense` without activation: `x . kernel[in,out] + bias`,
    applied over the last axis for rank-3 inputs."""
    return np.matmul(x, kernel) + bias


def sigmoid(x):
    """`tf.nn.sigmoid` = 1 / (1 + exp(-x))."""
    one = x.dtype.type(1)
    return one / (one + np.exp(-x))


def softmax(x, axis):
    """`tf.nn.softmax`: exp(x - max) / sum(exp(x - max)) along `axis`."""
    m = np.max(x, axis=axis, keepdims=True)
    e = np.exp(x - m)
    return e / np.sum(e, axis=axis, keepdims=True)


def log_softmax(x):
    """`tf.nn.log_softmax` over the last axis: `(x - max) - log(sum(exp(x - max)))`."""
    shifted = x - np.max(x, axis=-1, keepdims=True)
    return shifted - np.log(np.sum(np.exp(shifted), axis=-1, keepdims=True))


# ----------------------------------------------------------------------------------
# encoder / embedding / init state / attention / decoder step
# ----------------------------------------------------------------------------------

def mlp_encoder(x, enc_kernel, enc_bias):
    """`MlpEncoder.call` (`encoder/mlp_encoder.py:16-21`): Dense(F->D), ReLU, dropout
    (identity with `training=False`).  x[B,L,F] -> features[B,L,D]."""
    return np.maximum(dense(x, enc_kernel, enc_bias), x.dtype.type(0))


def embedding_lookup(table, ids):
    """`EmbeddingLayer.call` (`decoder/embedding.py:9-10`): row gather."""
    return table[np.asarray(ids).reshape(-1)]


def feature_proj_state(features, kernel, bias, mask=None):
    """`FeatureProjState.__call__` (`decoder/init_state.py:10-36`).

    Unmasked: Dense(mean over L).  Masked: the masked SUM over L (never divided by
    the mask count, `:17-32`), zero where the mask row sums to zero."""
    if mask is None:
        pooled = np.mean(features, axis=-2, dtype=features.dtype)
    else:
        msum = np.sum(features * mask[..., None], axis=-2, dtype=features.dtype)
        cnt = np.sum(mask, axis=-1, keepdims=True)
        pooled = np.where(cnt == 0, np.zeros_like(msum), msum)
    return dense(pooled, kernel, bias)


def bahdanau_attention(features, hidden, p, bias=None, w1f=None, activation=np.tanh):
    """`BahdanauAttention.call` (`attention/attention.py:40-65`); `activation` is the
    constructor's argument (`:31,38,47`, default tanh).

    score = V(tanh(W1(features) + W2(hidden)[:,None,:])) (+ bias[:, :, None]);
    weights = softmax over the L axis; context = sum_L weights * features.
    Returns (context[N,D], weights[N,L,1]).  `w1f` lets a caller pass W1(features)
    computed once (it does not depend on the step); the reference recomputes it."""
    if w1f is None:
        w1f = dense(features, p["att_W1_kernel"], p["att_W1_bias"])        # [N,L,U]
    w2h = dense(hidden[:, None, :], p["att_W2_kernel"], p["att_W2_bias"])  # [N,1,U]
    act = activation(w1f + w2h)
    score = dense(act, p["att_V_kernel"], p["att_V_bias"])                 # [N,L,1]
    if bias is not None:
        score = score + bias[..., None]
    weights = softmax(score, axis=1)
    context = np.sum(weights * features, axis=1, dtype=features.dtype)
    return context, weights


def lstm_cell(x, h_tm1, c_tm1, kernel, recurrent_kernel, bias):
    """Keras `LSTMCell.call` (TF 2.3 default implementation=2, tanh / sigmoid):
    z = (x.kernel + h.recurrent_kernel) + bias; split 4 -> i,f,g,o;
    c' = sig(f)*c + sig(i)*tanh(g); h' = sig(o)*tanh(c').  Returns (h', [h', c'])."""
    z = np.matmul(x, kernel)
    z = z + np.matmul(h_tm1, recurrent_kernel)
    z = z + bias
    z0, z1, z2, z3 = np.split(z, 4, axis=1)
    i = sigmoid(z0)
    f = sigmoid(z1)
    c = f * c_tm1 + i * np.tanh(z2)
    o = sigmoid(z3)
    h = o * np.tanh(c)
    return h, [h, c]


def lstm_decoder_step(p, dec_input, features, hidden, bias=None, w1f=None):
    """`LstmDecoder.call` (`decoder/lstm_recurrent_decoder.py:59-116`).

    The attention query and the beta gate use `hidden[-1]`, which is the LSTM CELL
    state c (Keras state order is [h, c]; `:80`).  LSTM input is [context, embedding]
    in that order (`:100`).  fc1 has no activation (`:106`).
    Returns (logits[N,V], [h,c], weights[N,L,1], context[N,D] already times beta)."""
    ids = np.asarray(dec_input).reshape(-1)
    emb = embedding_lookup(p["embedding"], ids)
    h_state = hidden[-1]
    context, weights = bahdanau_attention(features, h_state, p, bias=bias, w1f=w1f)
    beta = sigmoid(dense(h_state, p["fc3_kernel"], p["fc3_bias"]))
    context = context * beta
    lstm_input = np.concatenate([context, emb], axis=-1)
    output, state = lstm_cell(lstm_input, hidden[0], hidden[1], p["lstm_kernel"],
                              p["lstm_recurrent_kernel"], p["lstm_bias"])
    output = dense(output, p["fc1_kernel"], p["fc1_bias"])
    logits = dense(output, p["fc2_kernel"], p["fc2_bias"])
    return logits, state, weights, context


def gru_cell(x, h_tm1, kernel, recurrent_kernel, bias):
    """Keras `GRUCell.call` (TF 2.3 defaults: reset_after=True, implementation=2, tanh /
    sigmoid; variables kernel [in,3H], recurrent_kernel [H,3H], bias [2,3H] = input bias,
    recurrent bias; gate order z, r, h):
    mx = x.kernel + bias[0]; mh = h.recurrent_kernel + bias[1];
    z = sig(mx_z + mh_z); r = sig(mx_r + mh_r); hh = tanh(mx_h + r*mh_h);
    h' = z*h + (1-z)*hh.  Returns (h', [h'])."""
    mx = np.matmul(x, kernel) + bias[0]
    mh = np.matmul(h_tm1, recurrent_kernel) + bias[1]
    x_z, x_r, x_h = np.split(mx, 3, axis=-1)
    r_z, r_r, r_h = np.split(mh, 3, axis=-1)
    z = sigmoid(x_z + r_z)
    r = sigmoid(x_r + r_r)
    hh = np.tanh(x_h + r * r_h)
    h = z * h_tm1 + (1 - z) * hh
    return h, [h]


def gru_decoder_step(p, dec_input, features, hidden, bias=None, w1f=None):
    """`RecurrentDecoder.call` (`decoder/recurrent_decoder.py:56-115`): the LSTM decoder's
    skeleton with one state h: the attention query (`:77`) and the beta gate (`:90`) read
    h, the cell is a GRUCell on [context, embedding] (`:97-99`), fc1 has no activation
    (`:103`).  Returns (logits[N,V], [h], weights[N,L,1], context[N,D] times beta)."""
    ids = np.asarray(dec_input).reshape(-1)
    emb = embedding_lookup(p["embedding"], ids)
    h_state = hidden[0]
    context, weights = bahdanau_attention(features, h_state, p, bias=bias, w1f=w1f)
    beta = sigmoid(dense(h_state, p["fc3_kernel"], p["fc3_bias"]))
    context = context * beta
    gru_input = np.concatenate([context, emb], axis=-1)
    output, state = gru_cell(gru_input, h_state, p["gru_kernel"], p["gru_recurrent_kernel"],
                             p["gru_bias"])
    output = dense(output, p["fc1_kernel"], p["fc1_bias"])
    logits = dense(output, p["fc2_kernel"], p["fc2_bias"])
    return logits, state, weights, context


def reset_state_gru(p, batch_size, features, reset_state_from_feat=1, bias=None):
    """`RecurrentDecoder.reset_state` (`decoder/recurrent_decoder.py:117-127`): one state,
    h = 0 + reset_state_from_feat * FeatureProjState_0(features)."""
    units = p["gru_recurrent_kernel"].shape[0]
    dt = p["gru_kernel"].dtype
    zeros = np.zeros((batch_size, units), dtype=dt)
    if features is None:
        return [zeros]
    proj = feature_proj_state(features, p["init0_kernel"], p["init0_bias"], mask=bias)
    return [zeros + dt.type(reset_state_from_feat) * proj]


def reset_state(p, batch_size, features, reset_state_from_feat=1, bias=None):
    """`LstmDecoder.reset_state` (`decoder/lstm_recurrent_decoder.py:118-131`):
    state_idx = 0 + reset_state_from_feat * FeatureProjState_idx(features)."""
    units = p["lstm_recurrent_kernel"].shape[0]
    dt = p["lstm_kernel"].dtype
    zeros = np.zeros((batch_size, units), dtype=dt)
    if features is None:
        return [zeros, zeros.copy()]
    out = []
    for idx in range(2):
        proj = feature_proj_state(features, p["init%d_kernel" % idx],
                                  p["init%d_bias" % idx], mask=bias)
        out.append(zeros + dt.type(reset_state_from_feat) * proj)
    return out


# ----------------------------------------------------------------------------------
# beam search (beam_search/beam_search.py)
# ----------------------------------------------------------------------------------

BeamSearchState = namedtuple("BeamSearchState", ["log_probs", "finished", "lengths"])
BeamSearchStepOutput = namedtuple("BeamSearchStepOutput",
                                  ["scores", "predicted_ids", "beam_parent_ids"])
BeamSearchConfig = namedtuple("BeamSearchConfig", [
    "beam_width", "vocab_size", "eos_token", "length_penalty_weight",
    "choose_successors_fn"])


def create_initial_beam_state(beam_width, dtype=np.float32):
    """`beam_search.py:70-83`."""
    return BeamSearchState(np.zeros([beam_width], dtype),
                           np.zeros([beam_width], bool),
                           np.zeros([beam_width], np.int32))


def create_initial_beam_state_batch(batch_size, beam_width, dtype=np.float32):
    """`beam_search.py:85-98`."""
    return BeamSearchState(np.zeros([batch_size, beam_width], dtype),
                           np.zeros([batch_size, beam_width], bool),
                           np.zeros([batch_size, beam_width], np.int32))


def length_penalty(sequence_lengths, penalty_factor, dtype=np.float32):
    """`beam_search.py:115-128`: `(5 + len)^w / 6^w`.  The numerator is an fp32
    `tf.pow`; the denominator `(5. + 1.) ** w` is a Python double folded to the
    tensor dtype before the division."""
    dt = np.dtype(dtype).type
    num = np.power(dt(5.0) + sequence_lengths.astype(dtype), dt(penalty_factor))
    den = dt((5.0 + 1.0) ** penalty_factor)
    return num / den


def hyp_score(log_probs, sequence_lengths, length_penalty_weight):
    """`beam_search.py:131-141`: log_probs / length_penalty."""
    return log_probs / length_penalty(sequence_lengths, length_penalty_weight,
                                      dtype=log_probs.dtype)


def choose_top_k(scores_flat, top_k):
    """`beam_search.py:144-148` -> `tf.nn.top_k`: values sorted descending; among
    equal values the lower index comes first.  Works on the last axis."""
    order = np.argsort(-scores_flat, axis=-1, kind="stable")[..., :top_k]
    return np.take_along_axis(scores_flat, order, axis=-1), order.astype(np.int32)


def mask_probs(probs, eos_token, finished):
    """`beam_search.py:165-197`: rows of finished beams become
    [f32.min ... 0@eos ... f32.min]; computed as `(1-fm)*row + fm*probs`."""
    dt = probs.dtype.type
    vocab_size = probs.shape[-1]
    fm = (dt(1.0) - finished.astype(probs.dtype))[..., None]
    non_finished = fm * probs
    row = np.full([vocab_size], dt(F32_MIN), dtype=probs.dtype)
    row[eos_token] = 0
    with np.errstate(over="ignore", invalid="ignore"):
        finished_examples = (dt(1.0) - fm) * row
        return finished_examples + non_finished


def batch_gather(a, b):
    """`beam_search.py:347-358`: per-row gather `a[i, b[i, :]]`; both operands
    round-trip through float32 and the result is float32."""
    a32 = np.asarray(a).astype(np.float32)
    idx = np.asarray(b).astype(np.float32).astype(np.int32)
    return np.take_along_axis(a32, idx, axis=-1)


def beam_search_step(time_, logits, beam_state, config):
    """Unbatched step, `beam_search.py:200-298`.  logits[W,V]."""
    W, V = config.beam_width, config.vocab_size
    eos = config.eos_token
    probs = log_softmax(logits)
    probs = mask_probs(probs, eos, beam_state.finished)
    with np.errstate(over="ignore", invalid="ignore"):
        total = beam_state.log_probs[:, None] + probs
        lengths_to_add = np.ones([W, V], np.int32)
        lengths_to_add[:, eos] = 0                       # tf.one_hot(.., V, 0, 1)
        add_mask = 1 - beam_state.finished.astype(np.int32)
        new_len = beam_state.lengths[:, None] + add_mask[:, None] * lengths_to_add
        scores = hyp_score(total, new_len, config.length_penalty_weight)
    scores_flat = scores.reshape(-1) if int(time_) > 0 else scores[0]
    fn = config.choose_successors_fn or choose_top_k
    next_scores, word_indices = fn(scores_flat, W)
    word_indices = np.asarray(word_indices).astype(np.int32)
    next_probs = total.reshape(-1)[word_indices]
    next_word = word_indices % V
    next_beam = word_indices // V
    next_finished = np.logical_or(beam_state.finished[next_beam], next_word == eos)
    add = (next_word != eos).astype(np.int32) * (1 - next_finished.astype(np.int32))
    next_len = beam_state.lengths[next_beam] + add
    return (BeamSearchStepOutput(next_scores, next_word, next_beam),
            BeamSearchState(next_probs, next_finished, next_len))


def beam_search_step_batch(time_, logits, beam_state, config, prob_fn=None,
                           start_offset=0):
    """Batched step, `beam_search.py:360-543`.  logits[B,W,V]; state arrays [B,W].

    At `time_ == 0` only beam `start_offset` is scored (`:454-455`); the chosen
    indices are then in [0,V) so the parent is 0 and `next_log_probs` is gathered
    from beam 0's totals (`:475-477`) whatever `start_offset` is."""
    B = logits.shape[0]
    Wl = logits.shape[1]
    W, V = config.beam_width, config.vocab_size
    eos = config.eos_token
    logits_r = logits.reshape(-1, V)
    finished_r = beam_state.finished.reshape(-1)
    probs = (prob_fn or log_softmax)(logits_r)
    probs = mask_probs(probs, eos, finished_r)
    with np.errstate(over="ignore", invalid="ignore"):
        total = beam_state.log_probs.reshape(-1, 1) + probs
        total = total.reshape(B, -1, V)
        lengths_to_add = np.ones([1, Wl, V], np.int32)
        lengths_to_add[:, :, eos] = 0
        add_mask = (1 - beam_state.finished.astype(np.int32))[..., None]
        new_len = beam_state.lengths[..., None] + add_mask * lengths_to_add
        scores = hyp_score(total, new_len, config.length_penalty_weight)
    scores_flat = scores.reshape(B, -1) if int(time_) > 0 else scores[:, start_offset, :]
    fn = config.choose_successors_fn or choose_top_k
    next_scores, word_indices = fn(scores_flat, W)
    word_indices = np.asarray(word_indices).astype(np.int32)
    total_flat = total.reshape(B, -1)
    next_probs = batch_gather(total_flat, word_indices).astype(total.dtype)
    next_word = word_indices % V
    next_beam = word_indices // V
    gathered_fin = batch_gather(beam_state.finished, next_beam).astype(bool)
    next_finished = np.logical_or(gathered_fin, next_word == eos)
    add = (next_word != eos).astype(np.int32) * (1 - next_finished.astype(np.int32))
    next_len = batch_gather(beam_state.lengths, next_beam).astype(np.int32) + add
    return (BeamSearchStepOutput(next_scores, next_word, next_beam),
            BeamSearchState(next_probs, next_finished, next_len))


def loc_optimal_beam_path(beam_sequence):
    """Unbatched back-tracking, `beam_search.py:300-345`.  Returns a list of T ids."""
    if len(beam_sequence) == 0:
        return []
    out, state = beam_sequence[-1]
    scores = out.scores
    fin = state.finished.astype(scores.dtype)
    if np.max(fin) > 0:
        scores = scores + (scores.dtype.type(1.0) - fin) * scores.dtype.type(-1e9)
    best = int(np.argmax(scores))
    seq = [out.predicted_ids[best]]
    beam = out.beam_parent_ids[best]
    for idx in range(len(beam_sequence) - 2, -1, -1):
        out, _ = beam_sequence[idx]
        seq.append(out.predicted_ids[beam])
        beam = out.beam_parent_ids[beam]
    return list(reversed(seq))


def loc_optimal_beam_path_batch(beam_sequence):
    """Batched back-tracking, `beam_search.py:545-613`.  If any beam of an image is
    finished, unfinished beams get -1e9 added (`:565-579`); argmax = first max
    (`:582`); then parents are followed backwards (`:595-609`).  Returns a list of T
    arrays [B,1] float32 (the dtype `batch_gather` produces), time-major."""
    if len(beam_sequence) == 0:
        return []
    out, state = beam_sequence[-1]
    scores = out.scores
    dt = scores.dtype.type
    fin = state.finished.astype(scores.dtype)
    any_fin = np.max(fin, axis=-1, keepdims=True)
    mask = (dt(1.0) - fin) * dt(-1e9)
    scores = np.where(any_fin > 0, scores + mask, scores)
    best = np.argmax(scores, axis=-1)[:, None]
    seq = [batch_gather(out.predicted_ids, best)]
    beam = batch_gather(out.beam_parent_ids, best)
    for idx in range(len(beam_sequence) - 2, -1, -1):
        out, _ = beam_sequence[idx]
        seq.append(batch_gather(out.predicted_ids, beam))
        beam = batch_gather(out.beam_parent_ids, beam)
    return list(reversed(seq))


# ----------------------------------------------------------------------------------
# whole-sequence driver
# ----------------------------------------------------------------------------------

def tile_features(features, beam_width, tile_mode):
    """`tf.tile(features, [W,1,1])` (`recurrent_example.py:486`) puts image r % B in
    row r ("reference"), although logits are later read as row = image*W + beam
    (`:541`).  "repeat" is the self-consistent layout (row r -> image r // W)."""
    if tile_mode == "reference":
        return np.tile(features, (beam_width, 1, 1))
    if tile_mode == "repeat":
        return np.repeat(features, beam_width, axis=0)
    raise ValueError("tile_mode must be 'reference' or 'repeat'")


def predict_image_caption_sequence(p, img, beam_width, max_len, start_id, eos_id,
                                   length_penalty_weight=1.0, tile_mode="reference",
                                   reset_state_from_feat=1, return_trace=False,
                                   hoist_w1=False, cell="lstm", choose_successors_fn=None):
    """`predict_image_caption_sequence` (`recurrent_example.py:477-562`).

    encoder -> tile -> reset_state -> T x (decoder step, beam step) -> back-track.
    The recurrent state is NOT reordered by parent beam between steps (`:526-529`,
    `:558`): only the token ids come from the beam search.  No early exit (`:517`).
    `hoist_w1` computes the step-invariant W1(features) once instead of every step (same
    values; a faster CPU variant used as a second baseline by the benchmark).
    `cell="gru"` runs the same loop over the GRU sibling decoder
    (`decoder/recurrent_decoder.py`; no driver in the reference constructs it, the loop is
    the LSTM one with the decoder swapped).
    `choose_successors_fn` replaces `choose_top_k` in the beam-search config (the driver
    passes `choose_top_k`, `:505-509`; the parity tests pass a chooser that follows another
    implementation through proven near-ties).
    Returns int32 tokens [B,T] (and the per-step trace if asked)."""
    if cell not in ("lstm", "gru"):
        raise ValueError("cell must be 'lstm' or 'gru'")
    step_fn = lstm_decoder_step if cell == "lstm" else gru_decoder_step
    reset_fn = reset_state if cell == "lstm" else reset_state_gru
    B = img.shape[0]
    W = beam_width
    features = mlp_encoder(img, p["enc_kernel"], p["enc_bias"])
    features = tile_features(features, W, tile_mode)
    dec_input = np.full([B * W, 1], start_id, np.int32)
    hidden = reset_fn(p, B * W, features, reset_state_from_feat)
    V = p["fc2_kernel"].shape[1]
    state = create_initial_beam_state_batch(B, W, dtype=img.dtype)
    config = BeamSearchConfig(W, V, eos_id, length_penalty_weight,
                              choose_successors_fn or choose_top_k)
    seq = []
    trace = []
    w1f = dense(features, p["att_W1_kernel"], p["att_W1_bias"]) if hoist_w1 else None
    for i in range(max_len):
        logits, hidden, weights, ctx = step_fn(p, dec_input, features, hidden, w1f=w1f)
        logits3 = logits.reshape(-1, W, V)
        out, state = beam_search_step_batch(i, logits3, state, config)
        seq.append((out, state))
        if return_trace:
            trace.append({"logits": logits3, "out": out, "state": state,
                          "h": hidden[0], "c": hidden[-1], "ctx": ctx,
                          "weights": weights})
        dec_input = out.predicted_ids.reshape(-1, 1)
    path = loc_optimal_beam_path_batch(seq)
    tokens = np.concatenate(path, axis=1).astype(np.int32)    # [B,T]
    if return_trace:
        return tokens, trace
    return tokens


def sequence_length_fn(sequences, sequence_mask, keepdims=False):
    """`utils/agent_utils.py:30-32`: number of tokens greater than the pad token."""
    flag = (np.asarray(sequences) > sequence_mask).astype(np.int32)
    return np.sum(flag, axis=-1, keepdims=keepdims)


def sample_categorical(logits, u, step=0, row_len=None, start_token=-1, eos_token=-1, pad_token=-1):
    """One step of `SequenceSampler.__call__` (`utils/agent_utils.py:81-125`): f32.min added at
    the pad / start / eos columns, softmax, a draw per row, then eos where step + 1 == length
    and pad where step >= length.  `tf.random.categorical` is replaced by the inverse of the
    cumulative distribution (float64) at the given uniform numbers u [N]."""
    x = np.array(logits, dtype=np.float32, copy=True)
    for tok in (pad_token, start_token, eos_token):
        if tok >= 0:
            with np.errstate(over="ignore"):
                x[:, tok] = x[:, tok] + np.float32(F32_MIN)
    e = np.exp((x - x.max(axis=1, keepdims=True)).astype(np.float64))
    cdf = np.cumsum(e, axis=1)
    target = np.asarray(u, dtype=np.float64) * cdf[:, -1]
    out = np.empty(x.shape[0], np.int32)
    for n in range(x.shape[0]):
        c = int(np.searchsorted(cdf[n], target[n], side="right"))
        c = min(c, x.shape[1] - 1)
        while e[n, c] == 0.0 and c > 0:      # (a column without mass is never drawn)
            c -= 1
        out[n] = c
    if row_len is not None:
        ln = np.asarray(row_len)
        out = np.where(step + 1 == ln, np.int32(eos_token), out)
        out = np.where(step >= ln, np.int32(pad_token), out).astype(np.int32)
    return out


def sequence_sampler(p, features, sequences, uniforms, start_token, eos_token, pad_token=0,
                     cell="lstm", reset_state_from_feat=1):
    """`SequenceSampler.__call__` (`utils/agent_utils.py:52-132`) over `RnnDecoderNetwork(decoder)`:
    T - 1 decoder steps from the start token, one draw per row and step (uniforms [T-1, B],
    see `sample_categorical`).  Returns int32 [B, T], start token first."""
    step_fn = lstm_decoder_step if cell == "lstm" else gru_decoder_step
    reset_fn = reset_state if cell == "lstm" else reset_state_gru
    sequences = np.asarray(sequences)
    B, T = sequences.shape
    row_len = sequence_length_fn(sequences, pad_token)
    hidden = reset_fn(p, B, features, reset_state_from_feat)
    dec_input = np.full([B, 1], start_token, np.int32)
    samples = [dec_input.copy()]
    for t in range(1, T):
        logits, hidden, _, _ = step_fn(p, dec_input, features, hidden)
        sample = sample_categorical(logits, uniforms[t - 1], step=t, row_len=row_len,
                                    start_token=start_token, eos_token=eos_token,
                                    pad_token=pad_token)
        samples.append(sample.reshape(-1, 1))
        dec_input = sample.reshape(-1, 1)
    return np.concatenate(samples, axis=1).astype(np.int32)


def baseline_sequence_beam_search(p, features, sequences, beam_width, vocab_size, max_length,
                                  start_token, eos_token, pad_token=0, cell="lstm",
                                  reset_state_from_feat=1, return_trace=False):
    """`BaselineSequeneBeamSearcher.__call__` (`utils/agent_utils.py:152-262`) over
    `RnnDecoderNetwork(decoder)` (`utils/reinforce.py:75-85`).

    features [B,L,D] are already encoded; tf.tile layout (row r -> image r % B, `:162`,
    lengths tiled the same way `:184`) while the logits are read as row = image*W + beam
    (`:239-240`).  Loop index i runs 1..max_length-1 and is the beam step's time, so the
    time-0 form never runs.  s_mask is float32.min at the start token (`:188-193`); with
    max_length <= 0 (length taken from `sequences`) e_mask / p_mask are float32.min
    everywhere but eos / pad and are added where i+1 == length / i+1 > length
    (`:195-237`).  length_penalty_weight is 1.0 (`:172`).  The state is not reordered
    (`:218-221`).  Returns int32 [B, steps+1] with the start token in column 0 (`:257-260`)."""
    step_fn = lstm_decoder_step if cell == "lstm" else gru_decoder_step
    reset_fn = reset_state if cell == "lstm" else reset_state_gru
    B, W, V = features.shape[0], beam_width, vocab_size
    fmin = np.float32(np.finfo(np.float32).min)
    dec_input = np.full([B * W, 1], start_token, np.int32)
    feats = np.tile(features, (W, 1, 1))
    hidden = reset_fn(p, B * W, feats, reset_state_from_feat)
    state = create_initial_beam_state_batch(B, W, dtype=features.dtype)
    config = BeamSearchConfig(W, V, eos_token, 1.0, choose_top_k)
    forced = max_length <= 0
    if forced:
        max_length = np.asarray(sequences).shape[1]
    seq_len = np.tile(sequence_length_fn(sequences, pad_token, keepdims=True), (W, 1))  # [B*W,1]
    s_mask = np.zeros((1, V), np.float32)
    s_mask[0, start_token] = fmin
    e_mask = np.full((1, V), fmin, np.float32)
    e_mask[0, eos_token] = 0.0
    p_mask = np.full((1, V), fmin, np.float32)
    p_mask[0, pad_token] = 0.0
    seq, trace = [], []
    with np.errstate(over="ignore", invalid="ignore"):
        for i in range(1, max_length):
            logits, hidden, _, _ = step_fn(p, dec_input, feats, hidden)
            logits = logits + s_mask
            if forced:
                logits = logits + e_mask * (i + 1 == seq_len).astype(np.float32)
                logits = logits + p_mask * (i + 1 > seq_len).astype(np.float32)
            logits3 = logits.reshape(-1, W, V)
            out, state = beam_search_step_batch(i, logits3, state, config)
            seq.append((out, state))
            if return_trace:
                trace.append({"logits": logits3, "out": out, "state": state})
            dec_input = out.predicted_ids.reshape(-1, 1)
    path = loc_optimal_beam_path_batch(seq)
    tokens = np.concatenate(path, axis=1).astype(np.int32)
    tokens = np.concatenate([np.full((B, 1), start_token, np.int32), tokens], axis=1)
    if return_trace:
        return tokens, trace
    return tokens


# ----------------------------------------------------------------------------------
# synthetic weights / inputs (Keras default initialisers)
# ----------------------------------------------------------------------------------

def _glorot(rng, fan_in, fan_out, shape=None):
    lim = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=shape or (fan_in, fan_out)).astype(np.float32)


def make_params(seed, F, D, E, H, V, U=None, C=None):
    """Synthetic weights: Glorot-uniform Dense/LSTM kernels (Keras default; the
    decoder passes `recurrent_initializer='glorot_uniform'`,
    `lstm_recurrent_decoder.py:23,36`), zero biases except the LSTM forget-gate slice
    (unit_forget_bias), embedding U(-0.05, 0.05)."""
    U = U or H
    C = C or D
    rng = np.random.default_rng(seed)
    p = {}
    p["enc_kernel"] = _glorot(rng, F, D)
    p["enc_bias"] = np.zeros(D, np.float32)
    p["embedding"] = rng.uniform(-0.05, 0.05, size=(V, E)).astype(np.float32)
    for idx in range(2):
        p["init%d_kernel" % idx] = _glorot(rng, D, H)
        p["init%d_bias" % idx] = np.zeros(H, np.float32)
    p["att_W1_kernel"] = _glorot(rng, D, U)
    p["att_W1_bias"] = np.zeros(U, np.float32)
    p["att_W2_kernel"] = _glorot(rng, H, U)
    p["att_W2_bias"] = np.zeros(U, np.float32)
    p["att_V_kernel"] = _glorot(rng, U, 1)
    p["att_V_bias"] = np.zeros(1, np.float32)
    p["fc3_kernel"] = _glorot(rng, H, 1)
    p["fc3_bias"] = np.zeros(1, np.float32)
    p["lstm_kernel"] = _glorot(rng, C + E, 4 * H)
    p["lstm_recurrent_kernel"] = _glorot(rng, H, 4 * H)
    b = np.zeros(4 * H, np.float32)
    b[H:2 * H] = 1.0
    p["lstm_bias"] = b
    p["fc1_kernel"] = _glorot(rng, H, H)
    p["fc1_bias"] = np.zeros(H, np.float32)
    p["fc2_kernel"] = _glorot(rng, H, V)
    p["fc2_bias"] = np.zeros(V, np.float32)
    return p


def make_peaked_params(seed, F, D, E, H, V, U=None, C=None, eos_id=4):
    """`make_params` rescaled towards what trained weights look like, for parity tests whose
    token decisions must not hinge on fp32 summation order: Keras-default weights give
    near-flat word distributions (logit spread 0.02: half of all top-k decisions have a
    relative score margin below 1e-5 and every image decodes to the same few tokens).  Here the
    embedding (x20), the LSTM kernels (x4, x2), the attention score vector (x8), the initial
    state projections (x2) and the vocabulary projection (x64) are scaled up and the end token
    gets a bias of 9: logit spread about 5, captions that differ per image, and the end token
    reached at every length between 1 and T for about half of the images."""
    p = make_params(seed, F, D, E, H, V, U, C)
    f = np.float32
    for name, gain in (("embedding", 20), ("lstm_kernel", 4), ("lstm_recurrent_kernel", 2),
                       ("fc2_kernel", 64), ("att_V_kernel", 8), ("init0_kernel", 2),
                       ("init1_kernel", 2)):
        p[name] = (p[name] * f(gain)).astype(np.float32)
    p["fc2_bias"][eos_id] = 9.0
    return p


def add_gru_params(p, seed):
    """GRU variables for the sibling decoder beside the LSTM ones of `make_params` (own
    random stream, so the LSTM parameters of a seed do not change): Glorot-uniform kernels
    (`recurrent_decoder.py:21,33` asks for a glorot_uniform recurrent kernel), zero bias
    [2,3H] (Keras GRUCell default with reset_after=True)."""
    rng = np.random.default_rng([seed, 7919])
    CE = p["lstm_kernel"].shape[0]
    H = p["lstm_recurrent_kernel"].shape[0]
    q = dict(p)
    q["gru_kernel"] = _glorot(rng, CE, 3 * H)
    q["gru_recurrent_kernel"] = _glorot(rng, H, 3 * H)
    q["gru_bias"] = np.zeros((2, 3 * H), np.float32)
    return q


def perturb_biases(p, seed, scale=0.05):
    """Give every bias a non-zero value so tests exercise the bias adds."""
    rng = np.random.default_rng(seed)
    q = dict(p)
    for k, v in p.items():
        if k.endswith("_bias"):
            q[k] = (v + rng.normal(0, scale, size=v.shape)).astype(np.float32)
    return q


def make_images(seed, B, L, F):
    """Synthetic CNN features: max(0, N(0,1)) * 0.5 (non-negative, half zeros)."""
    rng = np.random.default_rng(seed)
    return (np.maximum(rng.standard_normal((B, L, F)), 0.0) * 0.5).astype(np.float32)


def cast_params(p, dtype):
    return {k: v.astype(dtype) for k, v in p.items()}


# ----------------------------------------------------------------------------------
# Transformer decoder path (decoder/transform_decoder.py, attention/attention.py:160-562,
# decoder/ffn_layer.py, embedding/position_embedding.py).  No driver in the reference
# constructs it; the loop below is `predict_image_caption_sequence` with the decoder swapped.
# The restatement follows the code AS WRITTEN:
#   * `MultiHeadAttention(hidden_size, num_heads)` binds positionally to
#     (num_heads=hidden_size, key_dim=num_heads) (`transform_decoder.py:65-72` against
#     `attention.py:259-262`); `heads` and `key_dim` are free parameters here, the as-written
#     model has heads = hidden_size, key_dim = params["num_heads"].
#   * the additive causal bias (0 valid, -1e9 masked, `attention.py:160-180`) is handed over as
#     `attention_mask` and goes through `(1 - mask) * -1e9` (`attention.py:471-473`): valid
#     positions get -1e9 added, masked ones about -1e18 (`mask_mode="as_written"`; "causal" is
#     the conventional additive bias).
#   * the pre-norm wrapper normalises only its first argument (`transform_decoder.py:35-38`):
#     the self-attention query comes from LayerNorm(x), key and value from the raw x
#     (`kv_from_normed=False` as written).
#   * no cache: every step re-runs the whole prefix [0, start, ...] (`:177`, `:233-235`).
# TF 2.3 contracts used: EinsumDense "abc,cde->abde" (+ bias [d,e]) and "abcd,cde->abe";
# LayerNormalization(axis=-1, epsilon=1e-6) = (x-mean)/sqrt(var+eps)*gamma+beta with the
# biased variance; Embedding = row gather; Softmax over the last axis.
# ----------------------------------------------------------------------------------

def relative_position_embedding(length, hidden_size, min_timescale=1.0, max_timescale=1.0e4):
    """`RelativePositionEmbedding.call` (`embedding/position_embedding.py:67-102`), float32:
    [length, hidden] = [sin(pos * inv_timescales) | cos(pos * inv_timescales)]."""
    position = np.arange(length, dtype=np.float32)
    num_timescales = hidden_size // 2
    inc = np.float32(math.log(float(max_timescale) / float(min_timescale))) / \
        (np.float32(num_timescales) - np.float32(1))
    inv = np.float32(min_timescale) * np.exp(np.arange(num_timescales, dtype=np.float32) * -inc)
    scaled = position[:, None] * inv[None, :].astype(np.float32)
    return np.concatenate([np.sin(scaled), np.cos(scaled)], axis=1).astype(np.float32)


def layer_norm(x, gamma, beta, eps=1e-6):
    mean = np.mean(x, axis=-1, keepdims=True, dtype=x.dtype)
    var = np.mean(np.square(x - mean), axis=-1, keepdims=True, dtype=x.dtype)
    return (x - mean) / np.sqrt(var + x.dtype.type(eps)) * gamma + beta


def decoder_self_attention_bias(length, dtype=np.float32):
    """`get_decoder_self_attention_bias` (`attention/attention.py:160-180`): [1,1,T,T], 0 on and
    below the diagonal, -1e9 above."""
    valid = np.tril(np.ones((length, length), dtype=dtype))
    return (dtype(-1e9) * (dtype(1.0) - valid)).reshape(1, 1, length, length)


def multi_head_attention(query, value, p, prefix, heads, key_dim, attention_mask=None):
    """`MultiHeadAttention.call` (`attention/attention.py:526-562`) with key = value.
    query [B,T,Dq], value [B,S,Dv]; kernels `prefix + q/k/v_kernel` [Din, heads*key_dim],
    `o_kernel` [heads*key_dim, Dq].  `attention_mask` (broadcastable to [B,heads,T,S]) goes
    through `(1 - mask) * -1e9` exactly as `_masked_softmax` does (`:461-474`)."""
    B, T, _ = query.shape
    S = value.shape[1]
    dt = query.dtype
    q = (dense(query, p[prefix + "q_kernel"], p[prefix + "q_bias"])).reshape(B, T, heads, key_dim)
    k = (dense(value, p[prefix + "k_kernel"], p[prefix + "k_bias"])).reshape(B, S, heads, key_dim)
    v = (dense(value, p[prefix + "v_kernel"], p[prefix + "v_bias"])).reshape(B, S, heads, key_dim)
    q = q * dt.type(1.0 / math.sqrt(float(key_dim)))
    scores = np.einsum("aecd,abcd->acbe", k, q)                      # [B,heads,T,S]
    if attention_mask is not None:
        with np.errstate(over="ignore"):
            adder = (dt.type(1.0) - attention_mask.astype(dt)) * dt.type(-1e9)
        scores = scores + adder
    w = softmax(scores, axis=-1)
    out = np.einsum("acbe,aecd->abcd", w, v).reshape(B, T, heads * key_dim)
    return dense(out, p[prefix + "o_kernel"], p[prefix + "o_bias"])


_CHR_IDX = "abcdefghijklmnopqrstuvwxyz"


def _attention_equations(rank, attn_axes):
    """`_build_attention_equation` (`attention/attention.py:84-125`): the einsum equations of
    the query-key product and of the combination for projected tensors of rank `rank`
    (batch, <dims>, heads, channels) and the given attention axes."""
    target = _CHR_IDX[:rank]
    batch_dims = tuple(int(i) for i in np.delete(range(rank), tuple(attn_axes) + (rank - 1,)))
    offset = rank
    source = ""
    for i in range(rank):
        if i in batch_dims or i == rank - 1:
            source += target[i]
        else:
            source += _CHR_IDX[offset]
            offset += 1
    product = "".join([target[i] for i in batch_dims] + [target[i] for i in attn_axes] +
                      [source[i] for i in attn_axes])
    return "%s,%s->%s" % (source, target, product), "%s,%s->%s" % (product, source, target), len(product)


def multi_head_attention_general(query, value, p, prefix, num_heads, key_dim, key=None, value_dim=None,
                                 attention_axes=None, attention_mask=None, output_shape=None,
                                 use_bias=True):
    """`MultiHeadAttention.call` (`attention/attention.py:526-562`) with every constructor
    option (`:259-294`): returns (attention_output, attention_scores).  Kernels are the flattened
    EinsumDense kernels: `q/k_kernel` [Din, heads*key_dim], `v_kernel` [Din, heads*value_dim],
    `o_kernel` [heads*value_dim, prod(output_shape)].  The soft-max is over the LAST axis of the
    scores only (`:457-458`), also when several attention axes are given."""
    dt = query.dtype
    value_dim = value_dim if value_dim else key_dim
    if key is None:
        key = value

    def proj(x, name, d):
        b = p[prefix + name + "_bias"] if use_bias else None
        y = np.matmul(x, p[prefix + name + "_kernel"])
        if b is not None:
            y = y + b
        return y.astype(dt).reshape(x.shape[:-1] + (num_heads, d))

    q = proj(query, "q", key_dim)
    k = proj(key, "k", key_dim)
    v = proj(value, "v", value_dim)
    rank = q.ndim
    if attention_axes is None:
        attention_axes = tuple(range(1, rank - 2))                        # `:449-452`
    elif not hasattr(attention_axes, "__len__"):
        attention_axes = (attention_axes,)
    attention_axes = tuple(attention_axes)
    dot_eq, combine_eq, _ = _attention_equations(rank, attention_axes)
    q = q * dt.type(1.0 / math.sqrt(float(key_dim)))                       # `:501`
    scores = np.einsum(dot_eq, k, q)                                       # `:506`
    if attention_mask is not None:                                         # `:461-474`
        m = np.asarray(attention_mask)
        axis = -len(attention_axes) * 2 - 1
        for _ in range(scores.ndim - m.ndim):
            m = np.expand_dims(m, axis=axis)
        with np.errstate(over="ignore"):
            scores = scores + (dt.type(1.0) - m.astype(dt)) * dt.type(-1e9)
    scores = softmax(scores, axis=-1)
    out = np.einsum(combine_eq, scores, v)                                 # [query dims, heads, value_dim]
    if output_shape:
        out_dims = tuple(output_shape) if hasattr(output_shape, "__len__") else (int(output_shape),)
    else:
        out_dims = (query.shape[-1],)
    y = np.matmul(out.reshape(out.shape[:-2] + (num_heads * value_dim,)), p[prefix + "o_kernel"])
    if use_bias:
        y = y + p[prefix + "o_bias"]
    return y.astype(dt).reshape(query.shape[:-1] + out_dims), scores


def transformer_decoder_stack(p, x, encoder_outputs, n_layers, heads, key_dim,
                              mask_mode="as_written", kv_from_normed=False):
    """`DecoderStack.call` (`decoder/transform_decoder.py:92-151`) with the pre-norm wrapper
    (`:30-42`): x += sublayer(LayerNorm(x), ...) for self-attention, encoder-decoder attention
    and the feed-forward network, then the output LayerNorm."""
    T = x.shape[1]
    bias = decoder_self_attention_bias(T, x.dtype.type)
    if mask_mode == "as_written":
        self_mask = bias                      # the bias travels as `attention_mask`
    elif mask_mode == "causal":
        self_mask = np.tril(np.ones((T, T), x.dtype)).reshape(1, 1, T, T)
    else:
        raise ValueError("mask_mode must be 'as_written' or 'causal'")
    for l in range(n_layers):
        pre = "l%d_" % l
        y = layer_norm(x, p[pre + "self_ln_gamma"], p[pre + "self_ln_beta"])
        x = x + multi_head_attention(y, y if kv_from_normed else x, p, pre + "self_", heads,
                                     key_dim, attention_mask=self_mask)
        y = layer_norm(x, p[pre + "cross_ln_gamma"], p[pre + "cross_ln_beta"])
        x = x + multi_head_attention(y, encoder_outputs, p, pre + "cross_", heads, key_dim)
        y = layer_norm(x, p[pre + "ffn_ln_gamma"], p[pre + "ffn_ln_beta"])
        f = np.maximum(dense(y, p[pre + "ffn_filter_kernel"], p[pre + "ffn_filter_bias"]), 0)
        x = x + dense(f, p[pre + "ffn_output_kernel"], p[pre + "ffn_output_bias"])
    return layer_norm(x, p["out_ln_gamma"], p["out_ln_beta"])


def transform_decoder_step(p, dec_input, features, hidden, n_layers, heads, key_dim,
                           mask_mode="as_written", kv_from_normed=False):
    """`TransformDecoder.call` / `decode` (`decoder/transform_decoder.py:167-231`): the state
    is the int prefix [N,len]; targets = [prefix | dec_input]; embedding + position encoding
    (`:206-210`), the stack, fc2 on every position, the last position's logits.
    Returns (logits[N,V], [targets], None, outputs[:, -1, :])."""
    targets = np.concatenate([hidden[0], np.asarray(dec_input).reshape(-1, 1)], axis=-1)
    x = p["t_embedding"][targets]                                    # Embedding: [N,len,hidden]
    x = x + relative_position_embedding(targets.shape[1], x.shape[-1])[None].astype(x.dtype)
    out = transformer_decoder_stack(p, x, features, n_layers, heads, key_dim, mask_mode,
                                    kv_from_normed)
    logits = dense(out, p["t_fc2_kernel"], p["t_fc2_bias"])
    return logits[:, -1, :], [targets], None, out[:, -1, :]


def predict_transformer_caption_sequence(p, img, beam_width, max_len, start_id, eos_id, n_layers,
                                         heads, key_dim, length_penalty_weight=1.0,
                                         tile_mode="reference", mask_mode="as_written",
                                         kv_from_normed=False, return_trace=False):
    """The loop of `predict_image_caption_sequence` (`recurrent_example.py:477-562`) over the
    transformer decoder: encoder -> tile -> reset_state = zeros [N,1] int
    (`transform_decoder.py:233-235`) -> T x (decoder step, beam step) -> back-track.  As in
    that loop the state (here the token prefix of each row) is not reordered by parent beam."""
    B, W = img.shape[0], beam_width
    features = tile_features(mlp_encoder(img, p["enc_kernel"], p["enc_bias"]), W, tile_mode)
    dec_input = np.full([B * W, 1], start_id, np.int32)
    hidden = [np.zeros((B * W, 1), np.int32)]
    V = p["t_fc2_kernel"].shape[1]
    state = create_initial_beam_state_batch(B, W, dtype=img.dtype)
    config = BeamSearchConfig(W, V, eos_id, length_penalty_weight, choose_top_k)
    seq, trace = [], []
    for i in range(max_len):
        logits, hidden, _, last = transform_decoder_step(p, dec_input, features, hidden, n_layers,
                                                         heads, key_dim, mask_mode, kv_from_normed)
        logits3 = logits.reshape(-1, W, V)
        out, state = beam_search_step_batch(i, logits3, state, config)
        seq.append((out, state))
        if return_trace:
            trace.append({"logits": logits3, "out": out, "state": state, "last": last})
        dec_input = out.predicted_ids.reshape(-1, 1)
    tokens = np.concatenate(loc_optimal_beam_path_batch(seq), axis=1).astype(np.int32)
    return (tokens, trace) if return_trace else tokens


def make_transformer_params(seed, F, D, V, hidden, n_layers, heads, key_dim, filter_size):
    """Synthetic weights of the transformer path: Glorot-uniform kernels (EinsumDense fans
    taken over the flattened [in, heads*key_dim] matrices), zero biases, LayerNorm gamma 1 /
    beta 0, embedding U(-0.05, 0.05); the MLP encoder as in `make_params` (features dim D)."""
    rng = np.random.default_rng([seed, 104729])
    HD = heads * key_dim
    p = {"enc_kernel": _glorot(rng, F, D), "enc_bias": np.zeros(D, np.float32),
         "t_embedding": rng.uniform(-0.05, 0.05, size=(V, hidden)).astype(np.float32),
         "t_fc2_kernel": _glorot(rng, hidden, V), "t_fc2_bias": np.zeros(V, np.float32),
         "out_ln_gamma": np.ones(hidden, np.float32), "out_ln_beta": np.zeros(hidden, np.float32)}
    for l in range(n_layers):
        pre = "l%d_" % l
        for att, d_kv in (("self_", hidden), ("cross_", D)):
            p[pre + att + "q_kernel"] = _glorot(rng, hidden, HD)
            p[pre + att + "k_kernel"] = _glorot(rng, d_kv, HD)
            p[pre + att + "v_kernel"] = _glorot(rng, d_kv, HD)
            p[pre + att + "o_kernel"] = _glorot(rng, HD, hidden)
            for b in ("q_bias", "k_bias", "v_bias"):
                p[pre + att + b] = np.zeros(HD, np.float32)
            p[pre + att + "o_bias"] = np.zeros(hidden, np.float32)
        for ln in ("self_ln_", "cross_ln_", "ffn_ln_"):
            p[pre + ln + "gamma"] = np.ones(hidden, np.float32)
            p[pre + ln + "beta"] = np.zeros(hidden, np.float32)
        p[pre + "ffn_filter_kernel"] = _glorot(rng, hidden, filter_size)
        p[pre + "ffn_filter_bias"] = np.zeros(filter_size, np.float32)
        p[pre + "ffn_output_kernel"] = _glorot(rng, filter_size, hidden)
        p[pre + "ffn_output_bias"] = np.zeros(hidden, np.float32)
    return p


def perturb_transformer_params(p, seed, scale=0.05):
    """Non-trivial biases and LayerNorm gamma / beta so that tests exercise them."""
    rng = np.random.default_rng(seed)
    q = dict(p)
    for k, v in p.items():
        if k.endswith("_bias") or k.endswith("_beta") or k.endswith("_gamma"):
            q[k] = (v + rng.normal(0, scale, size=v.shape)).astype(np.float32)
    return q
