"""torch-CPU op-for-op restatement of the reference's decode loop.  TEST INFRASTRUCTURE ONLY.

The CPU baseline of the benchmark: the same op sequence as `predict_image_caption_sequence`
(`recurrent_example.py:477-562`) executed with torch CPU kernels (multi-threaded MKL / oneDNN,
the closest stand-in here for TensorFlow's Eigen CPU kernels - TensorFlow is not installed),
INCLUDING what the reference does inefficiently: `W1(features)` recomputed every step
(`attention/attention.py:48`), the features physically tiled W times
(`recurrent_example.py:486`), full `[B,W,V]` masks, one-hots and an elementwise pow for the
length penalty (`beam_search/beam_search.py:396-445`), and one gather per batch row for every
`batch_gather` (`tf.map_fn`, `beam_search.py:347-358`).

Only `tests/` and the `cpu_baseline` leg of `bench.py` import this file; it is checked
token-for-token against `oracle/caption_oracle.py` (`tests/test_oracle_crosscheck.py`).  Parity
with the reference itself is unpinned beyond the beam step (see the header of the numpy oracle)."""
from __future__ import annotations

import torch

F32_MIN = torch.finfo(torch.float32).min


def dense(x, kernel, bias):
    """Keras `Dense` without activation over the last axis."""
    return torch.matmul(x, kernel) + bias


def batch_gather(a, b):
    """`beam_search.py:347-358`: `tf.map_fn` over the batch rows, one gather each, both
    operands cast to float32 and the result left float32."""
    a32 = a.to(torch.float32)
    idx = b.to(torch.float32).to(torch.int64)
    return torch.stack([a32[i][idx[i]] for i in range(a32.shape[0])])


def bahdanau_attention(features, hidden, p):
    """`attention/attention.py:40-65`, W1(features) recomputed on every call."""
    w1f = dense(features, p["att_W1_kernel"], p["att_W1_bias"])
    w2h = dense(hidden[:, None, :], p["att_W2_kernel"], p["att_W2_bias"])
    score = dense(torch.tanh(w1f + w2h), p["att_V_kernel"], p["att_V_bias"])
    weights = torch.softmax(score, dim=1)
    return torch.sum(weights * features, dim=1), weights


def lstm_decoder_step(p, dec_input, features, hidden):
    """`decoder/lstm_recurrent_decoder.py:59-116` with the Keras LSTMCell contract
    (z = (x.K + h.R) + b, gates i, f, g, o)."""
    emb = p["embedding"][dec_input.reshape(-1).to(torch.int64)]
    h_state = hidden[-1]
    context, weights = bahdanau_attention(features, h_state, p)
    context = context * torch.sigmoid(dense(h_state, p["fc3_kernel"], p["fc3_bias"]))
    x = torch.cat([context, emb], dim=-1)
    z = torch.matmul(x, p["lstm_kernel"])
    z = z + torch.matmul(hidden[0], p["lstm_recurrent_kernel"])
    z = z + p["lstm_bias"]
    z0, z1, z2, z3 = torch.chunk(z, 4, dim=1)
    c = torch.sigmoid(z1) * hidden[1] + torch.sigmoid(z0) * torch.tanh(z2)
    h = torch.sigmoid(z3) * torch.tanh(c)
    out = dense(h, p["fc1_kernel"], p["fc1_bias"])
    return dense(out, p["fc2_kernel"], p["fc2_bias"]), [h, c], weights, context


def beam_search_step_batch(time_, logits, log_probs, finished, lengths, W, V, eos, lpw):
    """`beam_search.py:360-543`, tensors [B,W] in, the step output and next state out."""
    B = logits.shape[0]
    probs = torch.log_softmax(logits.reshape(-1, V), dim=-1)
    fm = (1.0 - finished.reshape(-1).to(torch.float32))[:, None]
    row = torch.full([V], F32_MIN, dtype=torch.float32)
    row[eos] = 0
    probs = (1.0 - fm) * row + fm * probs                                    # mask_probs
    total = (log_probs.reshape(-1, 1) + probs).reshape(B, -1, V)
    lengths_to_add = torch.ones([1, W, V], dtype=torch.int32)
    lengths_to_add[:, :, eos] = 0
    add_mask = (1 - finished.to(torch.int32))[..., None]
    new_len = lengths[..., None] + add_mask * lengths_to_add
    penalty = torch.pow(5.0 + new_len.to(torch.float32), float(lpw)) / float((5.0 + 1.0) ** lpw)
    scores = total / penalty
    scores_flat = scores.reshape(B, -1) if time_ > 0 else scores[:, 0, :]
    next_scores, word_indices = torch.sort(scores_flat, dim=-1, descending=True, stable=True)
    next_scores, word_indices = next_scores[:, :W], word_indices[:, :W].to(torch.int32)
    next_probs = batch_gather(total.reshape(B, -1), word_indices)
    next_word = word_indices % V
    next_beam = torch.div(word_indices, V, rounding_mode="floor")
    next_finished = torch.logical_or(batch_gather(finished, next_beam).to(torch.bool), next_word == eos)
    add = (next_word != eos).to(torch.int32) * (1 - next_finished.to(torch.int32))
    next_len = batch_gather(lengths, next_beam).to(torch.int32) + add
    return (next_scores, next_word, next_beam), (next_probs, next_finished, next_len)


def loc_optimal_beam_path_batch(seq):
    """`beam_search.py:545-613`."""
    (scores, ids, parents), (_, finished, _) = seq[-1]
    fin = finished.to(torch.float32)
    any_fin = fin.max(dim=-1, keepdim=True).values
    scores = torch.where(any_fin > 0, scores + (1.0 - fin) * -1e9, scores)
    best = torch.argmax(scores, dim=-1)[:, None]
    out = [batch_gather(ids, best)]
    beam = batch_gather(parents, best)
    for idx in range(len(seq) - 2, -1, -1):
        (_, ids, parents), _ = seq[idx]
        out.append(batch_gather(ids, beam))
        beam = batch_gather(parents, beam)
    return list(reversed(out))


def to_torch(p):
    return {k: torch.from_numpy(v) for k, v in p.items()}


@torch.no_grad()
def predict_image_caption_sequence(p, img, beam_width, max_len, start_id, eos_id,
                                   length_penalty_weight=1.0):
    """`recurrent_example.py:477-562` in the reference's own tile order.  p: dict of torch
    float32 tensors (to_torch), img: torch [B,L,F].  Returns int32 tokens [B,T]."""
    B, W = img.shape[0], beam_width
    V = p["fc2_kernel"].shape[1]
    features = torch.relu(dense(img, p["enc_kernel"], p["enc_bias"]))
    features = features.repeat(W, 1, 1)                                      # tf.tile
    dec_input = torch.full([B * W, 1], start_id, dtype=torch.int32)
    pooled = features.mean(dim=-2)
    hidden = [dense(pooled, p["init%d_kernel" % i], p["init%d_bias" % i]) for i in range(2)]
    state = (torch.zeros(B, W), torch.zeros(B, W, dtype=torch.bool), torch.zeros(B, W, dtype=torch.int32))
    seq = []
    for i in range(max_len):
        logits, hidden, _, _ = lstm_decoder_step(p, dec_input, features, hidden)
        out, state = beam_search_step_batch(i, logits.reshape(-1, W, V), *state, W, V, eos_id,
                                            length_penalty_weight)
        seq.append((out, state))
        dec_input = out[1].reshape(-1, 1)
    return torch.cat(loc_optimal_beam_path_batch(seq), dim=1).to(torch.int32)
