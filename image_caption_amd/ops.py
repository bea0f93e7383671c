"""Torch-tensor front end of the libicap operators.

torch is used for device memory and streams only: every function checks its
tensors (ROCm device, dtype, contiguity, shape) and hands raw device pointers plus
the current HIP stream to the C ABI."""
from __future__ import annotations

import sys
import threading

import torch

from . import _lib
from ._lib import ACT_NONE, ACT_RELU, TILE_REFERENCE, TILE_REPEAT  # noqa: F401


# Operands of one call must share a device, and the call runs on that device's current stream
# with that device current - also in a process that drives several GPUs while another device
# is the current one.  _chk notes the device of every tensor it checks; current_stream(),
# always the last argument of a call, resolves them.
_tls = threading.local()


def _pending():
    if not hasattr(_tls, "devs"):
        _tls.devs = set()
    return _tls.devs


def current_stream():
    devs = _pending()
    try:
        if len(devs) > 1:
            raise ValueError("operands live on different devices: %s" % sorted(map(str, devs)))
        dev = next(iter(devs)) if devs else torch.device("cuda", torch.cuda.current_device())
    finally:
        devs.clear()
        _tls.frame = None
    _tls.dev = dev
    return torch.cuda.current_stream(dev).cuda_stream


def _call(name, *args):
    """_lib.call with the operands' device current (set by current_stream() just before)."""
    dev = getattr(_tls, "dev", None)
    _tls.dev = None
    if dev is None or dev.index == torch.cuda.current_device():
        return _lib.call(name, *args)
    with torch.cuda.device(dev):
        return _lib.call(name, *args)


def _chk(t, dtype, name):
    try:
        if not isinstance(t, torch.Tensor):
            raise TypeError("%s must be a torch.Tensor" % name)
        if not t.is_cuda:
            raise ValueError("%s must live on the GPU (got device %s)" % (name, t.device))
        if t.dtype != dtype:
            raise TypeError("%s must be %s (got %s)" % (name, dtype, t.dtype))
        if not t.is_contiguous():
            raise ValueError("%s must be contiguous" % name)
    except Exception:
        _pending().clear()
        raise
    # Operands are collected per operator call, i.e. per frame of the calling operator: a set
    # left behind by a call whose argument list raised between two checks (and so never
    # reached current_stream()) belongs to another frame and is dropped here instead of
    # being mixed into this call's devices.
    frame = sys._getframe(1)
    if frame.f_code is _opt.__code__:
        frame = frame.f_back
    if getattr(_tls, "frame", None) is not frame:
        _pending().clear()
        _tls.frame = frame
    _pending().add(t.device)
    return t.data_ptr()


def _opt(t, dtype, name):
    return None if t is None else _chk(t, dtype, name)


f32, i32, u8 = torch.float32, torch.int32, torch.uint8


def dense(x, kernel, bias=None, act=ACT_NONE, out=None):
    """act(x . kernel + bias) over the last axis of x."""
    K, N = kernel.shape
    if x.shape[-1] != K:
        raise ValueError("dense: input last dim %d != kernel rows %d" % (x.shape[-1], K))
    M = x.numel() // K
    if out is None:
        out = torch.empty(*x.shape[:-1], N, dtype=f32, device=x.device)
    if M:
        _call("icap_dense_f32", _chk(x, f32, "x"), _chk(kernel, f32, "kernel"),
                  _opt(bias, f32, "bias"), _chk(out, f32, "out"), M, K, N, act, current_stream())
    return out


def pool_features(features, mask=None):
    G, L, D = features.shape
    out = torch.empty(G, D, dtype=f32, device=features.device)
    _call("icap_pool_features_f32", _chk(features, f32, "features"), _opt(mask, f32, "mask"),
              _chk(out, f32, "out"), G, L, D, current_stream())
    return out


def expand_rows(src, W, scale=1.0, tile_mode=TILE_REFERENCE):
    B, H = src.shape
    out = torch.empty(B * W, H, dtype=f32, device=src.device)
    _call("icap_expand_rows_f32", _chk(src, f32, "src"), _chk(out, f32, "out"), B, W, H,
              float(scale), tile_mode, current_stream())
    return out


def embedding(table, ids):
    V, E = table.shape
    ids = ids.reshape(-1)
    N = ids.numel()
    out = torch.empty(N, E, dtype=f32, device=table.device)
    _call("icap_embedding_f32", _chk(table, f32, "table"), _chk(ids, i32, "ids"),
              _chk(out, f32, "out"), N, V, E, current_stream())
    return out


ATT_ACTIVATIONS = {"tanh": 0, "linear": 1, "relu": 2, "sigmoid": 3}


def attention(pf, features, q, v, bv, score_bias=None, gate_state=None, gate_w=None, gate_b=0.0,
              rows_per_group=1, row_stride_g=1, row_stride_k=0, want_weights=True, activation="tanh"):
    """Returns (ctx[N,D], weights[N,L] or None).  activation: the function inside the score
    (one of ATT_ACTIVATIONS; anything but tanh serves rows_per_group = 1)."""
    G, L, U = pf.shape
    D = features.shape[-1]
    N = q.shape[0]
    if features.shape[:2] != (G, L) or q.shape[1] != U or v.numel() != U:
        raise ValueError("attention: inconsistent shapes")
    if N != G * rows_per_group:
        raise ValueError("attention: N must equal G * rows_per_group")
    H = gate_state.shape[1] if gate_state is not None else 0
    ctx = torch.empty(N, D, dtype=f32, device=pf.device)
    weights = torch.empty(N, L, dtype=f32, device=pf.device) if want_weights else None
    if activation not in ATT_ACTIVATIONS:
        raise ValueError("attention: activation must be one of %s" % sorted(ATT_ACTIVATIONS))
    _call("icap_attention_act_f32", _chk(pf, f32, "pf"), _chk(features, f32, "features"),
              _chk(q, f32, "q"), _chk(v, f32, "v"), float(bv), _opt(score_bias, f32, "score_bias"),
              _opt(gate_state, f32, "gate_state"), _opt(gate_w, f32, "gate_w"), float(gate_b),
              _chk(ctx, f32, "ctx"), _opt(weights, f32, "weights"), G, rows_per_group,
              row_stride_g, row_stride_k, L, U, D, H, ATT_ACTIVATIONS[activation], current_stream())
    return ctx, weights


def lstm_step(ctx, h, c, kernel, recurrent_kernel, bias, emb=None, table=None, ids=None,
              emb_proj=None):
    """Returns (h', c').  Pass emb[N,E], or ids[N] with table[V,E] and/or the pre-multiplied
    emb_proj[V,4H] = table . kernel[C:C+E]."""
    N, Cdim = ctx.shape
    H = h.shape[1]
    if ids is not None:
        ids = ids.reshape(-1)
        if emb_proj is not None:
            V = emb_proj.shape[0]
            E = kernel.shape[0] - Cdim
        else:
            V, E = table.shape
    else:
        V, E = 0, emb.shape[1]
    if kernel.shape != (Cdim + E, 4 * H) or recurrent_kernel.shape != (H, 4 * H):
        raise ValueError("lstm_step: kernel shapes do not match [C+E,4H] / [H,4H]")
    h_out = torch.empty_like(h)
    c_out = torch.empty_like(c)
    _call("icap_lstm_step_f32", _chk(ctx, f32, "ctx"), _opt(emb, f32, "emb"),
              _opt(table, f32, "table"), _opt(ids, i32, "ids"), _opt(emb_proj, f32, "emb_proj"),
              _chk(h, f32, "h"),
              _chk(c, f32, "c"), _chk(kernel, f32, "kernel"),
              _chk(recurrent_kernel, f32, "recurrent_kernel"), _chk(bias, f32, "bias"),
              _chk(h_out, f32, "h_out"), _chk(c_out, f32, "c_out"), N, Cdim, E, H, V,
              current_stream())
    return h_out, c_out


def gru_pack(kernel, recurrent_kernel, bias):
    """Keras GRUCell variables (kernel [CE,3H], recurrent_kernel [H,3H], bias [2,3H]) re-laid
    as the four-column operands gru_step takes: (kernel4 [CE,4H], recurrent4 [H,4H],
    bias4 [4H])."""
    CE, H3 = kernel.shape
    H = H3 // 3
    if H3 != 3 * H or recurrent_kernel.shape != (H, 3 * H) or bias.numel() != 6 * H:
        raise ValueError("gru_pack: expected kernel [CE,3H], recurrent_kernel [H,3H], bias [2,3H]")
    k4 = torch.empty(CE, 4 * H, dtype=f32, device=kernel.device)
    r4 = torch.empty(H, 4 * H, dtype=f32, device=kernel.device)
    b4 = torch.empty(4 * H, dtype=f32, device=kernel.device)
    _call("icap_gru_pack_f32", _chk(kernel, f32, "kernel"),
              _chk(recurrent_kernel, f32, "recurrent_kernel"), _chk(bias, f32, "bias"),
              k4.data_ptr(), r4.data_ptr(), b4.data_ptr(), CE, H, current_stream())
    return k4, r4, b4


def gru_step(ctx, h, kernel4, recurrent4, bias4, emb=None, table=None, ids=None, emb_proj=None):
    """One GRU cell step on [ctx | emb | h] with operands from gru_pack; returns h'.  Pass
    emb[N,E], or ids[N] with table[V,E] and/or emb_proj[V,4H] = table . kernel4[C:C+E]."""
    N, Cdim = ctx.shape
    H = h.shape[1]
    if ids is not None:
        ids = ids.reshape(-1)
        if emb_proj is not None:
            V = emb_proj.shape[0]
            E = kernel4.shape[0] - Cdim
        else:
            V, E = table.shape
    else:
        V, E = 0, emb.shape[1]
    if kernel4.shape != (Cdim + E, 4 * H) or recurrent4.shape != (H, 4 * H):
        raise ValueError("gru_step: packed kernel shapes do not match [C+E,4H] / [H,4H]")
    h_out = torch.empty_like(h)
    _call("icap_gru_step_f32", _chk(ctx, f32, "ctx"), _opt(emb, f32, "emb"),
              _opt(table, f32, "table"), _opt(ids, i32, "ids"), _opt(emb_proj, f32, "emb_proj"),
              _chk(h, f32, "h"), _chk(kernel4, f32, "kernel4"), _chk(recurrent4, f32, "recurrent4"),
              _chk(bias4, f32, "bias4"), _chk(h_out, f32, "h_out"), N, Cdim, E, H, V,
              current_stream())
    return h_out


def logit_masks(logits, start_token=-1, row_len=None, step_plus_1=0, eos_token=-1, pad_token=-1):
    """The baseline searcher's additive masks on logits [N,V], in place (returns logits):
    float32.min on column start_token; with row_len [N] int32, on all columns but eos_token
    where step_plus_1 == row_len[n] and on all but pad_token where step_plus_1 > row_len[n]."""
    V = logits.shape[-1]
    N = logits.numel() // V
    _call("icap_logit_masks_f32", _chk(logits, f32, "logits"), N, V, int(start_token),
              _opt(row_len, i32, "row_len"), int(step_plus_1), int(eos_token), int(pad_token),
              current_stream())
    return logits


# ---- transformer decoder path (one target position per call) --------------------------

def position_table(length, hidden, device):
    """[length, hidden] sinusoid position encoding (RelativePositionEmbedding)."""
    t = torch.empty(length, hidden, dtype=f32, device=device)
    _call("icap_position_table_f32", _chk(t, f32, "table"), int(length), int(hidden), current_stream())
    return t


def embed_pos(table, ids, pos_row):
    """table[ids] + pos_row for one target position: [N, hidden]."""
    V, D = table.shape
    N = ids.numel()
    x = torch.empty(N, D, dtype=f32, device=table.device)
    _call("icap_embed_pos_f32", _chk(table, f32, "table"), _chk(ids.reshape(-1), i32, "ids"),
              _chk(pos_row, f32, "pos_row"), x.data_ptr(), N, V, D, current_stream())
    return x


def add_(x, o):
    """x += o in place; o either has x's size or is broadcast over the leading axes."""
    n, period = x.numel(), o.numel()
    if n % period:
        raise ValueError("add_: the addend does not tile the tensor")
    _call("icap_add_f32", _chk(x, f32, "x"), _chk(o, f32, "o"), n, period, current_stream())
    return x


def residual_layernorm(x, o, gamma, beta, eps=1e-6):
    """x += o in place (o may be None); returns LayerNorm(x) * gamma + beta."""
    D = x.shape[-1]
    N = x.numel() // D
    y = torch.empty_like(x)
    _call("icap_residual_layernorm_f32", _chk(x, f32, "x"), _opt(o, f32, "o"),
              _chk(gamma, f32, "gamma"), _chk(beta, f32, "beta"), y.data_ptr(), N, D, float(eps),
              current_stream())
    return y


def mha_self_step(q, kv_cache, heads, key_dim, pos, mask_mode=0):
    """q [N, heads*key_dim], kv_cache [Tcap, N, 2*heads*key_dim] (position `pos` already
    written) -> combined values [N, heads*key_dim] of the newest position."""
    N = q.shape[0]
    out = torch.empty_like(q)
    _call("icap_mha_self_step_f32", _chk(q, f32, "q"), _chk(kv_cache, f32, "kv_cache"),
              out.data_ptr(), N, int(heads), int(key_dim), int(pos), int(mask_mode), current_stream())
    return out


def mha_scores(q, k, mask=None, softmax_segment=None):
    """q [Bf,T,N,Dk], k [Bf,S,N,Dk], mask None or [Bf,N,T,S] (1 = attend) -> attention
    probabilities [Bf,N,T,S]; the soft-max runs over runs of softmax_segment key positions
    (default: all S)."""
    Bf, T, N, Dk = q.shape
    S = k.shape[1]
    if k.shape != (Bf, S, N, Dk):
        raise ValueError("mha_scores: k must be [Bf,S,N,Dk]")
    if mask is not None and tuple(mask.shape) != (Bf, N, T, S):
        raise ValueError("mha_scores: mask must be [Bf,N,T,S]")
    probs = torch.empty(Bf, N, T, S, dtype=f32, device=q.device)
    _call("icap_mha_scores_f32", _chk(q, f32, "q"), _chk(k, f32, "k"), _opt(mask, f32, "mask"),
          probs.data_ptr(), Bf, T, S, N, Dk, S if softmax_segment is None else int(softmax_segment),
          current_stream())
    return probs


def mha_combine(probs, v):
    """probs [Bf,N,T,S], v [Bf,S,N,Dv] -> [Bf,T,N,Dv]."""
    Bf, N, T, S = probs.shape
    Dv = v.shape[-1]
    if v.shape != (Bf, S, N, Dv):
        raise ValueError("mha_combine: v must be [Bf,S,N,Dv]")
    out = torch.empty(Bf, T, N, Dv, dtype=f32, device=probs.device)
    _call("icap_mha_combine_f32", _chk(probs, f32, "probs"), _chk(v, f32, "v"), out.data_ptr(), Bf, T,
          S, N, Dv, current_stream())
    return out


def mha_cross_step(q, kv, G, rows_per_group, row_stride_g, row_stride_k, heads, key_dim, mask=None):
    """q [N, heads*key_dim], kv [G, L, 2*heads*key_dim] -> combined values [N, heads*key_dim].
    mask: attention_mask values [G, rows_per_group, L] or [rows_per_group, L] (one block for
    every group), applied as (1 - mask) * -1e9 on the scores."""
    L = kv.shape[1]
    out = torch.empty_like(q)
    per_group = 0
    if mask is not None:
        if tuple(mask.shape) == (G, rows_per_group, L):
            per_group = 1
        elif tuple(mask.shape) != (rows_per_group, L):
            raise ValueError("mha_cross_step: mask must be [G,R,L] or [R,L]")
    _call("icap_mha_cross_step_f32", _chk(q, f32, "q"), _chk(kv, f32, "kv"), out.data_ptr(),
              int(G), int(rows_per_group), int(row_stride_g), int(row_stride_k), int(L), int(heads),
              int(key_dim), _opt(mask, f32, "mask"), per_group, current_stream())
    return out


_beam_ws = {}


def _beam_workspace(B, W, device):
    """Device scratch for the beam step, cached per (device, stream) and grown on demand."""
    need = int(_lib.load().icap_beam_step_workspace_bytes(B, W))
    key = (device, current_stream())
    ws = _beam_ws.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(max(need, 1 << 16), dtype=torch.uint8, device=device)
        _beam_ws[key] = ws
    return ws


def beam_step(logits, log_probs, finished, lengths, time, eos_token, length_penalty_weight,
              start_offset=0, input_is_log_prob=False, beam_width=None):
    """logits [B,R,V], state [B,R] -> (scores, ids, parents, next_log_probs, next_finished(u8),
    next_lengths), each [B,W]; W = beam_width (default: the R rows of the input)."""
    B, R, V = logits.shape
    W = R if beam_width is None else int(beam_width)
    dev = logits.device
    scores = torch.empty(B, W, dtype=f32, device=dev)
    ids = torch.empty(B, W, dtype=i32, device=dev)
    parents = torch.empty(B, W, dtype=i32, device=dev)
    nlp = torch.empty(B, W, dtype=f32, device=dev)
    nfin = torch.empty(B, W, dtype=u8, device=dev)
    nlen = torch.empty(B, W, dtype=i32, device=dev)
    ws = _beam_workspace(B, R, dev)
    _call("icap_beam_step_rows_f32", _chk(logits, f32, "logits"),
              _chk(log_probs, f32, "log_probs"), _chk(finished, u8, "finished"),
              _chk(lengths, i32, "lengths"), int(time), int(start_offset), int(eos_token),
              float(length_penalty_weight), 1 if input_is_log_prob else 0, scores.data_ptr(),
              ids.data_ptr(), parents.data_ptr(), nlp.data_ptr(), nfin.data_ptr(), nlen.data_ptr(),
              B, R, W, V, ws.data_ptr(), ws.numel(), current_stream())
    return scores, ids, parents, nlp, nfin, nlen


def sample_categorical(logits, u, step=0, row_len=None, start_token=-1, eos_token=-1, pad_token=-1,
                       out=None):
    """One step of the sequence sampler (reference utils/agent_utils.py:81-125): a draw per row
    from softmax(logits with f32.min added at the start / eos / pad columns) by inversion of
    the cumulative distribution at u [N] in [0,1); with row_len [N] the forcing by position
    (eos where step + 1 == row_len, pad where step >= row_len).  Returns int32 [N]."""
    N, V = logits.shape
    if out is None:
        out = torch.empty(N, dtype=i32, device=logits.device)
    _call("icap_sample_categorical_f32", _chk(logits, f32, "logits"), _chk(u, f32, "u"),
              _opt(row_len, i32, "row_len"), int(step), int(start_token), int(eos_token),
              int(pad_token), _chk(out, i32, "out"), N, V, current_stream())
    return out


def backtrack(ids, parents, last_scores, last_finished):
    """ids/parents [T,B,W] -> (tokens[B,T] i32, best_beam[B] i32, best_score[B] f32)."""
    T, B, W = ids.shape
    dev = ids.device
    tokens = torch.empty(B, T, dtype=i32, device=dev)
    best = torch.empty(B, dtype=i32, device=dev)
    best_s = torch.empty(B, dtype=f32, device=dev)
    _call("icap_backtrack", _chk(ids, i32, "ids"), _chk(parents, i32, "parents"),
              _chk(last_scores, f32, "last_scores"), _chk(last_finished, u8, "last_finished"),
              tokens.data_ptr(), best.data_ptr(), best_s.data_ptr(), T, B, W, current_stream())
    return tokens, best, best_s


def beam_scores(logits, log_probs, finished, lengths, eos_token, length_penalty_weight,
                input_is_log_prob=False):
    """(scores[B,W,V], totals[B,W,V]) of one beam-search step before the successor choice."""
    B, W, V = logits.shape
    scores = torch.empty(B, W, V, dtype=f32, device=logits.device)
    totals = torch.empty(B, W, V, dtype=f32, device=logits.device)
    _call("icap_beam_scores_f32", _chk(logits, f32, "logits"), _chk(log_probs, f32, "log_probs"),
              _chk(finished, u8, "finished"), _chk(lengths, i32, "lengths"), int(eos_token),
              float(length_penalty_weight), 1 if input_is_log_prob else 0, scores.data_ptr(),
              totals.data_ptr(), B, W, V, current_stream())
    return scores, totals


def beam_gather(totals, finished, lengths, word_indices, eos_token):
    """The step after the successor choice: word_indices[B,K] (flat into W*V) ->
    (ids, parents, next_log_probs, next_finished(u8), next_lengths), all [B,K]."""
    B, W, V = totals.shape
    K = word_indices.shape[-1]
    dev = totals.device
    ids = torch.empty(B, K, dtype=i32, device=dev)
    parents = torch.empty(B, K, dtype=i32, device=dev)
    nlp = torch.empty(B, K, dtype=f32, device=dev)
    nfin = torch.empty(B, K, dtype=u8, device=dev)
    nlen = torch.empty(B, K, dtype=i32, device=dev)
    _call("icap_beam_gather_f32", _chk(totals, f32, "totals"), _chk(finished, u8, "finished"),
              _chk(lengths, i32, "lengths"), _chk(word_indices, i32, "word_indices"),
              int(eos_token), ids.data_ptr(), parents.data_ptr(), nlp.data_ptr(), nfin.data_ptr(),
              nlen.data_ptr(), B, W, V, K, current_stream())
    return ids, parents, nlp, nfin, nlen


def mask_probs(probs, eos_token, finished):
    """probs[..., V] with rows of finished beams replaced by the all-mass-on-eos row."""
    V = probs.shape[-1]
    N = probs.numel() // V
    out = torch.empty_like(probs)
    if N:
        _call("icap_mask_probs_f32", _chk(probs, f32, "probs"), _chk(finished, u8, "finished"),
                  out.data_ptr(), int(eos_token), N, V, current_stream())
    return out


def hyp_score(log_probs, lengths, length_penalty_weight):
    """log_probs / length_penalty(lengths) elementwise (same shapes); log_probs None ->
    the penalty itself."""
    out = torch.empty(lengths.shape, dtype=f32, device=lengths.device)
    if lengths.numel():
        _call("icap_hyp_score_f32", _opt(log_probs, f32, "log_probs"),
                  _chk(lengths, i32, "sequence_lengths"), float(length_penalty_weight),
                  out.data_ptr(), lengths.numel(), current_stream())
    return out


def batch_gather(a, idx):
    """out[b, y] = a[b, idx[b, y]] (float32 [B,X], int32 [B,Y] -> float32 [B,Y])."""
    B, X = a.shape
    Y = idx.shape[-1]
    out = torch.empty(B, Y, dtype=f32, device=a.device)
    if B and Y:
        _call("icap_batch_gather_f32", _chk(a, f32, "a"), _chk(idx, i32, "b"), out.data_ptr(),
                  B, X, Y, current_stream())
    return out


# ---- bf16-operand mode ---------------------------------------------------------------

class PackedWeight:
    """A weight matrix packed for the bf16 kernels: bf16, transposed to [N_pad, K]."""

    def __init__(self, K, N, device, gate_units=0):
        self.K, self.N, self.gate_units = int(K), int(N), int(gate_units)
        nbytes = int(_lib.load().icap_packed_weight_bytes(self.K, self.N))
        self.buf = torch.empty(nbytes, dtype=torch.uint8, device=device)

    def fill(self, kernel, k_offset=0):
        """Packs rows of `kernel` [K_src, N] at k positions [k_offset, k_offset + K_src)."""
        K_src, N = kernel.shape
        if N != self.N:
            raise ValueError("PackedWeight.fill: kernel has %d columns, expected %d" % (N, self.N))
        _call("icap_pack_weight_bf16", _chk(kernel, f32, "kernel"), self.buf.data_ptr(),
                  K_src, N, int(k_offset), self.K, self.gate_units, current_stream())
        return self


def pack_weight(kernel, gate_units=0):
    """kernel [K, N] fp32 -> PackedWeight."""
    return PackedWeight(kernel.shape[0], kernel.shape[1], kernel.device, gate_units).fill(kernel)


def dense_bf16(x, packed, bias=None, act=ACT_NONE):
    """act(bf16(x) . bf16(W) + bias) with fp32 accumulation; K must be a multiple of 64."""
    if x.shape[-1] != packed.K:
        raise ValueError("dense_bf16: input last dim %d != packed K %d" % (x.shape[-1], packed.K))
    M = x.numel() // packed.K
    out = torch.empty(*x.shape[:-1], packed.N, dtype=f32, device=x.device)
    if M:
        _call("icap_dense_bf16", _chk(x, f32, "x"), packed.buf.data_ptr(), _opt(bias, f32, "bias"),
                  out.data_ptr(), M, packed.K, packed.N, act, current_stream())
    return out


def lstm_step_bf16(ctx, h, c, packed, bias, emb_proj=None, ids=None):
    """One LSTM cell step on [ctx | h] with bf16 operands; `packed` holds kernel[0:C] and the
    recurrent kernel with interleaved gates (PackedWeight(C + H, 4H, gate_units=H))."""
    N, C = ctx.shape
    H = h.shape[-1]
    if packed.K != C + H or packed.N != 4 * H or packed.gate_units != H:
        raise ValueError("lstm_step_bf16: packed weight does not match C=%d H=%d" % (C, H))
    h_out = torch.empty_like(h)
    c_out = torch.empty_like(c)
    V = emb_proj.shape[0] if emb_proj is not None else 0
    _call("icap_lstm_step_bf16", _chk(ctx, f32, "ctx"), _chk(h, f32, "h"), _chk(c, f32, "c"),
              packed.buf.data_ptr(), _opt(bias, f32, "bias"), _opt(emb_proj, f32, "emb_proj"),
              _opt(ids, i32, "ids"), h_out.data_ptr(), c_out.data_ptr(), N, C, H, V, current_stream())
    return h_out, c_out


def gru_step_bf16(ctx, h, packed, bias4, emb_proj=None, ids=None):
    """One GRU cell step on [ctx | h] with bf16 operands; `packed` holds the four-column
    operands of gru_pack (kernel4[0:C] then recurrent4) with interleaved gates
    (PackedWeight(C + H, 4H, gate_units=H)); returns h'."""
    N, C = ctx.shape
    H = h.shape[-1]
    if packed.K != C + H or packed.N != 4 * H or packed.gate_units != H:
        raise ValueError("gru_step_bf16: packed weight does not match C=%d H=%d" % (C, H))
    h_out = torch.empty_like(h)
    V = emb_proj.shape[0] if emb_proj is not None else 0
    _call("icap_gru_step_bf16", _chk(ctx, f32, "ctx"), _chk(h, f32, "h"), packed.buf.data_ptr(),
              _opt(bias4, f32, "bias4"), _opt(emb_proj, f32, "emb_proj"), _opt(ids, i32, "ids"),
              h_out.data_ptr(), N, C, H, V, current_stream())
    return h_out


# ---- fused vocab projection + beam step ----------------------------------------------

def dense_topk(x, kernel, bias, eos_token, top_k):
    """Logits x . kernel + bias reduced on the fly to per-row records (see icap.h):
    returns (records [M, half_tiles, 2 + 2*top_k] float32, eos_logit [M] float32)."""
    K, N = kernel.shape
    M = x.numel() // K
    lib = _lib.load()
    ht, rl = int(lib.icap_dense_topk_half_tiles(N)), int(lib.icap_dense_topk_record_floats(top_k))
    if rl == 0:
        raise ValueError("dense_topk: top_k must be 1, 3 or 5")
    rec = torch.empty(M, ht, rl, dtype=f32, device=x.device)
    eos_logit = torch.zeros(M, dtype=f32, device=x.device)
    _call("icap_dense_topk_f32", _chk(x, f32, "x"), _chk(kernel, f32, "kernel"),
              _opt(bias, f32, "bias"), M, K, N, int(eos_token), int(top_k), rec.data_ptr(),
              eos_logit.data_ptr(), current_stream())
    return rec, eos_logit


def beam_step_from_records_wide(rec, eos_logit, log_probs, finished, lengths, time, eos_token,
                                length_penalty_weight, V, compact_first=False):
    """The records step for beams up to 16 wide on records of 5 logits per half tile.
    Returns the six outputs of beam_step plus the overflow word (int32 [1]; non-zero: the
    records could not prove the result, repeat the step on full logits)."""
    B, W = log_probs.shape
    dev = log_probs.device
    scores = torch.empty(B, W, dtype=f32, device=dev)
    ids = torch.empty(B, W, dtype=i32, device=dev)
    parents = torch.empty(B, W, dtype=i32, device=dev)
    nlp = torch.empty(B, W, dtype=f32, device=dev)
    nfin = torch.empty(B, W, dtype=u8, device=dev)
    nlen = torch.empty(B, W, dtype=i32, device=dev)
    overflow = torch.zeros(1, dtype=i32, device=dev)
    HT, RL = rec.shape[1], rec.shape[2]
    if RL != 12:
        raise ValueError("beam_step_from_records_wide: records must hold 5 logits per half tile")
    _call("icap_beam_step_from_records_wide_f32", _chk(rec, f32, "rec"), _chk(eos_logit, f32, "eos_logit"),
          HT, 5, _chk(log_probs, f32, "log_probs"), _chk(finished, u8, "finished"),
          _chk(lengths, i32, "lengths"), int(time), int(eos_token), float(length_penalty_weight),
          scores.data_ptr(), ids.data_ptr(), parents.data_ptr(), nlp.data_ptr(), nfin.data_ptr(),
          nlen.data_ptr(), B, W, int(V), overflow.data_ptr(), 1 if compact_first else 0,
          current_stream())
    return scores, ids, parents, nlp, nfin, nlen, overflow


def beam_step_from_records(rec, eos_logit, log_probs, finished, lengths, time, eos_token,
                           length_penalty_weight, V):
    """beam_step on the records of dense_topk; state tensors [B, W]."""
    B, W = log_probs.shape
    M, ht, rl = rec.shape
    if M != B * W:
        raise ValueError("beam_step_from_records: %d record rows for B*W = %d" % (M, B * W))
    top_k = (rl - 2) // 2
    dev = rec.device
    scores = torch.empty(B, W, dtype=f32, device=dev)
    ids = torch.empty(B, W, dtype=i32, device=dev)
    parents = torch.empty(B, W, dtype=i32, device=dev)
    nlp = torch.empty(B, W, dtype=f32, device=dev)
    nfin = torch.empty(B, W, dtype=u8, device=dev)
    nlen = torch.empty(B, W, dtype=i32, device=dev)
    _call("icap_beam_step_from_records_f32", _chk(rec, f32, "records"),
              _chk(eos_logit, f32, "eos_logit"), ht, top_k, _chk(log_probs, f32, "log_probs"),
              _chk(finished, u8, "finished"), _chk(lengths, i32, "lengths"), int(time),
              int(eos_token), float(length_penalty_weight), scores.data_ptr(), ids.data_ptr(),
              parents.data_ptr(), nlp.data_ptr(), nfin.data_ptr(), nlen.data_ptr(), B, W, int(V),
              current_stream())
    return scores, ids, parents, nlp, nfin, nlen


# ---- split mode: fp32-accurate products on the bf16 matrix pipe -----------------------

class PackedSplitWeight:
    """A weight matrix split into three bf16 planes (hi + mid + lo == the fp32 value) and
    packed [3, N_pad, K] for the split-mode kernels."""

    def __init__(self, K, N, device, gate_units=0):
        self.K, self.N, self.gate_units = int(K), int(N), int(gate_units)
        nbytes = int(_lib.load().icap_packed_split_bytes(self.K, self.N))
        self.buf = torch.empty(nbytes, dtype=torch.uint8, device=device)

    def fill(self, kernel, k_offset=0):
        K_src, N = kernel.shape
        if N != self.N:
            raise ValueError("PackedSplitWeight.fill: kernel has %d columns, expected %d" % (N, self.N))
        _call("icap_pack_weight_split", _chk(kernel, f32, "kernel"), self.buf.data_ptr(),
                  K_src, N, int(k_offset), self.K, self.gate_units, current_stream())
        return self


def pack_weight_split(kernel, gate_units=0):
    return PackedSplitWeight(kernel.shape[0], kernel.shape[1], kernel.device, gate_units).fill(kernel)


def dense_split(x, packed, bias=None, act=ACT_NONE):
    """act(x . W + bias) with fp32-accurate products on the bf16 matrix pipe; K % 32 == 0."""
    if x.shape[-1] != packed.K:
        raise ValueError("dense_split: input last dim %d != packed K %d" % (x.shape[-1], packed.K))
    M = x.numel() // packed.K
    out = torch.empty(*x.shape[:-1], packed.N, dtype=f32, device=x.device)
    if M:
        _call("icap_dense_split", _chk(x, f32, "x"), packed.buf.data_ptr(), _opt(bias, f32, "bias"),
                  out.data_ptr(), M, packed.K, packed.N, act, current_stream())
    return out


def dense_topk_split(x, packed, bias, eos_token, top_k):
    """dense_topk with the split-mode product; same records."""
    M = x.numel() // packed.K
    lib = _lib.load()
    ht, rl = int(lib.icap_dense_topk_half_tiles(packed.N)), int(lib.icap_dense_topk_record_floats(top_k))
    if rl == 0:
        raise ValueError("dense_topk_split: top_k must be 1, 3 or 5")
    rec = torch.empty(M, ht, rl, dtype=f32, device=x.device)
    eos_logit = torch.zeros(M, dtype=f32, device=x.device)
    _call("icap_dense_topk_split", _chk(x, f32, "x"), packed.buf.data_ptr(), _opt(bias, f32, "bias"),
              M, packed.K, packed.N, int(eos_token), int(top_k), rec.data_ptr(), eos_logit.data_ptr(),
              current_stream())
    return rec, eos_logit


def lstm_step_split(ctx, h, c, packed, bias, emb_proj=None, ids=None):
    """One LSTM cell step on [ctx | h] in split mode; `packed` = PackedSplitWeight(C + H, 4H,
    gate_units=H) filled with kernel[0:C] at 0 and the recurrent kernel at C."""
    N, C = ctx.shape
    H = h.shape[-1]
    if packed.K != C + H or packed.N != 4 * H or packed.gate_units != H:
        raise ValueError("lstm_step_split: packed weight does not match C=%d H=%d" % (C, H))
    h_out = torch.empty_like(h)
    c_out = torch.empty_like(c)
    V = emb_proj.shape[0] if emb_proj is not None else 0
    _call("icap_lstm_step_split", _chk(ctx, f32, "ctx"), _chk(h, f32, "h"), _chk(c, f32, "c"),
              packed.buf.data_ptr(), _opt(bias, f32, "bias"), _opt(emb_proj, f32, "emb_proj"),
              _opt(ids, i32, "ids"), h_out.data_ptr(), c_out.data_ptr(), N, C, H, V, current_stream())
    return h_out, c_out
