"""Seeded synthetic weights and inputs of the reference model's shapes (no checkpoints
or datasets are available offline): Keras default initialisers for every layer of the
path and ReLU-Gaussian CNN features."""
from __future__ import annotations

import numpy as np


def _glorot(rng, fan_in, fan_out):
    lim = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=(fan_in, fan_out)).astype(np.float32)


def make_params(seed, F, D, E, H, V, U=None):
    """Glorot-uniform Dense / LSTM kernels (the decoder asks for a glorot_uniform
    recurrent kernel, reference decoder/lstm_recurrent_decoder.py:23,36), zero biases
    with the LSTM forget-gate slice at 1 (Keras unit_forget_bias), embedding
    U(-0.05, 0.05).  Keys are the engine's parameter names."""
    U = U or H
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
    p["lstm_kernel"] = _glorot(rng, D + E, 4 * H)
    p["lstm_recurrent_kernel"] = _glorot(rng, H, 4 * H)
    b = np.zeros(4 * H, np.float32)
    b[H:2 * H] = 1.0
    p["lstm_bias"] = b
    p["fc1_kernel"] = _glorot(rng, H, H)
    p["fc1_bias"] = np.zeros(H, np.float32)
    p["fc2_kernel"] = _glorot(rng, H, V)
    p["fc2_bias"] = np.zeros(V, np.float32)
    return p


def add_gru_params(p, seed):
    """GRU variables for the GRU decoder (reference decoder/recurrent_decoder.py:30-34) beside
    the LSTM ones, from a random stream of their own: Glorot-uniform kernel [D+E,3H] and
    recurrent kernel [H,3H], zero bias [2,3H]."""
    rng = np.random.default_rng([seed, 7919])
    CE = p["lstm_kernel"].shape[0]
    H = p["lstm_recurrent_kernel"].shape[0]
    q = dict(p)
    q["gru_kernel"] = _glorot(rng, CE, 3 * H)
    q["gru_recurrent_kernel"] = _glorot(rng, H, 3 * H)
    q["gru_bias"] = np.zeros((2, 3 * H), np.float32)
    return q


def make_transformer_params(seed, F, D, V, hidden, layers, heads, key_dim, filter_size):
    """Weights of the transformer decoder path under the TransformerEngine's names:
    Glorot-uniform kernels (attention projections as flattened [in, heads*key_dim] matrices),
    zero biases, LayerNorm gamma 1 / beta 0, embedding U(-0.05, 0.05)."""
    rng = np.random.default_rng([seed, 104729])
    HD = heads * key_dim
    p = {"enc_kernel": _glorot(rng, F, D), "enc_bias": np.zeros(D, np.float32),
         "t_embedding": rng.uniform(-0.05, 0.05, size=(V, hidden)).astype(np.float32),
         "t_fc2_kernel": _glorot(rng, hidden, V), "t_fc2_bias": np.zeros(V, np.float32),
         "out_ln_gamma": np.ones(hidden, np.float32), "out_ln_beta": np.zeros(hidden, np.float32)}
    for l in range(layers):
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


def make_images(seed, B, L, F):
    """max(0, N(0,1)) * 0.5: non-negative, about half zeros, like post-ReLU CNN maps."""
    rng = np.random.default_rng(seed)
    return (np.maximum(rng.standard_normal((B, L, F)), 0.0) * 0.5).astype(np.float32)


def save_params(path, params):
    """The engine's own weight container: an .npz of float32 arrays in Keras layouts."""
    np.savez(path, **{k: np.asarray(v, np.float32) for k, v in params.items()})


def load_params(path):
    with np.load(path, allow_pickle=False) as z:
        return {k: z[k] for k in z.files}
