"""Guards the numpy oracle against transcription slips by recomputing each
parity-unpinned stage with independent torch-CPU primitives.  The reference has no
test for these stages, so this is a cross-check, not a reference-derived pin."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import caption_oracle as co

DIMS = dict(F=24, D=8, E=8, H=12, V=37)


def _params(seed=0):
    return co.perturb_biases(co.make_params(seed, **DIMS), seed + 100)


def _t(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def test_encoder_matches_torch_linear():
    p = _params()
    img = co.make_images(1, 3, 5, DIMS["F"])
    ref = F.relu(F.linear(_t(img), _t(p["enc_kernel"]).T, _t(p["enc_bias"]))).numpy()
    np.testing.assert_allclose(co.mlp_encoder(img, p["enc_kernel"], p["enc_bias"]), ref,
                               rtol=1e-5, atol=1e-6)


def test_lstm_cell_matches_torch_lstmcell():
    p = _params(2)
    rng = np.random.default_rng(5)
    n, H = 7, DIMS["H"]
    x = rng.standard_normal((n, DIMS["D"] + DIMS["E"])).astype(np.float32)
    h = rng.standard_normal((n, H)).astype(np.float32) * 0.3
    c = rng.standard_normal((n, H)).astype(np.float32) * 0.3
    cell = torch.nn.LSTMCell(x.shape[1], H)
    with torch.no_grad():
        cell.weight_ih.copy_(_t(p["lstm_kernel"]).T)        # gate order i,f,g,o in both
        cell.weight_hh.copy_(_t(p["lstm_recurrent_kernel"]).T)
        cell.bias_ih.copy_(_t(p["lstm_bias"]))
        cell.bias_hh.zero_()
        h_ref, c_ref = cell(_t(x), (_t(h), _t(c)))
    out, (h1, c1) = co.lstm_cell(x, h, c, p["lstm_kernel"], p["lstm_recurrent_kernel"],
                                 p["lstm_bias"])
    np.testing.assert_allclose(h1, h_ref.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(c1, c_ref.numpy(), rtol=1e-5, atol=1e-6)
    assert out is h1


def test_gru_cell_matches_torch_grucell():
    """Keras reset_after=True GRU is torch's GRUCell with the gates reordered
    (Keras z, r, h; torch r, z, n) and the two bias rows as bias_ih / bias_hh."""
    p = co.perturb_biases(co.add_gru_params(_params(2), 2), 3, scale=0.2)
    rng = np.random.default_rng(6)
    n, H = 7, DIMS["H"]
    x = rng.standard_normal((n, DIMS["D"] + DIMS["E"])).astype(np.float32)
    h = rng.standard_normal((n, H)).astype(np.float32) * 0.3

    def reorder(a):                       # columns [z | r | h] -> [r | z | h]
        z, r, c = np.split(a, 3, axis=-1)
        return np.concatenate([r, z, c], axis=-1)

    cell = torch.nn.GRUCell(x.shape[1], H)
    with torch.no_grad():
        cell.weight_ih.copy_(_t(reorder(p["gru_kernel"])).T)
        cell.weight_hh.copy_(_t(reorder(p["gru_recurrent_kernel"])).T)
        cell.bias_ih.copy_(_t(reorder(p["gru_bias"][0])))
        cell.bias_hh.copy_(_t(reorder(p["gru_bias"][1])))
        h_ref = cell(_t(x), _t(h))
    out, (h1,) = co.gru_cell(x, h, p["gru_kernel"], p["gru_recurrent_kernel"], p["gru_bias"])
    np.testing.assert_allclose(h1, h_ref.numpy(), rtol=1e-5, atol=1e-6)
    assert out is h1


def test_gru_decoder_uses_h_as_query_and_single_state():
    p = co.add_gru_params(_params(4), 4)
    rng = np.random.default_rng(7)
    n, L = 3, 4
    feats = np.abs(rng.standard_normal((n, L, DIMS["D"]))).astype(np.float32)
    hidden = co.reset_state_gru(p, n, feats, 1)
    assert len(hidden) == 1
    ids = np.array([[1], [2], [3]], np.int32)
    logits, state, weights, ctx = co.gru_decoder_step(p, ids, feats, hidden)
    assert len(state) == 1 and logits.shape == (n, DIMS["V"]) and weights.shape == (n, L, 1)
    c0, _ = co.bahdanau_attention(feats, hidden[0], p)
    beta = co.sigmoid(co.dense(hidden[0], p["fc3_kernel"], p["fc3_bias"]))
    np.testing.assert_allclose(ctx, c0 * beta, rtol=1e-6)
    tok = co.predict_image_caption_sequence(p, co.make_images(1, 2, L, DIMS["F"]), 2, 5, 3, 4,
                                            cell="gru")
    assert tok.shape == (2, 5)


def test_attention_matches_torch():
    p = _params(3)
    rng = np.random.default_rng(6)
    n, L = 4, 5
    feats = rng.standard_normal((n, L, DIMS["D"])).astype(np.float32)
    q = rng.standard_normal((n, DIMS["H"])).astype(np.float32)
    bias = rng.standard_normal((n, L)).astype(np.float32)
    for b in (None, bias):
        ctx, w = co.bahdanau_attention(feats, q, p, bias=b)
        w1 = F.linear(_t(feats), _t(p["att_W1_kernel"]).T, _t(p["att_W1_bias"]))
        w2 = F.linear(_t(q), _t(p["att_W2_kernel"]).T, _t(p["att_W2_bias"]))[:, None, :]
        s = F.linear(torch.tanh(w1 + w2), _t(p["att_V_kernel"]).T, _t(p["att_V_bias"]))
        if b is not None:
            s = s + _t(b)[..., None]
        a = torch.softmax(s, dim=1)
        c_ref = (a * _t(feats)).sum(1)
        np.testing.assert_allclose(w, a.numpy(), rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(ctx, c_ref.numpy(), rtol=1e-5, atol=1e-6)
        assert w.shape == (n, L, 1)


def test_decoder_step_uses_cell_state_as_query():
    """hidden[-1] is c: changing h alone must not change the attention weights."""
    p = _params(4)
    rng = np.random.default_rng(7)
    n, L = 3, 4
    feats = np.abs(rng.standard_normal((n, L, DIMS["D"]))).astype(np.float32)
    h = rng.standard_normal((n, DIMS["H"])).astype(np.float32)
    c = rng.standard_normal((n, DIMS["H"])).astype(np.float32)
    ids = np.array([[3], [5], [7]], np.int32)
    _, _, w_a, ctx_a = co.lstm_decoder_step(p, ids, feats, [h, c])
    _, _, w_b, ctx_b = co.lstm_decoder_step(p, ids, feats, [h * 2 + 1, c])
    np.testing.assert_array_equal(w_a, w_b)
    np.testing.assert_array_equal(ctx_a, ctx_b)
    _, _, w_c, _ = co.lstm_decoder_step(p, ids, feats, [h, c + 1])
    assert np.abs(w_a - w_c).max() > 0


def test_init_state_mean_and_masked_sum():
    p = _params(5)
    rng = np.random.default_rng(8)
    feats = rng.standard_normal((3, 6, DIMS["D"])).astype(np.float32)
    got = co.feature_proj_state(feats, p["init0_kernel"], p["init0_bias"])
    ref = feats.mean(1) @ p["init0_kernel"] + p["init0_bias"]
    np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-6)
    mask = np.array([[1, 1, 0, 0, 1, 0], [0] * 6, [1] * 6], np.float32)
    got = co.feature_proj_state(feats, p["init0_kernel"], p["init0_bias"], mask=mask)
    ref = (feats * mask[..., None]).sum(1) @ p["init0_kernel"] + p["init0_bias"]
    np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(got[1], p["init0_bias"], rtol=1e-6)


def test_driver_tile_modes_agree_when_layouts_coincide():
    p = _params(6)
    img = co.make_images(9, 1, 5, DIMS["F"])
    a = co.predict_image_caption_sequence(p, img, 3, 6, 3, 4, tile_mode="reference")
    b = co.predict_image_caption_sequence(p, img, 3, 6, 3, 4, tile_mode="repeat")
    np.testing.assert_array_equal(a, b)                     # B = 1
    img = co.make_images(10, 4, 5, DIMS["F"])
    a = co.predict_image_caption_sequence(p, img, 1, 6, 3, 4, tile_mode="reference")
    b = co.predict_image_caption_sequence(p, img, 1, 6, 3, 4, tile_mode="repeat")
    np.testing.assert_array_equal(a, b)                     # W = 1
    assert a.shape == (4, 6) and a.dtype == np.int32


def test_driver_repeat_mode_is_per_image_independent():
    p = _params(7)
    img = co.make_images(11, 5, 5, DIMS["F"])
    full = co.predict_image_caption_sequence(p, img, 3, 7, 3, 4, tile_mode="repeat")
    for b in range(5):
        one = co.predict_image_caption_sequence(p, img[b:b + 1], 3, 7, 3, 4, tile_mode="repeat")
        np.testing.assert_array_equal(full[b], one[0])


def test_hoisted_w1_variant_gives_the_same_tokens():
    p = co.perturb_biases(co.make_params(2, 24, 8, 8, 16, 50), 3)
    img = co.make_images(4, 3, 5, 24)
    a = co.predict_image_caption_sequence(p, img, 3, 6, 3, 4)
    b = co.predict_image_caption_sequence(p, img, 3, 6, 3, 4, hoist_w1=True)
    np.testing.assert_array_equal(a, b)


def test_baseline_searcher_oracle_forces_eos_at_the_target_length():
    """One image (so the tile layout is trivial): in the target-length form the sequence
    starts with the start token, never repeats it, and has eos at the target's eos position
    (or earlier) and from there on; with max_length > 0 the length is max_length."""
    p = _params(5)
    rng = np.random.default_rng(11)
    feats = np.abs(rng.standard_normal((1, 4, DIMS["D"]))).astype(np.float32)
    seqs = np.array([[3, 9, 8, 7, 4, 0, 0, 0]], np.int32)          # length 5
    tok = co.baseline_sequence_beam_search(p, feats, seqs, 2, DIMS["V"], 0, 3, 4, 0)
    assert tok.shape == (1, 8) and tok[0, 0] == 3
    assert (tok[0, 1:] != 3).all()
    first_eos = int(np.argmax(tok[0] == 4))
    assert 1 <= first_eos <= 4 and (tok[0, first_eos:] == 4).all()
    tok5 = co.baseline_sequence_beam_search(p, feats, seqs, 2, DIMS["V"], 5, 3, 4, 0)
    assert tok5.shape == (1, 5) and tok5[0, 0] == 3 and (tok5[0, 1:] != 3).all()
    assert co.sequence_length_fn(seqs, 0).tolist() == [5]


# ---- transformer path: the oracle's restatement against independent torch-CPU primitives ----

def test_layer_norm_matches_torch():
    rng = np.random.default_rng(0)
    x = rng.standard_normal((5, 7, 48)).astype(np.float32) * 3 + 1
    g = (1 + 0.1 * rng.standard_normal(48)).astype(np.float32)
    b = (0.1 * rng.standard_normal(48)).astype(np.float32)
    ref = torch.nn.functional.layer_norm(torch.from_numpy(x), (48,), torch.from_numpy(g),
                                         torch.from_numpy(b), eps=1e-6).numpy()
    np.testing.assert_allclose(co.layer_norm(x, g, b), ref, rtol=2e-5, atol=2e-6)


def test_position_embedding_matches_the_published_formula():
    """sin / cos of pos / 10000^(i / (n - 1)), n = hidden / 2 timescales ("Attention is all
    you need" 3.5 as tensor2tensor lays it out: all sines, then all cosines)."""
    length, hidden = 23, 64
    n = hidden // 2
    pos = torch.arange(length, dtype=torch.float64)[:, None]
    inv = torch.pow(torch.tensor(1.0e4, dtype=torch.float64),
                    -torch.arange(n, dtype=torch.float64) / (n - 1))[None, :]
    ref = torch.cat([torch.sin(pos * inv), torch.cos(pos * inv)], dim=1).numpy()
    got = co.relative_position_embedding(length, hidden)
    assert got.shape == (length, hidden) and got.dtype == np.float32
    np.testing.assert_allclose(got, ref, rtol=0, atol=5e-6)


def test_decoder_self_attention_bias_is_causal():
    b = co.decoder_self_attention_bias(5)
    assert b.shape == (1, 1, 5, 5)
    assert (b[0, 0][np.tril_indices(5)] == 0).all()
    assert (b[0, 0][np.triu_indices(5, 1)] == np.float32(-1e9)).all()


def _mha_params(rng, d_q, d_kv, heads, key_dim, prefix="m_"):
    HD = heads * key_dim
    p = {}
    for name, shape in (("q_kernel", (d_q, HD)), ("k_kernel", (d_kv, HD)), ("v_kernel", (d_kv, HD)),
                        ("o_kernel", (HD, d_q))):
        p[prefix + name] = (rng.standard_normal(shape) / np.sqrt(shape[0])).astype(np.float32)
    for name, n in (("q_bias", HD), ("k_bias", HD), ("v_bias", HD), ("o_bias", d_q)):
        p[prefix + name] = (0.1 * rng.standard_normal(n)).astype(np.float32)
    return p


@pytest.mark.parametrize("causal", [False, True])
def test_multi_head_attention_matches_torch_sdpa(causal):
    """With a 0/1 mask (or none) the oracle's attention is ordinary scaled dot-product attention
    per head."""
    rng = np.random.default_rng(3)
    B, T, S, d_q, d_kv, heads, kd = 2, 5, 5 if causal else 7, 24, 24 if causal else 16, 3, 8
    p = _mha_params(rng, d_q, d_kv, heads, kd)
    q_in = rng.standard_normal((B, T, d_q)).astype(np.float32)
    kv_in = q_in if causal else rng.standard_normal((B, S, d_kv)).astype(np.float32)
    mask = np.tril(np.ones((T, S), np.float32)).reshape(1, 1, T, S) if causal else None
    got = co.multi_head_attention(q_in, kv_in, p, "m_", heads, kd, attention_mask=mask)
    tq, tkv = torch.from_numpy(q_in), torch.from_numpy(kv_in)
    proj = lambda x, k: (x @ torch.from_numpy(p["m_" + k + "_kernel"]) + torch.from_numpy(p["m_" + k + "_bias"]))
    split = lambda x: x.reshape(x.shape[0], x.shape[1], heads, kd).permute(0, 2, 1, 3)
    o = torch.nn.functional.scaled_dot_product_attention(
        split(proj(tq, "q")), split(proj(tkv, "k")), split(proj(tkv, "v")), is_causal=causal)
    o = o.permute(0, 2, 1, 3).reshape(B, T, heads * kd)
    ref = (o @ torch.from_numpy(p["m_o_kernel"]) + torch.from_numpy(p["m_o_bias"])).numpy()
    np.testing.assert_allclose(got, ref, rtol=2e-4, atol=2e-5)


def _mha_general_params(rng, d_q, d_k, d_v, heads, kd, vd, out):
    p = {}
    for n, (i, o) in (("q", (d_q, heads * kd)), ("k", (d_k, heads * kd)), ("v", (d_v, heads * vd)),
                      ("o", (heads * vd, out))):
        p["g_" + n + "_kernel"] = (rng.standard_normal((i, o)) / np.sqrt(i)).astype(np.float32)
        p["g_" + n + "_bias"] = (0.1 * rng.standard_normal(o)).astype(np.float32)
    return p


def test_general_multi_head_attention_reduces_to_the_plain_form_and_to_torch():
    """The oracle's general restatement (einsum equations rebuilt as the reference builds them)
    equals the key = value restatement on a 3-D call, and torch's scaled dot-product attention
    with a key tensor of its own and value_dim != key_dim."""
    rng = np.random.default_rng(5)
    B, T, S, d, heads, kd, vd = 2, 4, 6, 16, 3, 8, 12
    q_in = rng.standard_normal((B, T, d)).astype(np.float32)
    v_in = rng.standard_normal((B, S, d)).astype(np.float32)
    k_in = rng.standard_normal((B, S, 20)).astype(np.float32)
    p = _mha_general_params(rng, d, d, d, heads, kd, kd, d)
    a, _ = co.multi_head_attention_general(q_in, v_in, p, "g_", heads, kd)
    b = co.multi_head_attention(q_in, v_in, p, "g_", heads, kd)
    np.testing.assert_allclose(a, b, rtol=1e-6, atol=1e-6)
    p = _mha_general_params(rng, d, 20, d, heads, kd, vd, 10)
    got, scores = co.multi_head_attention_general(q_in, v_in, p, "g_", heads, kd, key=k_in, value_dim=vd,
                                                  output_shape=10)
    assert got.shape == (B, T, 10) and scores.shape == (B, heads, T, S)
    np.testing.assert_allclose(scores.sum(-1), 1.0, rtol=1e-5)
    t = lambda x: torch.from_numpy(x)
    proj = lambda x, n, dd: (t(x) @ t(p["g_" + n + "_kernel"]) + t(p["g_" + n + "_bias"])).reshape(
        x.shape[0], x.shape[1], heads, dd).permute(0, 2, 1, 3)
    o = torch.nn.functional.scaled_dot_product_attention(proj(q_in, "q", kd), proj(k_in, "k", kd),
                                                         proj(v_in, "v", vd))
    ref = (o.permute(0, 2, 1, 3).reshape(B, T, heads * vd) @ t(p["g_o_kernel"]) + t(p["g_o_bias"])).numpy()
    np.testing.assert_allclose(got, ref, rtol=2e-4, atol=2e-5)


def test_general_multi_head_attention_shapes_of_the_reference_docstring():
    """The two worked examples in the reference's class docstring (attention/attention.py:
    202-215): target [B,8,16], source [B,4,16], 2 heads of 2 -> output (B,8,16), weights
    (B,2,8,4); 2-D attention over axes (2,3) of a [B,5,3,4,16] input -> (B,5,3,4,16).  With two
    attention axes the soft-max covers the last key axis only (Softmax(axis=-1), `:457-458`), so
    the weights of a query position sum to the size of the other key axis."""
    rng = np.random.default_rng(6)
    B = 2
    p = _mha_general_params(rng, 16, 16, 16, 2, 2, 2, 16)
    target = rng.standard_normal((B, 8, 16)).astype(np.float32)
    source = rng.standard_normal((B, 4, 16)).astype(np.float32)
    out, w = co.multi_head_attention_general(target, source, p, "g_", 2, 2)
    assert out.shape == (B, 8, 16) and w.shape == (B, 2, 8, 4)
    x = rng.standard_normal((B, 5, 3, 4, 16)).astype(np.float32)
    out, w = co.multi_head_attention_general(x, x, p, "g_", 2, 2, attention_axes=(2, 3))
    assert out.shape == (B, 5, 3, 4, 16) and w.shape == (B, 5, 2, 3, 4, 3, 4)
    np.testing.assert_allclose(w.sum(axis=(-1, -2)), 3.0, rtol=1e-5)


def test_attention_with_another_activation_matches_torch():
    rng = np.random.default_rng(8)
    N, L, F, U = 3, 5, 6, 4
    p = {"att_W1_kernel": rng.standard_normal((F, U)).astype(np.float32), "att_W1_bias": np.zeros(U, np.float32),
         "att_W2_kernel": rng.standard_normal((7, U)).astype(np.float32), "att_W2_bias": np.zeros(U, np.float32),
         "att_V_kernel": rng.standard_normal((U, 1)).astype(np.float32), "att_V_bias": np.zeros(1, np.float32)}
    f = rng.standard_normal((N, L, F)).astype(np.float32)
    h = rng.standard_normal((N, 7)).astype(np.float32)
    ctx, w = co.bahdanau_attention(f, h, p, activation=lambda x: np.maximum(x, 0))
    tf_, th = torch.from_numpy(f), torch.from_numpy(h)
    hid = torch.relu(tf_ @ torch.from_numpy(p["att_W1_kernel"]) + (th @ torch.from_numpy(p["att_W2_kernel"]))[:, None])
    tw = torch.softmax(hid @ torch.from_numpy(p["att_V_kernel"]), dim=1)
    np.testing.assert_allclose(w, tw.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(ctx, (tw * tf_).sum(1).numpy(), rtol=1e-5, atol=1e-6)


def test_bias_passed_as_mask_gives_a_uniform_causal_mean():
    """The reference hands the additive causal bias to `attention_mask` (transform_decoder.py:
    139 -> attention.py:471-473): visible positions get -1e9 added in fp32, which absorbs the
    scores, hidden ones about -1e18 - the weights become uniform over the visible prefix."""
    rng = np.random.default_rng(4)
    B, T, d, heads, kd = 2, 6, 16, 4, 4
    p = _mha_params(rng, d, d, heads, kd)
    x = rng.standard_normal((B, T, d)).astype(np.float32)
    got = co.multi_head_attention(x, x, p, "m_", heads, kd,
                                  attention_mask=co.decoder_self_attention_bias(T))
    v = x @ p["m_v_kernel"] + p["m_v_bias"]                       # [B,T,HD]
    mean = np.cumsum(v, axis=1) / np.arange(1, T + 1, dtype=np.float32)[None, :, None]
    ref = mean @ p["m_o_kernel"] + p["m_o_bias"]
    np.testing.assert_allclose(got, ref, rtol=2e-4, atol=2e-5)


def test_transformer_step_equals_full_prefix_recomputation():
    """Growing the prefix one token at a time gives, at every position, the output the stack
    computes for that position of the full sequence (what an incremental engine relies on)."""
    dims = dict(F=12, D=8, V=30, hidden=16, n_layers=2, heads=16, key_dim=4, filter_size=32)
    p = co.perturb_transformer_params(co.make_transformer_params(1, **dims), 2)
    rng = np.random.default_rng(5)
    N, L, T = 3, 4, 5
    feats = rng.standard_normal((N, L, dims["D"])).astype(np.float32)
    ids = rng.integers(0, dims["V"], size=(N, T)).astype(np.int32)
    full = None
    hidden = [np.zeros((N, 1), np.int32)]
    for t in range(T):
        logits, hidden, _, last = co.transform_decoder_step(p, ids[:, t], feats, hidden,
                                                            dims["n_layers"], dims["heads"], dims["key_dim"])
        x = p["t_embedding"][hidden[0]]
        x = x + co.relative_position_embedding(hidden[0].shape[1], dims["hidden"])[None]
        full = co.transformer_decoder_stack(p, x, feats, dims["n_layers"], dims["heads"], dims["key_dim"])
        np.testing.assert_allclose(last, full[:, -1], rtol=1e-5, atol=1e-6)
    assert hidden[0].shape == (N, T + 1) and (hidden[0][:, 0] == 0).all()
    # earlier positions are unchanged by later tokens (causality)
    x_short = p["t_embedding"][hidden[0][:, :3]] + co.relative_position_embedding(3, dims["hidden"])[None]
    short = co.transformer_decoder_stack(p, x_short, feats, dims["n_layers"], dims["heads"], dims["key_dim"])
    np.testing.assert_allclose(short, full[:, :3], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("W", [1, 3])
def test_torch_cpu_reference_equals_numpy_oracle(W):
    """The torch-CPU op-for-op loop that the benchmark times as the CPU baseline decodes the
    same tokens as the numpy oracle (reference tile order, peaked weights so that captions
    differ and end tokens occur)."""
    import torch
    from oracle import torch_cpu_reference as tr
    dims = dict(F=24, D=8, E=8, H=12, V=37)
    p = co.perturb_biases(co.make_peaked_params(3, **dims), 4)
    img = co.make_images(5, 6, 5, dims["F"])
    ref = co.predict_image_caption_sequence(p, img, W, 7, 3, 4, 1.0, "reference")
    got = tr.predict_image_caption_sequence(tr.to_torch(p), torch.from_numpy(img), W, 7, 3, 4, 1.0)
    np.testing.assert_array_equal(got.numpy(), ref)
