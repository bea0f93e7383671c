"""Guards the numpy oracle against transcription slips by recomputing each
parity-unpinned stage with independent torch-CPU primitives.  The reference has no
test for these stages, so this is a cross-check, not a reference-derived pin."""
import numpy as np
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
