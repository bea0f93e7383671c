"""Transformer decoder path (reference decoder/transform_decoder.py, attention/attention.py,
decoder/ffn_layer.py, embedding/position_embedding.py) vs the oracle.

No reference test, fixture or driver covers this decoder: parity unpinned; the checker is the
oracle's full-prefix restatement of the code as written (heads = hidden_size x key_dim =
num_heads, bias-as-mask, keys / values from the un-normalised input), which also validates the
engine's incremental form (per-row key / value caches).  Bars as for the recurrent path: 1e-5
relative on activations and log-probs (a few 1e-5 after many layers), identical tokens."""
import numpy as np
import pytest
import torch

from oracle import caption_oracle as co
from tests.helpers import assert_close, dev, np_of, to_dev

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from image_caption_amd import ops as o
    return o


@pytest.mark.parametrize("length,hidden", [(5, 8), (22, 512), (3, 36)])
def test_position_table(ops, length, hidden):
    got = np_of(ops.position_table(length, hidden, dev()))
    ref = co.relative_position_embedding(length, hidden)
    np.testing.assert_allclose(got, ref, rtol=0, atol=2e-6)


def test_embed_pos(ops):
    rng = np.random.default_rng(0)
    table = rng.standard_normal((11, 12)).astype(np.float32)
    ids = rng.integers(0, 11, size=7).astype(np.int32)
    row = rng.standard_normal(12).astype(np.float32)
    got = np_of(ops.embed_pos(to_dev(table), to_dev(ids), to_dev(row)))
    np.testing.assert_array_equal(got, table[ids] + row[None])


@pytest.mark.parametrize("N,D,with_o", [(5, 8, False), (7, 36, True), (130, 512, True), (3, 2048, True)])
def test_residual_layernorm(ops, N, D, with_o):
    rng = np.random.default_rng(N + D)
    x = rng.standard_normal((N, D)).astype(np.float32) * 2 + 0.5
    o = rng.standard_normal((N, D)).astype(np.float32) if with_o else None
    g = (1 + 0.1 * rng.standard_normal(D)).astype(np.float32)
    b = (0.1 * rng.standard_normal(D)).astype(np.float32)
    xd = to_dev(x.copy())
    y = ops.residual_layernorm(xd, to_dev(o) if with_o else None, to_dev(g), to_dev(b))
    xs = (x + o) if with_o else x
    np.testing.assert_array_equal(np_of(xd), xs)
    ref = co.layer_norm(xs.astype(np.float64), g.astype(np.float64), b.astype(np.float64))
    assert_close(np_of(y), ref, what="layernorm")


@pytest.mark.parametrize("N,heads,kd,pos,mask_mode", [
    (3, 4, 4, 0, 0), (3, 4, 4, 3, 1), (5, 6, 8, 4, 0), (5, 6, 8, 4, 1), (4, 2, 64, 5, 1),
    (6, 512, 8, 7, 0), (6, 8, 64, 7, 1), (2, 3, 16, 2, 1)])
def test_mha_self_step(ops, N, heads, kd, pos, mask_mode):
    rng = np.random.default_rng(N * heads + kd + pos)
    HD = heads * kd
    Tcap = pos + 2
    q = rng.standard_normal((N, HD)).astype(np.float32)
    kv = rng.standard_normal((Tcap, N, 2 * HD)).astype(np.float32)
    got = np_of(ops.mha_self_step(to_dev(q), to_dev(kv), heads, kd, pos, mask_mode))
    qs = (q * np.float32(1.0 / np.sqrt(kd))).reshape(N, heads, kd)
    k = kv[:pos + 1, :, :HD].reshape(pos + 1, N, heads, kd)
    v = kv[:pos + 1, :, HD:].reshape(pos + 1, N, heads, kd)
    s = np.einsum("jnhd,nhd->nhj", k, qs).astype(np.float32)
    if mask_mode == 0:
        s = s + np.float32(-1e9)            # every visible position, in fp32 (attention.py:471-473)
    w = co.softmax(s.astype(np.float64), axis=-1)
    ref = np.einsum("nhj,jnhd->nhd", w, v).reshape(N, HD)
    assert_close(got, ref, what="self attention")
    if mask_mode == 0:
        # the absorbed scores make the weights uniform: a causal mean of the values
        np.testing.assert_allclose(got, v.mean(axis=0).reshape(N, HD), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("G,R,L,heads,kd,mode", [
    (3, 1, 5, 4, 4, 0), (3, 3, 5, 4, 8, 0), (3, 3, 5, 4, 8, 1), (2, 5, 7, 2, 64, 0),
    (4, 3, 64, 512, 8, 0), (4, 4, 9, 3, 16, 1),
    # few wide heads: the positions of a group are split over the waves of its workgroup
    (5, 3, 64, 8, 64, 0), (3, 3, 9, 8, 64, 1), (3, 4, 33, 16, 64, 0), (2, 2, 8, 8, 64, 0),
    (2, 1, 7, 8, 64, 1)])
def test_mha_cross_step(ops, G, R, L, heads, kd, mode):
    rng = np.random.default_rng(G * R + L + kd)
    HD, N = heads * kd, G * R
    q = rng.standard_normal((N, HD)).astype(np.float32)
    kv = rng.standard_normal((G, L, 2 * HD)).astype(np.float32)
    sg, sk = (R, 1) if mode == 1 else (1, G)
    got = np_of(ops.mha_cross_step(to_dev(q), to_dev(kv), G, R, sg, sk, heads, kd))
    ref = np.zeros((N, HD))
    for g in range(G):
        k = kv[g, :, :HD].reshape(L, heads, kd).astype(np.float64)
        v = kv[g, :, HD:].reshape(L, heads, kd).astype(np.float64)
        for r in range(R):
            n = g * sg + r * sk
            qs = (q[n].astype(np.float64) / np.sqrt(kd)).reshape(heads, kd)
            w = co.softmax(np.einsum("lhd,hd->hl", k, qs), axis=-1)
            ref[n] = np.einsum("hl,lhd->hd", w, v).reshape(HD)
    assert_close(got, ref, what="cross attention")


def _compare(dims, seed, B, L, T, W, tile_mode, mask_mode="as_written", kv_from_normed=False,
             rtol=2e-5, perturb=True):
    from image_caption_amd.tengine import TransformerConfig, TransformerEngine
    p = co.make_transformer_params(seed, **dims)
    if perturb:
        p = co.perturb_transformer_params(p, seed + 1)
    img = co.make_images(seed + 50, B, L, dims["F"])
    tok_o, trace_o = co.predict_transformer_caption_sequence(
        p, img, W, T, 3, 4, dims["n_layers"], dims["heads"], dims["key_dim"], 1.0, tile_mode,
        mask_mode, kv_from_normed, return_trace=True)
    cfg = TransformerConfig(B=B, W=W, L=L, F=dims["F"], D=dims["D"], hidden=dims["hidden"],
                            heads=dims["heads"], key_dim=dims["key_dim"], filter=dims["filter_size"],
                            layers=dims["n_layers"], V=dims["V"], T=T, tile_mode=tile_mode,
                            mask_mode=mask_mode, kv_from_normed=kv_from_normed)
    eng = TransformerEngine(cfg, p, device=dev())
    tok_g, _ = eng.decode(to_dev(img))
    tr = eng.trace()
    last = np_of(eng.last_output())
    eng.close()
    tok_g = np_of(tok_g)
    ids_o = np.stack([st["out"].predicted_ids for st in trace_o])
    par_o = np.stack([st["out"].beam_parent_ids for st in trace_o])
    lp_o = np.stack([st["state"].log_probs for st in trace_o])
    same = (np_of(tr["ids"]) == ids_o).all(axis=(0, 2)) & (np_of(tr["parents"]) == par_o).all(axis=(0, 2))
    nbad = int((~same).sum())
    with np.errstate(over="ignore"):
        assert_close(np.maximum(np_of(tr["log_probs"])[:, same], -1e30), np.maximum(lp_o[:, same], -1e30),
                     rtol=rtol, what="log_probs")
    if nbad == 0:
        np.testing.assert_array_equal(tok_g, tok_o)
        assert_close(last, trace_o[-1]["last"], rtol=10 * rtol, what="last output")
    return nbad


SMALL = dict(F=24, D=16, V=40, hidden=32, n_layers=2, heads=32, key_dim=4, filter_size=64)


@pytest.mark.parametrize("tile_mode", ["reference", "repeat"])
@pytest.mark.parametrize("W", [1, 3])
def test_tdecode_small_as_written(tile_mode, W):
    assert _compare(SMALL, seed=1, B=4, L=5, T=6, W=W, tile_mode=tile_mode) == 0


@pytest.mark.parametrize("B,L,T,W", [(1, 1, 1, 1), (1, 2, 3, 3), (2, 1, 2, 2), (3, 4, 1, 3), (2, 3, 12, 3)])
def test_tdecode_edge_sizes(B, L, T, W):
    """One image / one location / one step / greedy, and a prefix longer than the usual caption."""
    assert _compare(SMALL, seed=B + L + T, B=B, L=L, T=T, W=W, tile_mode="repeat") == 0


def test_tdecode_small_conventional_variant():
    dims = dict(SMALL, heads=4, key_dim=8)
    assert _compare(dims, seed=2, B=3, L=6, T=7, W=3, tile_mode="repeat", mask_mode="causal",
                    kv_from_normed=True) == 0
    assert _compare(dims, seed=3, B=3, L=6, T=5, W=2, tile_mode="reference", mask_mode="causal",
                    kv_from_normed=False) == 0


def test_tdecode_wide_beam_unfused_vocab_path():
    dims = dict(SMALL, V=41)                      # V % 4 != 0: logits buffer + beam kernels
    assert _compare(dims, seed=4, B=3, L=4, T=5, W=6, tile_mode="repeat") == 0


def test_tdecode_medium_as_written():
    """hidden 128 (heads 128 x key_dim 8, as written), 3 layers, V 2000."""
    dims = dict(F=64, D=128, V=2000, hidden=128, n_layers=3, heads=128, key_dim=8, filter_size=256)
    assert _compare(dims, seed=5, B=6, L=16, T=8, W=3, tile_mode="reference", rtol=3e-5) <= 1


def test_tdecode_config4_dims_short():
    """The named configuration's layer shapes (6 layers, d = 512, 512 heads x 8, filter 2048,
    V = 10000) on a few images and steps."""
    dims = dict(F=64, D=512, V=10000, hidden=512, n_layers=6, heads=512, key_dim=8, filter_size=2048)
    assert _compare(dims, seed=6, B=4, L=16, T=4, W=3, tile_mode="reference", rtol=5e-5,
                    perturb=False) <= 1


def test_tengine_rejects_bad_config():
    from image_caption_amd._lib import IcapError
    from image_caption_amd.tengine import TransformerConfig, TransformerEngine
    p = co.make_transformer_params(0, **SMALL)
    cfg = TransformerConfig(B=2, W=2, L=4, F=24, D=16, hidden=32, heads=16, key_dim=6, filter=64,
                            layers=2, V=40, T=3)
    with pytest.raises((IcapError, ValueError)):
        TransformerEngine(cfg, p, device=dev())
    with pytest.raises(KeyError):
        q = dict(p)
        del q["l1_ffn_output_bias"]
        TransformerEngine(TransformerConfig(B=2, W=2, L=4, F=24, D=16, hidden=32, heads=32, key_dim=4,
                                            filter=64, layers=2, V=40, T=3), q, device=dev())


def _tmodules(p, dims):
    from image_caption_amd.decoder import TransformDecoder
    params = {"hidden_size": dims["hidden"], "num_heads": dims["key_dim"],
              "num_hidden_layers": dims["n_layers"], "filter_size": dims["filter_size"],
              "attention_dropout": 0.0, "relu_dropout": 0.0, "layer_postprocess_dropout": 0.0}
    dec = TransformDecoder(params, dims["V"], device=dev())
    dec.load_params(p, dev())
    return dec


@pytest.mark.parametrize("heads,kd", [(5, 8), (8, 64)])
def test_multi_head_attention_layer_with_and_without_mask(ops, heads, kd):
    from image_caption_amd.attention.attention import MultiHeadAttention, get_decoder_self_attention_bias
    rng = np.random.default_rng(3)
    B, T, S, dim = 3, 4, 6, 16
    q_in = rng.standard_normal((B, T, dim)).astype(np.float32)
    v_in = rng.standard_normal((B, S, 12)).astype(np.float32)
    p = {}
    for n, (i, o) in (("q", (dim, heads * kd)), ("k", (12, heads * kd)), ("v", (12, heads * kd)),
                      ("o", (heads * kd, dim))):
        p["x_" + n + "_kernel"] = (rng.standard_normal((i, o)) / np.sqrt(i)).astype(np.float32)
        p["x_" + n + "_bias"] = (0.1 * rng.standard_normal(o)).astype(np.float32)
    layer = MultiHeadAttention(heads, kd, device=dev())
    layer.set_weights(*[(p["x_%s_kernel" % n], p["x_%s_bias" % n]) for n in "qkvo"], device=dev())
    got = np_of(layer(to_dev(q_in), to_dev(v_in)))
    ref = co.multi_head_attention(q_in, v_in, p, "x_", heads, kd)
    assert_close(got, ref, what="mha")
    mask = (rng.random((B, T, S)) > 0.3).astype(np.float32)
    mask[:, :, 0] = 1
    got = np_of(layer(to_dev(q_in), to_dev(v_in), attention_mask=to_dev(mask)))
    ref = co.multi_head_attention(q_in, v_in, p, "x_", heads, kd, attention_mask=mask[:, None])
    assert_close(got, ref, what="mha masked")
    # the decoder's bias tensor handed over as a mask, self-attention shapes
    bias = np_of(get_decoder_self_attention_bias(T, device=dev()))
    np.testing.assert_array_equal(bias, co.decoder_self_attention_bias(T))
    p2 = dict(p, x_k_kernel=p["x_q_kernel"], x_v_kernel=p["x_q_kernel"])
    layer2 = MultiHeadAttention(heads, kd, device=dev())
    layer2.set_weights(*[(p2["x_%s_kernel" % n], p2["x_%s_bias" % n]) for n in "qkvo"], device=dev())
    got = np_of(layer2(to_dev(q_in), to_dev(q_in), attention_mask=to_dev(bias)))
    ref = co.multi_head_attention(q_in, q_in, p2, "x_", heads, kd, attention_mask=bias)
    assert_close(got, ref, what="mha bias-as-mask")


def _general_params(rng, d_q, d_k, d_v, heads, kd, vd, out):
    p = {}
    for n, (i, o) in (("q", (d_q, heads * kd)), ("k", (d_k, heads * kd)), ("v", (d_v, heads * vd)),
                      ("o", (heads * vd, out))):
        p["g_" + n + "_kernel"] = (rng.standard_normal((i, o)) / np.sqrt(i)).astype(np.float32)
        p["g_" + n + "_bias"] = (0.1 * rng.standard_normal(o)).astype(np.float32)
    return p


@pytest.mark.parametrize("case", ["scores", "key_tensor", "value_dim", "output_shape", "no_bias",
                                  "two_axes", "one_of_two_axes", "head_mask", "docstring_1d"])
def test_multi_head_attention_general_forms(case):
    """Every option of the reference's constructor / call signature the decoder stack does not
    use (attention/attention.py:259-270,526-562), against the oracle's general restatement:
    output and, where asked for, the attention probabilities."""
    from image_caption_amd.attention.attention import MultiHeadAttention
    rng = np.random.default_rng(11)
    heads, kd = 3, 8
    B, T, S, d = 3, 5, 7, 16
    kw, okw = {}, {}
    q_in = rng.standard_normal((B, T, d)).astype(np.float32)
    v_in = rng.standard_normal((B, S, d)).astype(np.float32)
    k_in, mask, vd, out_dim, d_k = None, None, kd, d, d
    if case == "key_tensor":
        d_k = 20
        k_in = rng.standard_normal((B, S, d_k)).astype(np.float32)
        mask = (rng.random((B, T, S)) > 0.3).astype(np.float32)
        mask[..., 0] = 1
    elif case == "value_dim":
        vd = 12
        kw["value_dim"] = okw["value_dim"] = vd
    elif case == "output_shape":
        out_dim = 10
        kw["output_shape"] = okw["output_shape"] = (2, 5)
    elif case == "no_bias":
        kw["use_bias"] = okw["use_bias"] = False
    elif case == "two_axes":
        q_in = rng.standard_normal((2, 5, 3, 4, d)).astype(np.float32)
        v_in = q_in
        kw["attention_axes"] = okw["attention_axes"] = (2, 3)
    elif case == "one_of_two_axes":
        q_in = rng.standard_normal((2, 3, 4, d)).astype(np.float32)
        v_in = rng.standard_normal((2, 3, 6, d)).astype(np.float32)
        kw["attention_axes"] = okw["attention_axes"] = 2
        mask = (rng.random((2, 3, 4, 6)) > 0.3).astype(np.float32)
        mask[..., 0] = 1
    elif case == "head_mask":
        mask = (rng.random((B, heads, T, S)) > 0.3).astype(np.float32)
        mask[..., 0] = 1
    elif case == "docstring_1d":
        heads, kd, vd = 2, 2, 2
        q_in = rng.standard_normal((B, 8, 16)).astype(np.float32)
        v_in = rng.standard_normal((B, 4, 16)).astype(np.float32)
    p = _general_params(rng, d, d_k, d, heads, kd, vd, out_dim)
    layer = MultiHeadAttention(heads, kd, device=dev(), **kw)
    layer.set_weights(*[(p["g_%s_kernel" % n], p["g_%s_bias" % n]) for n in "qkvo"], device=dev())
    got, probs = layer(to_dev(q_in), to_dev(v_in), key=None if k_in is None else to_dev(k_in),
                       attention_mask=None if mask is None else to_dev(mask),
                       return_attention_scores=True)
    ref, ref_probs = co.multi_head_attention_general(q_in, v_in, p, "g_", heads, kd, key=k_in,
                                                     attention_mask=mask, **okw)
    assert tuple(got.shape) == ref.shape and tuple(probs.shape) == ref_probs.shape
    assert_close(np_of(probs), ref_probs, what=case + " probabilities")
    assert_close(np_of(got), ref, what=case + " output")
    if case == "docstring_1d":        # the shapes of the reference's own example (`:204-210`)
        assert tuple(got.shape) == (B, 8, 16) and tuple(probs.shape) == (B, 2, 8, 4)
    if case == "two_axes":            # `:211-215`
        assert tuple(got.shape) == (2, 5, 3, 4, 16)
    if case in ("scores", "head_mask"):
        # without the scores the same call takes the step kernel (or, with a head axis in the
        # mask, the general one): same output
        again = layer(to_dev(q_in), to_dev(v_in), attention_mask=None if mask is None else to_dev(mask))
        assert_close(np_of(again), ref, what=case + " output without scores")


def test_transform_decoder_module_steps_and_engine_agree():
    """The module API (full-prefix recompute per call, as the reference) against the oracle
    step, and the module-level decode loop against the fused engine."""
    from image_caption_amd.beam_search import beam_search as bs
    from image_caption_amd.tengine import TransformerConfig, TransformerEngine
    dims = SMALL
    p = co.perturb_transformer_params(co.make_transformer_params(7, **dims), 8)
    dec = _tmodules(p, dims)
    B, W, L, T = 3, 2, 5, 5
    img = co.make_images(9, B, L, dims["F"])
    feats = co.tile_features(co.mlp_encoder(img, p["enc_kernel"], p["enc_bias"]), W, "reference")
    N = B * W
    hidden_o = [np.zeros((N, 1), np.int32)]
    hidden_g = dec.reset_state(torch.zeros(N, 1, device=dev()), None)
    assert hidden_g[0].dtype == torch.int32 and tuple(hidden_g[0].shape) == (N, 1)
    with pytest.raises(KeyError):
        dec((torch.full((N, 1), 3, dtype=torch.int32, device=dev()), {"features": to_dev(feats)}, hidden_g))
    st = bs.create_initial_beam_state_batch(B, W, device=dev())
    cfg = bs.BeamSearchConfig(W, dims["V"], 4, 1.0, bs.choose_top_k)
    dec_input = torch.full((N, 1), 3, dtype=torch.int32, device=dev())
    ids_np = np.full((N, 1), 3, np.int32)
    seq = []
    feats_g = to_dev(feats)
    for i in range(T):
        lo, hidden_o, _, last_o = co.transform_decoder_step(p, ids_np, feats, hidden_o, dims["n_layers"],
                                                            dims["heads"], dims["key_dim"])
        lg, hidden_g, none, last_g = dec((dec_input, {"features": feats_g}, hidden_g), training=False)
        assert none is None
        np.testing.assert_array_equal(np_of(hidden_g[0]), hidden_o[0])
        assert_close(np_of(last_g), last_o, rtol=5e-5, what="outputs step %d" % i)
        assert_close(np_of(lg), lo, rtol=5e-5, what="logits step %d" % i)
        out, st = bs.beam_search_step_batch(i, lg.reshape(-1, W, dims["V"]).contiguous(), st, cfg)
        seq.append((out, st))
        dec_input = out.predicted_ids.reshape(-1, 1)
        ids_np = np_of(dec_input)
    path = torch.cat(bs.loc_optimal_beam_path_batch(seq), dim=1).to(torch.int32)
    cfg_e = TransformerConfig(B=B, W=W, L=L, F=dims["F"], D=dims["D"], hidden=dims["hidden"],
                              heads=dims["heads"], key_dim=dims["key_dim"], filter=dims["filter_size"],
                              layers=dims["n_layers"], V=dims["V"], T=T)
    eng = TransformerEngine(cfg_e, p, device=dev())
    tok, _ = eng.decode(to_dev(img))
    np.testing.assert_array_equal(np_of(tok), np_of(path))
    # exported module weights drive the engine as well
    from image_caption_amd.encoder import MlpEncoder
    enc = MlpEncoder(dims["D"], name="mlp_feature_encoder", device=dev())
    enc.fc.set_weights(p["enc_kernel"], p["enc_bias"], dev())
    eng2 = TransformerEngine(cfg_e, dec.export_params(enc), device=dev())
    tok2, _ = eng2.decode(to_dev(img))
    np.testing.assert_array_equal(np_of(tok2), np_of(tok))
