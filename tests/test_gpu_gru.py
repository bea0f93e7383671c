"""GRU decoder (reference decoder/recurrent_decoder.py) vs the oracle: the cell step, the
decoder module and the whole-sequence decode with cell="gru".

No reference test or fixture covers this decoder (no driver in the reference constructs
it): parity unpinned, the checker is the oracle's restatement of recurrent_decoder.py and
of the Keras GRUCell contract.  Bars as for the LSTM path: 1e-5 relative on states and
log-probs; whole-sequence decodes are checked by the oracle following the engine's records
through proven near-ties (tests/helpers.check_decode_following)."""
import numpy as np
import pytest
import torch

from oracle import caption_oracle as co
from tests.helpers import assert_close, check_decode_following, dev, np_of, prefer_fused, to_dev
from tests.test_gpu_e2e import SMALL

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from image_caption_amd import ops as o
    return o


def test_gru_pack_layout(ops):
    rng = np.random.default_rng(0)
    CE, H = 7, 5
    k = rng.standard_normal((CE, 3 * H)).astype(np.float32)
    r = rng.standard_normal((H, 3 * H)).astype(np.float32)
    b = rng.standard_normal((2, 3 * H)).astype(np.float32)
    k4, r4, b4 = (np_of(t) for t in ops.gru_pack(to_dev(k), to_dev(r), to_dev(b)))
    np.testing.assert_array_equal(k4[:, :3 * H], k)
    np.testing.assert_array_equal(k4[:, 3 * H:], 0)
    np.testing.assert_array_equal(r4[:, :2 * H], r[:, :2 * H])
    np.testing.assert_array_equal(r4[:, 2 * H:3 * H], 0)
    np.testing.assert_array_equal(r4[:, 3 * H:], r[:, 2 * H:])
    np.testing.assert_array_equal(b4[:2 * H], b[0, :2 * H] + b[1, :2 * H])
    np.testing.assert_array_equal(b4[2 * H:3 * H], b[0, 2 * H:])
    np.testing.assert_array_equal(b4[3 * H:], b[1, 2 * H:])


@pytest.mark.parametrize("N,C,E,H,V,use_ids", [
    (5, 4, 4, 6, 11, True), (5, 4, 4, 6, 11, False), (7, 6, 5, 9, 13, True),
    (130, 8, 8, 36, 50, True), (96, 256, 256, 512, 1000, True), (200, 256, 256, 512, 0, False),
    (1152, 64, 64, 512, 300, True), (1536, 256, 256, 512, 10000, True)])
def test_gru_step(ops, N, C, E, H, V, use_ids):
    rng = np.random.default_rng(N + C + H + 1)
    kernel = (rng.standard_normal((C + E, 3 * H)) / np.sqrt(C + E)).astype(np.float32)
    rec = (rng.standard_normal((H, 3 * H)) / np.sqrt(H)).astype(np.float32)
    bias = rng.standard_normal((2, 3 * H)).astype(np.float32) * 0.2
    ctx = rng.standard_normal((N, C)).astype(np.float32)
    h = rng.standard_normal((N, H)).astype(np.float32) * 0.5
    k4, r4, b4 = ops.gru_pack(to_dev(kernel), to_dev(rec), to_dev(bias))
    if use_ids:
        table = rng.standard_normal((V, E)).astype(np.float32)
        ids = rng.integers(0, V, size=N).astype(np.int32)
        emb = table[ids]
        h1 = ops.gru_step(to_dev(ctx), to_dev(h), k4, r4, b4, table=to_dev(table), ids=to_dev(ids))
        emb_proj = ops.dense(to_dev(table), k4[C:].contiguous(), None)
        h2 = ops.gru_step(to_dev(ctx), to_dev(h), k4, r4, b4, ids=to_dev(ids), emb_proj=emb_proj)
    else:
        emb = rng.standard_normal((N, E)).astype(np.float32)
        h1 = ops.gru_step(to_dev(ctx), to_dev(h), k4, r4, b4, emb=to_dev(emb))
    x = np.concatenate([ctx, emb], axis=1)
    h_ref, _ = co.gru_cell(x.astype(np.float64), h.astype(np.float64), kernel.astype(np.float64),
                           rec.astype(np.float64), bias.astype(np.float64))
    assert_close(np_of(h1), h_ref, what="gru h")
    if use_ids:
        assert_close(np_of(h2), h_ref, what="gru h (emb_proj)")


def _gru_modules(p, dims, reset=1):
    from image_caption_amd.decoder import EmbeddingLayer, RecurrentDecoder
    from image_caption_amd.encoder import MlpEncoder
    d = dev()
    emb = EmbeddingLayer(dims["V"], dims["E"], device=d)
    enc = MlpEncoder(dims["D"], name="mlp_feature_encoder", device=d)
    enc.fc.set_weights(p["enc_kernel"], p["enc_bias"], d)
    dec = RecurrentDecoder(emb, dims["E"], dims["H"], dims["V"], reset_state_from_feat=reset, device=d)
    dec.load_params(p, d)
    return enc, dec


def _gru_params(seed, dims, bias_seed=None):
    p = co.add_gru_params(co.make_params(seed, **dims), seed)
    return co.perturb_biases(p, bias_seed) if bias_seed is not None else p


@pytest.mark.parametrize("dims,N,L", [(SMALL, 6, 5), (dict(F=64, D=256, E=256, H=512, V=1000), 12, 64)])
def test_gru_decoder_step_module(dims, N, L):
    from image_caption_amd.decoder import RecurrentDecoder
    p = _gru_params(3, dims, 4)
    _, dec = _gru_modules(p, dims)
    rng = np.random.default_rng(5)
    feats = np.abs(rng.standard_normal((N, L, dims["D"]))).astype(np.float32)
    ids = rng.integers(0, dims["V"], size=(N, 1)).astype(np.int32)
    hidden_o = co.reset_state_gru(p, N, feats, 1)
    hidden_g = dec.reset_state(torch.zeros(N, 1, device=dev()), {"features": to_dev(feats)})
    assert len(hidden_g) == 1
    assert_close(np_of(hidden_g[0]), hidden_o[0], what="reset_state")
    with pytest.raises(KeyError):
        dec((to_dev(ids), {"features": to_dev(feats)}, hidden_g))
    feats_g = to_dev(feats)
    for step in range(3):
        lo, hidden_o, wo, co_ = co.gru_decoder_step(p, ids, feats, hidden_o)
        lg, hidden_g, wg, cg = dec((to_dev(ids), {"features": feats_g}, hidden_g), training=False)
        assert tuple(wg.shape) == (N, L, 1) and tuple(lg.shape) == (N, dims["V"])
        assert_close(np_of(wg), wo, what="weights step %d" % step)
        assert_close(np_of(cg), co_, what="context step %d" % step)
        assert_close(np_of(hidden_g[0]), hidden_o[0], rtol=2e-5, what="h step %d" % step)
        assert_close(np_of(lg), lo, rtol=3e-5, what="logits step %d" % step)
        ids = np.argmax(lo, axis=1).astype(np.int32)[:, None]
    zeros = dec.reset_state(torch.zeros(N, 1, device=dev()), {})
    assert len(zeros) == 1 and float(zeros[0].abs().sum()) == 0.0
    assert isinstance(dec.copy(), RecurrentDecoder)


def _compare_gru_decode(dims, seed, B, L, T, W, tile_mode, bias_seed=None, lpw=1.0, **cfg_kw):
    """GRU engine decode against the oracle following it through proven near-ties: every step
    of every image checked.  Returns (images with a near-tie adoption, the adopted ties)."""
    from image_caption_amd.engine import CaptionConfig, CaptionEngine
    p = _gru_params(seed, dims, bias_seed)
    img = co.make_images(seed + 50, B, L, dims["F"])
    cfg = CaptionConfig(B=B, W=W, L=L, F=dims["F"], D=dims["D"], E=dims["E"], H=dims["H"],
                        U=dims["H"], V=dims["V"], T=T, start_id=3, eos_id=4,
                        length_penalty_weight=lpw, tile_mode=tile_mode, cell="gru", **cfg_kw)
    eng = CaptionEngine(cfg, p, device=dev())
    prefer_fused(eng)
    tok_g, _ = eng.decode(to_dev(img))
    tr_g = {k: np_of(v) for k, v in eng.trace().items()}
    tok_g = np_of(tok_g)
    eng.close()
    return check_decode_following(p, img, W, T, 3, 4, lpw, tile_mode, tok_g, tr_g, cell="gru")


@pytest.mark.parametrize("tile_mode", ["reference", "repeat"])
@pytest.mark.parametrize("W", [1, 3])
def test_gru_decode_small(tile_mode, W):
    nbad, _ = _compare_gru_decode(SMALL, seed=1, B=5, L=5, T=7, W=W, tile_mode=tile_mode, bias_seed=9)
    assert nbad == 0


@pytest.mark.parametrize("precision", ["fp32", "fp32_split"])
def test_gru_decode_medium_production_dims(precision):
    dims = dict(F=2048, D=256, E=256, H=512, V=10000)
    nbad, explained = _compare_gru_decode(dims, seed=0, B=16, L=64, T=20, W=3, tile_mode="reference",
                                          precision=precision)
    assert nbad <= 1, explained


def test_gru_sequence_beam_search_entry():
    from image_caption_amd import sequence_beam_search
    p = _gru_params(4, SMALL, 5)
    enc, dec = _gru_modules(p, SMALL)
    img = co.make_images(6, 4, 5, SMALL["F"])
    tok_o = co.predict_image_caption_sequence(p, img, 3, 6, 3, 4, 1.0, "reference", cell="gru")
    tok_g = sequence_beam_search(enc, dec, to_dev(img), 3, 6, 3, 4, tile_mode="reference")
    np.testing.assert_array_equal(np_of(tok_g), tok_o)


@pytest.mark.parametrize("N,C,H,V", [(5, 64, 64, 11), (70, 128, 64, 50), (1536, 256, 512, 10000)])
def test_gru_step_bf16(ops, N, C, H, V):
    """bf16 operands, fp32 accumulate: against the same cell on bf16-rounded operands (1e-5)
    and against the fp32 cell (2e-2 elementwise on unit-variance operands)."""
    from tests.helpers import round_bf16
    rng = np.random.default_rng(N + C + H + 2)
    E = 64
    kernel = (rng.standard_normal((C + E, 3 * H)) / np.sqrt(C + H)).astype(np.float32)
    rec = (rng.standard_normal((H, 3 * H)) / np.sqrt(C + H)).astype(np.float32)
    bias = rng.standard_normal((2, 3 * H)).astype(np.float32) * 0.1
    ctx = rng.standard_normal((N, C)).astype(np.float32)
    h = rng.standard_normal((N, H)).astype(np.float32) * 0.5
    emb_proj4 = rng.standard_normal((V, 4 * H)).astype(np.float32) * 0.3
    emb_proj4[:, 3 * H:] = 0                              # the token never feeds rec_h
    ids = rng.integers(0, V, size=N).astype(np.int32)
    k4, r4, b4 = ops.gru_pack(to_dev(kernel), to_dev(rec), to_dev(bias))
    packed = ops.PackedWeight(C + H, 4 * H, "cuda", gate_units=H)
    packed.fill(k4[:C].contiguous(), 0).fill(r4, C)
    h_g = ops.gru_step_bf16(to_dev(ctx), to_dev(h), packed, b4, to_dev(emb_proj4), to_dev(ids))

    def cell(rnd):
        mx = rnd(ctx).astype(np.float64) @ rnd(kernel[:C]).astype(np.float64) + bias[0] + \
            emb_proj4[ids][:, :3 * H]
        mh = rnd(h).astype(np.float64) @ rnd(rec).astype(np.float64) + bias[1]
        sig = lambda v: 1 / (1 + np.exp(-v))
        xz, xr, xh = np.split(mx, 3, axis=1)
        rz, rr, rh = np.split(mh, 3, axis=1)
        z, r = sig(xz + rz), sig(xr + rr)
        return (z * h + (1 - z) * np.tanh(xh + r * rh)).astype(np.float32)
    assert_close(np_of(h_g), cell(round_bf16), rtol=1e-5, what="h vs bf16 emulation")
    assert_close(np_of(h_g), cell(lambda v: v), rtol=2e-2, what="h vs fp32")


def test_gru_engine_bf16_scores_within_the_bf16_bar():
    """bf16 mode over the GRU decoder: best-caption scores within 1e-2 of the fp32 oracle's
    (token identity is not a meaningful bar on near-flat random-weight distributions)."""
    from image_caption_amd.engine import CaptionConfig, CaptionEngine
    dims = dict(F=128, D=64, E=64, H=64, V=200)
    p = _gru_params(3, dims, 4)
    B, W, L, T = 8, 3, 16, 6
    img = co.make_images(5, B, L, dims["F"])
    _, trace_o = co.predict_image_caption_sequence(p, img, W, T, 3, 4, 1.0, "reference",
                                                   return_trace=True, cell="gru")
    cfg = CaptionConfig(B=B, W=W, L=L, F=dims["F"], D=dims["D"], E=dims["E"], H=dims["H"], U=dims["H"],
                        V=dims["V"], T=T, cell="gru", precision="bf16")
    eng = CaptionEngine(cfg, p, device=dev())
    _, best = eng.decode(to_dev(img))
    tok2, best2 = eng.decode(to_dev(img))
    np.testing.assert_array_equal(np_of(best), np_of(best2))          # deterministic
    last = trace_o[-1]
    so, fo = last["out"].scores, last["state"].finished
    so = np.where(fo.any(axis=1, keepdims=True) & ~fo, so - 1e9, so)
    np.testing.assert_allclose(np_of(best), so.max(axis=1), rtol=1e-2)
    eng.close()


def test_gru_decode_full_size_headline_shape():
    """The headline shape (512 images, beam 3, 20 steps, V = 10000, fp32, reference tile order)
    over the GRU decoder (Keras-default weights: near-flat word distributions) against the oracle
    following it: every differing pick a proven near-tie at its rank (2e-6), all records,
    log-probs and scores of all images within 1e-5."""
    dims = dict(F=2048, D=256, E=256, H=512, V=10000)
    nbad, explained = _compare_gru_decode(dims, seed=0, B=512, L=64, T=20, W=3, tile_mode="reference")
    print("GRU full size: %d of 512 images took a near-tie adoption (%d picks)" % (nbad, len(explained)))
    assert nbad <= 51, explained
