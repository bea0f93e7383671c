"""The fused vocab projection + beam step for beams wider than the records are deep (beam
widths 6..16 on records of 5 logits per 64 columns): the records kernel proves per image
that no unrecorded logit can be a successor, or raises its overflow word, and the engine
repeats such a step on full logits.  Checked: the kernel against the unfused beam step on
the same logits (integers exact), the overflow word on adversarial logits, and the engine
- fused, with every step forced through the repetition, and on weights that overflow by
construction - against the unfused engine and the oracle."""
import numpy as np
import pytest
import torch

from oracle import caption_oracle as co
from tests.helpers import check_decode_following, dev, np_of, to_dev

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from image_caption_amd import ops as o
    return o


def _state(rng, B, W, time):
    if time == 0:
        return (np.zeros((B, W), np.float32), np.zeros((B, W), np.uint8), np.zeros((B, W), np.int32))
    lp = -np.abs(rng.standard_normal((B, W)) * 3).astype(np.float32)
    fin = (rng.random((B, W)) < 0.3).astype(np.uint8)
    ln = rng.integers(1, 9, size=(B, W)).astype(np.int32)
    return lp, fin, ln


@pytest.mark.parametrize("B,W,K,V,time,eos", [
    (5, 6, 64, 1000, 3, 4), (3, 10, 128, 50000, 2, 4), (4, 16, 64, 5000, 1, 0),
    (6, 10, 64, 2000, 0, 4), (2, 8, 32, 100, 5, 4), (3, 7, 64, 1004, 4, 1003),
    (2, 16, 32, 64, 2, 4)])
def test_wide_records_step_equals_unfused_step(ops, B, W, K, V, time, eos):
    rng = np.random.default_rng(B * 7 + W + V)
    x = rng.standard_normal((B * W, K)).astype(np.float32)
    w = (rng.standard_normal((K, V)) * 0.4).astype(np.float32)
    bias = rng.standard_normal(V).astype(np.float32)
    lp, fin, ln = _state(rng, B, W, time)
    d = to_dev
    logits = ops.dense(d(x), d(w), d(bias)).reshape(B, W, V)
    ref = ops.beam_step(logits, d(lp), d(fin), d(ln), time, eos, 1.0)
    rec, eos_x = ops.dense_topk(d(x), d(w), d(bias), eos, 5)
    got = ops.beam_step_from_records_wide(rec, eos_x, d(lp), d(fin), d(ln), time, eos, 1.0, V)
    overflow = int(got[6].item())
    if overflow:
        # allowed (the proof is conservative) but only where few candidates exist
        assert V <= 128, "unexpected overflow"
        return
    names = ("scores", "ids", "parents", "next_log_probs", "next_finished", "next_lengths")
    for n, a, b_ in zip(names, got[:6], ref):
        a, b_ = np_of(a), np_of(b_)
        if a.dtype.kind == "f":
            with np.errstate(over="ignore"):
                np.testing.assert_allclose(np.maximum(a, -1e30), np.maximum(b_, -1e30), rtol=3e-6, err_msg=n)
        else:
            np.testing.assert_array_equal(a, b_, err_msg=n)


def test_wide_records_step_raises_overflow_when_a_half_tile_hides_candidates(ops):
    """All large logits inside one 64-column half tile: its record holds 5 of them, the beam
    wants 10 - the kernel must say so; spread over different half tiles it must not."""
    B, W, K, V = 3, 10, 64, 4096
    rng = np.random.default_rng(3)
    x = rng.standard_normal((B * W, K)).astype(np.float32)
    w = (rng.standard_normal((K, V)) * 0.05).astype(np.float32)
    lp, fin, ln = _state(rng, B, W, 2)
    fin[:] = 0
    d = to_dev
    for cols, expect in ((np.arange(16) + 640, 1), (np.arange(16) * 64 + 7, 0)):
        bias = np.zeros(V, np.float32)
        bias[cols] = 30.0 + rng.random(16).astype(np.float32)
        rec, eos_x = ops.dense_topk(d(x), d(w), d(bias), 4, 5)
        got = ops.beam_step_from_records_wide(rec, eos_x, d(lp), d(fin), d(ln), 2, 4, 1.0, V)
        assert int(got[6].item()) == expect
        if not expect:
            ref = ops.beam_step(ops.dense(d(x), d(w), d(bias)).reshape(B, W, V), d(lp), d(fin), d(ln),
                                2, 4, 1.0)
            np.testing.assert_array_equal(np_of(got[1]), np_of(ref[1]))
            np.testing.assert_array_equal(np_of(got[2]), np_of(ref[2]))


def test_wide_records_step_rejects_bad_arguments(ops):
    from image_caption_amd._lib import IcapError
    rec = torch.zeros(4, 2, 8, device=dev())                  # records of 3, not 5
    z = torch.zeros(2, 2, device=dev())
    with pytest.raises(ValueError):
        ops.beam_step_from_records_wide(rec, torch.zeros(4, device=dev()), z, z.to(torch.uint8),
                                        z.to(torch.int32), 1, 4, 1.0, 100)
    rec = torch.zeros(34, 2, 12, device=dev())
    z = torch.zeros(2, 17, device=dev())
    with pytest.raises(IcapError):
        ops.beam_step_from_records_wide(rec, torch.zeros(34, device=dev()), z, z.to(torch.uint8),
                                        z.to(torch.int32), 1, 4, 1.0, 100)


DIMS = dict(F=64, D=32, E=32, H=64, V=3000)


def _engine_records(p, img, B, W, L, T, tile_mode, fused):
    from image_caption_amd.engine import CaptionConfig, CaptionEngine
    cfg = CaptionConfig(B=B, W=W, L=L, T=T, U=DIMS["H"], tile_mode=tile_mode,
                        **{k: DIMS[k] for k in "FDEHV"})
    eng = CaptionEngine(cfg, p, device=dev())
    eng.set_fused_vocab(fused)
    assert eng.fused_vocab == fused
    tok, sc = eng.decode(to_dev(img))
    out = dict(tok=np_of(tok).copy(), sc=np_of(sc).copy(),
               **{k: np_of(v).copy() for k, v in eng.trace().items()})
    eng.close()
    return out


@pytest.mark.parametrize("W", [6, 10, 16])
@pytest.mark.parametrize("tile_mode", ["reference", "repeat"])
def test_engine_wide_fused_equals_unfused_and_oracle(monkeypatch, W, tile_mode):
    B, L, T = 5, 6, 9
    p = co.perturb_biases(co.make_peaked_params(W, **DIMS), 3)
    p["fc2_bias"][4] = 3.0
    img = co.make_images(W + 1, B, L, DIMS["F"])
    a = _engine_records(p, img, B, W, L, T, tile_mode, True)
    b = _engine_records(p, img, B, W, L, T, tile_mode, False)
    monkeypatch.setenv("ICAP_FORCE_WIDE_RESCUE", "1")     # every step repeated on full logits
    c = _engine_records(p, img, B, W, L, T, tile_mode, True)
    monkeypatch.delenv("ICAP_FORCE_WIDE_RESCUE")
    for other, what in ((b, "unfused"), (c, "forced repetition")):
        for k in ("tok", "ids", "parents"):
            np.testing.assert_array_equal(a[k], other[k], err_msg="%s %s" % (what, k))
        for k in ("sc", "scores", "log_probs"):
            np.testing.assert_allclose(np.maximum(a[k], -1e30), np.maximum(other[k], -1e30), rtol=3e-6,
                                       err_msg="%s %s" % (what, k))
    # (the repetition computes its logits with the 128x128 tile kernel of the fused
    # projection, the unfused engine with whatever form its shape selects: same integers,
    # floats to rounding)
    tr = {k: a[k] for k in ("ids", "parents", "scores", "log_probs")}
    nbad, ties = check_decode_following(p, img, W, T, 3, 4, 1.0, tile_mode, a["tok"], tr)
    assert nbad == 0, ties


def test_engine_wide_on_weights_that_overflow_by_construction():
    """Sixteen words with a large bias inside one half tile: nearly every step's successors
    come from logits the records do not hold, so the engine's result rests on the repetition
    path alone.  Against the oracle."""
    B, W, L, T = 4, 10, 6, 8
    p = co.perturb_biases(co.make_peaked_params(2, **DIMS), 3)
    p["fc2_bias"][4] = 0.0
    p["fc2_bias"][1280:1296] += 25.0
    img = co.make_images(5, B, L, DIMS["F"])
    a = _engine_records(p, img, B, W, L, T, "repeat", True)
    tr = {k: a[k] for k in ("ids", "parents", "scores", "log_probs")}
    nbad, ties = check_decode_following(p, img, W, T, 3, 4, 1.0, "repeat", a["tok"], tr)
    assert nbad == 0, ties
    assert ((a["ids"] >= 1280) & (a["ids"] < 1296)).mean() > 0.5


def test_engine_wide_at_config5_dims_is_fused_by_default():
    from image_caption_amd.engine import CaptionConfig, CaptionEngine
    dims = dict(F=512, D=256, E=256, H=512, V=50000)
    eng = CaptionEngine(CaptionConfig(B=256, W=10, L=196, T=2, U=512, **dims),
                        co.make_params(0, **dims), device=dev())
    assert eng.fused_vocab
    eng.close()
