"""Every benchmark configuration on the GPU: the greedy small-batch case and the large-vocab
wide-beam case against the oracle (sizes the oracle finishes in seconds), the bf16 and the
transformer configurations at their full per-GPU size through size-independent properties
(determinism, permutation equivariance in the self-consistent tile order, valid captions),
and the equivalence of sharded and whole-batch decoding that the multi-GPU layout rests on."""
import numpy as np
import pytest
import torch

from oracle import caption_oracle as co
from tests.helpers import check_decode_following, dev, np_of, to_dev

pytestmark = pytest.mark.gpu

PROD = dict(F=2048, D=256, E=256, H=512, V=10000)


def _decode_vs_oracle(dims, B, W, L, T, tile_mode, fused, peaked, seed=0, precision="fp32"):
    from image_caption_amd.engine import CaptionConfig, CaptionEngine
    p = (co.make_peaked_params if peaked else co.make_params)(seed, **dims)
    img = co.make_images(seed + 50, B, L, dims["F"])
    cfg = CaptionConfig(B=B, W=W, L=L, F=dims["F"], D=dims["D"], E=dims["E"], H=dims["H"],
                        U=dims["H"], V=dims["V"], T=T, start_id=3, eos_id=4, tile_mode=tile_mode,
                        precision=precision)
    eng = CaptionEngine(cfg, p, device=dev())
    if fused is not None:
        eng.set_fused_vocab(fused)
    tok, _ = eng.decode(to_dev(img))
    tr = {k: np_of(v) for k, v in eng.trace().items()}
    tok = np_of(tok)
    was_fused = eng.fused_vocab
    eng.close()
    nbad, ties = check_decode_following(p, img, W, T, 3, 4, 1.0, tile_mode, tok, tr)
    return nbad, ties, was_fused, tok


@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("peaked", [True, False])
def test_config1_greedy_b32_against_oracle(fused, peaked):
    """BASELINE configs[0]: greedy (beam 1), 32 images, production dims, 20 steps - the
    small-batch kernels (skinny LSTM / dense forms, unfused beam step by default) and the fused
    projection, both against the oracle."""
    nbad, ties, was_fused, tok = _decode_vs_oracle(PROD, B=32, W=1, L=64, T=20, tile_mode="reference",
                                                   fused=fused, peaked=peaked)
    assert was_fused == fused
    assert nbad <= (0 if peaked else 2), ties
    if peaked:
        assert len({tuple(r) for r in tok.tolist()}) >= 16      # captions differ per image


SMALL = dict(F=48, D=128, E=32, H=128, V=200)
MID = dict(F=64, D=256, E=64, H=256, V=1000)


@pytest.mark.parametrize("dims,B,L,T,peaked", [(SMALL, 5, 6, 6, True), (SMALL, 40, 6, 6, False),
                                               (MID, 33, 9, 8, True), (PROD, 32, 64, 20, True),
                                               (PROD, 64, 64, 20, False)])
def test_greedy_decode_in_one_launch(monkeypatch, dims, B, L, T, peaked):
    """ICAP_SMALL_LOOP=1: the whole greedy step loop as one launch, against the oracle step by
    step and against the launch-per-kernel loop (the two accumulate their products in different
    orders: tokens equal except where Keras-default weights tie, scores within 1e-4)."""
    from image_caption_amd.engine import CaptionConfig, CaptionEngine
    p = (co.make_peaked_params if peaked else co.make_params)(B, **dims)
    p = co.perturb_biases(p, B + 1)
    img = co.make_images(B + 2, B, L, dims["F"])
    cfg = CaptionConfig(B=B, W=1, L=L, F=dims["F"], D=dims["D"], E=dims["E"], H=dims["H"],
                        U=dims["H"], V=dims["V"], T=T, start_id=3, eos_id=4)
    monkeypatch.setenv("ICAP_SMALL_LOOP", "1")
    eng = CaptionEngine(cfg, p, device=dev())
    assert eng.small_loop
    tok, sc = eng.decode(to_dev(img), check=True)
    tr = {k: np_of(v) for k, v in eng.trace().items()}
    tok, sc = np_of(tok), np_of(sc)
    # a second decode on the same engine (the barrier words start from zero again)
    tok2, _ = eng.decode(to_dev(img), check=True)
    np.testing.assert_array_equal(np_of(tok2), tok)
    eng.close()
    nbad, ties = check_decode_following(p, img, 1, T, 3, 4, 1.0, "reference", tok, tr)
    assert nbad <= (0 if peaked else 2), ties
    monkeypatch.delenv("ICAP_SMALL_LOOP")
    ref = CaptionEngine(cfg, p, device=dev())
    assert not ref.small_loop
    rtok, rsc = ref.decode(to_dev(img))
    rtok, rsc = np_of(rtok), np_of(rsc)
    ref.close()
    same = (rtok == tok).all(axis=-1).reshape(-1)
    assert same.sum() >= B - (0 if peaked else 2)
    np.testing.assert_allclose(sc.reshape(-1)[same], rsc.reshape(-1)[same], rtol=1e-4, atol=1e-4)


def test_config1_default_path_is_the_small_batch_one():
    from image_caption_amd.engine import CaptionConfig, CaptionEngine
    eng = CaptionEngine(CaptionConfig(B=32, W=1, L=64, T=2, **{k: PROD[k] for k in "FDEHV"},
                                      U=PROD["H"]), co.make_params(0, **PROD), device=dev())
    assert not eng.fused_vocab
    assert not eng.small_loop           # the one-launch loop is opt-in
    eng.close()


BIG_VOCAB = dict(F=512, D=256, E=256, H=512, V=50000)


@pytest.mark.parametrize("peaked", [True, False])
def test_config5_wide_beam_large_vocab_against_oracle(peaked):
    """BASELINE configs[4] at a small batch: beam 10, 196 locations, 512-dim features,
    50 000 words, 20 steps, end to end against the oracle."""
    nbad, ties, _, tok = _decode_vs_oracle(BIG_VOCAB, B=4, W=10, L=196, T=20, tile_mode="reference",
                                           fused=None, peaked=peaked)
    # (Keras-default weights over 50 000 words: log-probs within 1e-3 of each other, exact fp32
    # ties between candidates are the rule; every adopted pick is a proven tie)
    assert nbad <= (0 if peaked else 4), ties
    assert tok.min() >= 0 and tok.max() < BIG_VOCAB["V"]


def test_config5_repeat_order_and_beam16():
    nbad, ties, _, _ = _decode_vs_oracle(BIG_VOCAB, B=3, W=16, L=196, T=8, tile_mode="repeat",
                                         fused=None, peaked=True, seed=1)
    assert nbad == 0, ties


def _valid_captions(tok, V, eos=4):
    assert tok.min() >= 0 and tok.max() < V
    for row in tok:
        ends = np.nonzero(row == eos)[0]
        if len(ends):
            assert (row[ends[0]:] == eos).all()


def test_config3_bf16_beam5_full_share_properties():
    """BASELINE configs[2], one GPU's share at full size (512 images, beam 5, bf16 operands):
    a repeated decode is bit-identical, a permuted batch gives the permuted result (images are
    independent in the self-consistent tile order), captions are well formed, and the returned
    score is the best final beam's."""
    from image_caption_amd.engine import CaptionConfig, CaptionEngine
    B, W, T = 512, 5, 20
    p = co.make_peaked_params(2, **PROD)
    img = co.make_images(3, B, 64, PROD["F"])
    cfg = CaptionConfig(B=B, W=W, L=64, T=T, U=PROD["H"], tile_mode="repeat", precision="bf16",
                        **{k: PROD[k] for k in "FDEHV"})
    eng = CaptionEngine(cfg, p, device=dev())
    assert eng.fused_vocab
    tok_a, sc_a = eng.decode(to_dev(img))
    tok_a, sc_a = np_of(tok_a).copy(), np_of(sc_a).copy()
    tr_a = {k: np_of(v).copy() for k, v in eng.trace().items()}
    perm = np.random.default_rng(7).permutation(B)
    tok_b, sc_b = eng.decode(to_dev(img[perm]))
    np.testing.assert_array_equal(np_of(tok_b), tok_a[perm])
    np.testing.assert_array_equal(np_of(sc_b), sc_a[perm])
    tr_b = {k: np_of(v) for k, v in eng.trace().items()}
    for k in tr_a:
        np.testing.assert_array_equal(tr_b[k], tr_a[k][:, perm], err_msg=k)
    tok_c, sc_c = eng.decode(to_dev(img))
    np.testing.assert_array_equal(np_of(tok_c), tok_a)
    np.testing.assert_array_equal(np_of(sc_c), sc_a)
    _valid_captions(tok_a, PROD["V"])
    assert not np.isnan(sc_a).any() and (sc_a <= 0).all()
    assert len({tuple(r) for r in tok_a.tolist()}) > B // 4
    # parents in range, step scores non-increasing along the ranks (top-k order)
    assert tr_a["parents"].min() >= 0 and tr_a["parents"].max() < W
    s = np.maximum(tr_a["scores"], -1e30)
    assert (s[:, :, :-1] >= s[:, :, 1:]).all()
    eng.close()


def test_config4_transformer_b1024_full_size_properties():
    """BASELINE configs[3] at full size (1024 images, beam 3, 6 layers, d = 512, 20 steps):
    determinism, permutation equivariance in the self-consistent tile order, well-formed
    captions."""
    from image_caption_amd import synthetic
    from image_caption_amd.tengine import TransformerConfig, TransformerEngine
    B, W, T = 1024, 3, 20
    kw = dict(F=2048, D=512, hidden=512, heads=512, key_dim=8, filter=2048, layers=6, V=10000)
    p = synthetic.make_transformer_params(0, kw["F"], kw["D"], kw["V"], kw["hidden"], kw["layers"],
                                          kw["heads"], kw["key_dim"], kw["filter"])
    p["t_fc2_kernel"] = p["t_fc2_kernel"] * np.float32(64)      # peaked word distributions
    p["t_embedding"] = p["t_embedding"] * np.float32(20)
    img = co.make_images(5, B, 64, kw["F"])
    eng = TransformerEngine(TransformerConfig(B=B, W=W, L=64, T=T, tile_mode="repeat", **kw), p,
                            device=dev())
    tok_a, sc_a = eng.decode(to_dev(img))
    tok_a, sc_a = np_of(tok_a).copy(), np_of(sc_a).copy()
    perm = np.random.default_rng(9).permutation(B)
    tok_b, sc_b = eng.decode(to_dev(img[perm]))
    np.testing.assert_array_equal(np_of(tok_b), tok_a[perm])
    np.testing.assert_array_equal(np_of(sc_b), sc_a[perm])
    tok_c, _ = eng.decode(to_dev(img))
    np.testing.assert_array_equal(np_of(tok_c), tok_a)
    _valid_captions(tok_a, kw["V"])
    assert not np.isnan(sc_a).any()
    eng.close()


def _run_shard(p, x, W, T, precision):
    from image_caption_amd.engine import CaptionConfig, CaptionEngine
    cfg = CaptionConfig(B=x.shape[0], W=W, L=64, T=T, U=PROD["H"], tile_mode="repeat",
                        precision=precision, **{k: PROD[k] for k in "FDEHV"})
    eng = CaptionEngine(cfg, p, device=dev())
    eng.set_fused_vocab(True)
    tok, sc = eng.decode(to_dev(x))
    out = (np_of(tok).copy(), np_of(sc).copy(), {k: np_of(v).copy() for k, v in eng.trace().items()})
    eng.close()
    return out


@pytest.mark.parametrize("precision,W", [("fp32", 3), ("bf16", 5)])
def test_sharded_decode_equals_whole_batch_bit_for_bit(precision, W):
    """The multi-GPU layout (contiguous image shards, replicated weights, no per-step
    communication) on one GPU: two engines on the two 512-image halves of a 1024-image batch,
    in the self-consistent tile order, give bit for bit what one engine gives on the whole
    batch - tokens, scores and every per-step record.  (Each image's arithmetic is the same
    whatever the batch around it as long as the batch sizes select the same kernel forms,
    which 512 and 1024 images do.)"""
    from image_caption_amd.distributed import shard_bounds
    B, T = 1024, 10
    p = co.make_peaked_params(1, **PROD)
    img = co.make_images(8, B, 64, PROD["F"])
    tok, sc, tr = _run_shard(p, img, W, T, precision)
    parts = [_run_shard(p, img[slice(*shard_bounds(B, 2, r))], W, T, precision) for r in range(2)]
    np.testing.assert_array_equal(np.concatenate([q[0] for q in parts]), tok)
    np.testing.assert_array_equal(np.concatenate([q[1] for q in parts]), sc)
    for k in tr:
        np.testing.assert_array_equal(np.concatenate([q[2][k] for q in parts], axis=1), tr[k], err_msg=k)


def test_ragged_shards_equal_whole_batch():
    """Ragged shards (67 images over 3 ranks: 23 + 22 + 22).  Small shards may select other
    kernel forms than the whole batch (other summation order), so floats are held to 1e-5 and
    the tokens, on weights with clear decisions, to equality."""
    from image_caption_amd.distributed import shard_bounds
    B, W, T = 67, 3, 12
    p = co.make_peaked_params(1, **PROD)
    img = co.make_images(9, B, 64, PROD["F"])
    tok, sc, _ = _run_shard(p, img, W, T, "fp32")
    parts = [_run_shard(p, img[slice(*shard_bounds(B, 3, r))], W, T, "fp32") for r in range(3)]
    assert [q[0].shape[0] for q in parts] == [23, 22, 22]
    np.testing.assert_array_equal(np.concatenate([q[0] for q in parts]), tok)
    np.testing.assert_allclose(np.concatenate([q[1] for q in parts]), sc, rtol=1e-5)
    torch.cuda.synchronize()
