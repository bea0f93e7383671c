"""Split mode (fp32 operands split exactly into three bf16 pieces, six partial products summed
in fp32 on the bf16 matrix pipe): held to the SAME bars as the exact-f32 kernels - 1e-5
relative for operators, token identity with the near-tie rule end to end - plus a direct
measurement of how far its products are from exact ones."""
import numpy as np
import pytest
import torch

from oracle import caption_oracle as co
from tests.helpers import assert_close, np_of, to_dev
from tests.test_gpu_e2e import PROD, _compare_decode

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from image_caption_amd import ops as _ops
    return _ops


@pytest.mark.parametrize("M,K,N,act,bias", [
    (1, 32, 1, 0, True), (7, 64, 5, 1, True), (64, 128, 128, 0, False), (65, 96, 129, 1, True),
    (200, 512, 300, 0, True), (1536, 512, 512, 0, True), (300, 2048, 256, 1, True),
    (1536, 512, 10000, 0, True)])
def test_dense_split(ops, M, K, N, act, bias):
    rng = np.random.default_rng(M * 3 + K + N)
    x = rng.standard_normal((M, K)).astype(np.float32)
    w = (rng.standard_normal((K, N)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32) if bias else None
    got = np_of(ops.dense_split(to_dev(x), ops.pack_weight_split(to_dev(w)),
                                to_dev(b) if bias else None, act))
    ref = x.astype(np.float64) @ w.astype(np.float64)
    if bias:
        ref = ref + b
    if act:
        ref = np.maximum(ref, 0)
    assert_close(got, ref.astype(np.float32), rtol=1e-5, what="dense_split")
    # as close to the exact result as the exact-f32 MFMA kernel is
    exact = np_of(ops.dense(to_dev(x), to_dev(w), to_dev(b) if bias else None, act=act))
    err_split = np.abs(got - ref).max()
    err_exact = np.abs(exact - ref).max()
    assert err_split <= 2.0 * err_exact + 1e-7, (err_split, err_exact)


def test_split_is_exact_where_fp32_is(ops):
    """Small integers: every product and sum is exact in fp32, so the result must be exact."""
    rng = np.random.default_rng(0)
    x = rng.integers(-8, 9, size=(70, 64)).astype(np.float32)
    w = rng.integers(-8, 9, size=(64, 130)).astype(np.float32)
    got = np_of(ops.dense_split(to_dev(x), ops.pack_weight_split(to_dev(w))))
    np.testing.assert_array_equal(got, x @ w)
    # and values that need all 24 mantissa bits of both operands
    x2 = (rng.integers(1, 2 ** 24, size=(33, 32)).astype(np.float32)) * np.float32(2.0 ** -24)
    w2 = np.zeros((32, 128), np.float32)
    w2[np.arange(32), np.arange(32)] = 1.0          # identity block: y[:, :32] == x exactly
    got2 = np_of(ops.dense_split(to_dev(x2), ops.pack_weight_split(to_dev(w2))))
    np.testing.assert_array_equal(got2[:, :32], x2)


@pytest.mark.parametrize("N,C,H,V,with_emb", [
    (5, 32, 32, 11, True), (70, 128, 64, 50, True), (33, 64, 96, 7, False), (1536, 256, 512, 10000, True)])
def test_lstm_step_split(ops, N, C, H, V, with_emb):
    rng = np.random.default_rng(N + C + H)
    ctx = rng.standard_normal((N, C)).astype(np.float32)
    h = rng.standard_normal((N, H)).astype(np.float32) * 0.5
    c = rng.standard_normal((N, H)).astype(np.float32)
    kern = (rng.standard_normal((C, 4 * H)) / np.sqrt(C + H)).astype(np.float32)
    rec = (rng.standard_normal((H, 4 * H)) / np.sqrt(C + H)).astype(np.float32)
    bias = rng.standard_normal(4 * H).astype(np.float32) * 0.1
    emb_proj = rng.standard_normal((V, 4 * H)).astype(np.float32) * 0.3 if with_emb else None
    ids = rng.integers(0, V, size=N).astype(np.int32) if with_emb else None
    packed = ops.PackedSplitWeight(C + H, 4 * H, "cuda", gate_units=H)
    packed.fill(to_dev(kern), 0).fill(to_dev(rec), C)
    h_g, c_g = ops.lstm_step_split(to_dev(ctx), to_dev(h), to_dev(c), packed, to_dev(bias),
                                   to_dev(emb_proj) if with_emb else None,
                                   to_dev(ids) if with_emb else None)
    z = ctx.astype(np.float64) @ kern + h.astype(np.float64) @ rec + bias
    if with_emb:
        z = z + emb_proj[ids]
    i, f, g, o = np.split(z, 4, axis=1)
    sig = lambda v: 1 / (1 + np.exp(-v))
    cn = sig(f) * c + sig(i) * np.tanh(g)
    assert_close(np_of(c_g), cn.astype(np.float32), rtol=1e-5, what="c")
    assert_close(np_of(h_g), (sig(o) * np.tanh(cn)).astype(np.float32), rtol=1e-5, what="h")


def test_topk_records_split_equal_f32_records(ops):
    rng = np.random.default_rng(3)
    M, K, V = 48, 64, 5000
    x = rng.standard_normal((M, K)).astype(np.float32)
    w = (rng.standard_normal((K, V)) * 0.3).astype(np.float32)
    b = rng.standard_normal(V).astype(np.float32) * 0.1
    r32, e32 = ops.dense_topk(to_dev(x), to_dev(w), to_dev(b), 4, 3)
    rsp, esp = ops.dense_topk_split(to_dev(x), ops.pack_weight_split(to_dev(w)), to_dev(b), 4, 3)
    r32, rsp = np_of(r32), np_of(rsp)
    assert_close(rsp[:, :, :5], r32[:, :, :5], rtol=1e-5, what="record values", allow_inf=True)  # (-inf: empty slots)
    same = rsp[:, :, 5:].view(np.int32) == r32[:, :, 5:].view(np.int32)
    assert same.mean() > 0.999                      # columns agree except for fp32-level ties
    assert_close(np_of(esp), np_of(e32), rtol=1e-5, what="eos logit")


def test_engine_split_small_against_oracle():
    dims = dict(F=64, D=32, E=32, H=64, V=300)
    nbad, explained = _compare_decode({"precision": "fp32_split"}, dims, seed=1, B=6, L=7, T=8, W=3,
                                      tile_mode="reference", bias_seed=9)
    assert nbad == 0, explained


def test_engine_split_full_size_config2_against_oracle():
    """The headline configuration in split mode against the fp32 oracle, at the fp32 bars:
    the oracle follows the engine's records, every differing pick a proven near-tie at its rank
    (2e-6), log-probs and scores of all images within 1e-5.  (The exact-token run on the
    margin-screened batch is tests/test_gpu_e2e.py::test_decode_full_size_config2_exact_tokens.)"""
    nbad, explained = _compare_decode({"precision": "fp32_split"}, PROD, seed=0, B=512, L=64, T=20,
                                      W=3, tile_mode="reference")
    print("split mode, full size, flat weights: %d of 512 images took a near-tie adoption"
          % nbad)
    assert nbad <= 51, explained


def test_engine_split_rejects_unaligned_dims():
    from image_caption_amd.engine import CaptionConfig, CaptionEngine
    p = co.make_params(0, 24, 8, 8, 16, 30)
    cfg = CaptionConfig(B=2, W=2, L=4, F=24, D=8, E=8, H=16, U=16, V=30, T=3, start_id=3, eos_id=4,
                        precision="fp32_split")
    with pytest.raises(RuntimeError, match="multiples of 32"):
        CaptionEngine(cfg, {k: to_dev(v) for k, v in p.items()})
