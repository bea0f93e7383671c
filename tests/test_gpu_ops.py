"""Operator-level parity of the HIP kernels (through the C ABI) against the CPU oracle.
Tolerance: 1e-5 relative (the fp32 bar of the path); integer outputs bit-exact."""
import numpy as np
import pytest
import torch

from oracle import caption_oracle as co
from tests.helpers import assert_close, dev, np_of, to_dev

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from image_caption_amd import ops as _ops
    return _ops


@pytest.mark.parametrize("M,K,N,act", [
    (5, 7, 3, 0), (1, 1, 1, 1), (33, 20, 65, 1), (200, 512, 300, 0), (130, 36, 128, 1),
    (1536, 512, 2000, 0), (4096, 256, 512, 1), (96, 512, 10000, 0), (64, 2048, 256, 1),
    # few 64x64 tiles and K % 64 == 0: the split-K form (full, ragged rows, ragged columns)
    (1536, 512, 512, 0), (100, 128, 64, 1), (70, 64, 132, 0), (512, 256, 512, 1),
    # a few dozen rows, K % 128 == 0: the small-batch form (ragged rows and columns too)
    (32, 512, 512, 0), (32, 512, 10000, 0), (5, 256, 100, 1), (64, 768, 36, 0), (33, 1024, 2000, 1)])
def test_dense(ops, M, K, N, act):
    rng = np.random.default_rng(M * 7 + K * 3 + N)
    x = rng.standard_normal((M, K)).astype(np.float32)
    w = (rng.standard_normal((K, N)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    ref = x.astype(np.float64) @ w.astype(np.float64) + b
    if act:
        ref = np.maximum(ref, 0)
    got = np_of(ops.dense(to_dev(x), to_dev(w), to_dev(b), act=act))
    assert_close(got, ref, what="dense %s" % ((M, K, N),))
    got = np_of(ops.dense(to_dev(x), to_dev(w), None, act=act))
    ref = x.astype(np.float64) @ w.astype(np.float64)
    assert_close(got, np.maximum(ref, 0) if act else ref, what="dense nobias")


def test_dense_rank3_input(ops):
    rng = np.random.default_rng(3)
    x = rng.standard_normal((3, 5, 24)).astype(np.float32)
    w = rng.standard_normal((24, 8)).astype(np.float32)
    b = rng.standard_normal(8).astype(np.float32)
    got = np_of(ops.dense(to_dev(x), to_dev(w), to_dev(b), act=1))
    assert got.shape == (3, 5, 8)
    assert_close(got, co.mlp_encoder(x, w, b), what="rank3")


@pytest.mark.parametrize("G,L,D,masked", [(3, 5, 4, False), (7, 64, 256, False), (4, 6, 12, True)])
def test_pool_features(ops, G, L, D, masked):
    rng = np.random.default_rng(G + L)
    f = rng.standard_normal((G, L, D)).astype(np.float32)
    mask = None
    if masked:
        mask = (rng.random((G, L)) > 0.5).astype(np.float32)
        mask[1] = 0
    eye = np.eye(D, dtype=np.float32)
    ref = co.feature_proj_state(f, eye, np.zeros(D, np.float32), mask=mask)
    got = np_of(ops.pool_features(to_dev(f), None if mask is None else to_dev(mask)))
    assert_close(got, ref, what="pool")


@pytest.mark.parametrize("mode", [0, 1])
def test_expand_rows(ops, mode):
    rng = np.random.default_rng(1)
    src = rng.standard_normal((4, 6)).astype(np.float32)
    got = np_of(ops.expand_rows(to_dev(src), 3, scale=0.5, tile_mode=mode))
    ref = (np.tile(src, (3, 1)) if mode == 0 else np.repeat(src, 3, axis=0)) * 0.5
    np.testing.assert_array_equal(got, ref)


def test_embedding(ops):
    rng = np.random.default_rng(2)
    table = rng.standard_normal((11, 6)).astype(np.float32)
    ids = np.array([0, 10, 3, 3, 7], np.int32)
    got = np_of(ops.embedding(to_dev(table), to_dev(ids)))
    np.testing.assert_array_equal(got, table[ids])
    got = np_of(ops.embedding(to_dev(table), to_dev(np.array([-1, 11], np.int32))))
    np.testing.assert_array_equal(got, np.zeros((2, 6), np.float32))


def _attention_case(ops, G, R, L, U, D, H, mode, with_bias, with_gate, seed):
    rng = np.random.default_rng(seed)
    N = G * R
    p = {"att_W1_kernel": (rng.standard_normal((D, U)) / np.sqrt(D)).astype(np.float32),
         "att_W1_bias": rng.standard_normal(U).astype(np.float32) * 0.1,
         "att_W2_kernel": (rng.standard_normal((H, U)) / np.sqrt(H)).astype(np.float32),
         "att_W2_bias": rng.standard_normal(U).astype(np.float32) * 0.1,
         "att_V_kernel": (rng.standard_normal((U, 1)) / np.sqrt(U)).astype(np.float32),
         "att_V_bias": np.array([0.3], np.float32)}
    feats = np.abs(rng.standard_normal((G, L, D))).astype(np.float32)
    hidden = rng.standard_normal((N, H)).astype(np.float32)
    bias = rng.standard_normal((N, L)).astype(np.float32) if with_bias else None
    fc3_w = (rng.standard_normal((H, 1)) / np.sqrt(H)).astype(np.float32)
    fc3_b = np.array([-0.2], np.float32)
    # row n of group g, beam k
    if mode == "repeat":
        sg, sk = R, 1
        tiled = np.repeat(feats, R, axis=0)
    else:
        sg, sk = 1, G
        tiled = np.tile(feats, (R, 1, 1))
    ctx_ref, w_ref = co.bahdanau_attention(tiled, hidden, p, bias=bias)
    if with_gate:
        ctx_ref = ctx_ref * co.sigmoid(co.dense(hidden, fc3_w, fc3_b))
    pf = co.dense(feats, p["att_W1_kernel"], p["att_W1_bias"])
    q = co.dense(hidden, p["att_W2_kernel"], p["att_W2_bias"])
    ctx, w = ops.attention(
        to_dev(pf), to_dev(feats), to_dev(q), to_dev(p["att_V_kernel"].reshape(-1)), 0.3,
        score_bias=None if bias is None else to_dev(bias),
        gate_state=to_dev(hidden) if with_gate else None,
        gate_w=to_dev(fc3_w.reshape(-1)) if with_gate else None, gate_b=-0.2,
        rows_per_group=R, row_stride_g=sg, row_stride_k=sk)
    assert_close(np_of(w), w_ref[..., 0], what="attention weights")
    assert_close(np_of(ctx), ctx_ref, what="attention ctx")
    np.testing.assert_allclose(np_of(w).sum(1), 1.0, rtol=1e-5)


@pytest.mark.parametrize("G,R,L,U,D,H,mode,with_bias,with_gate", [
    (3, 1, 5, 8, 4, 6, "repeat", False, False),
    (3, 3, 5, 6, 6, 6, "reference", True, True),       # U, D not multiples of 4
    (4, 3, 64, 512, 256, 512, "repeat", False, True),  # production shape
    (4, 3, 64, 512, 256, 512, "reference", False, True),
    (2, 5, 64, 512, 256, 512, "repeat", True, True),
    (2, 10, 196, 512, 256, 512, "repeat", False, True),  # L > 64, R > 4
    (5, 16, 7, 260, 300, 12, "reference", True, True),   # odd sizes, D > 256
    (2, 2, 9, 1024, 8, 8, "repeat", False, False)])
def test_attention(ops, G, R, L, U, D, H, mode, with_bias, with_gate):
    _attention_case(ops, G, R, L, U, D, H, mode, with_bias, with_gate, seed=G * 100 + R)


@pytest.mark.parametrize("N,C,E,H,V,use_ids", [
    (5, 4, 4, 6, 11, True), (5, 4, 4, 6, 11, False), (7, 6, 5, 9, 13, True),
    (130, 8, 8, 36, 50, True), (96, 256, 256, 512, 1000, True), (200, 256, 256, 512, 0, False),
    # enough 96-row patches for the one-workgroup-per-CU form of the emb_proj variant:
    (1152, 64, 64, 512, 300, True), (1100, 128, 64, 512, 300, True), (1536, 256, 256, 512, 10000, True)])
def test_lstm_step(ops, N, C, E, H, V, use_ids):
    rng = np.random.default_rng(N + C + H)
    kernel = (rng.standard_normal((C + E, 4 * H)) / np.sqrt(C + E)).astype(np.float32)
    rec = (rng.standard_normal((H, 4 * H)) / np.sqrt(H)).astype(np.float32)
    bias = rng.standard_normal(4 * H).astype(np.float32) * 0.2
    ctx = rng.standard_normal((N, C)).astype(np.float32)
    h = rng.standard_normal((N, H)).astype(np.float32) * 0.5
    c = rng.standard_normal((N, H)).astype(np.float32) * 0.5
    if use_ids:
        table = rng.standard_normal((V, E)).astype(np.float32)
        ids = rng.integers(0, V, size=N).astype(np.int32)
        emb = table[ids]
        h1, c1 = ops.lstm_step(to_dev(ctx), to_dev(h), to_dev(c), to_dev(kernel), to_dev(rec),
                               to_dev(bias), table=to_dev(table), ids=to_dev(ids))
    else:
        emb = rng.standard_normal((N, E)).astype(np.float32)
        h1, c1 = ops.lstm_step(to_dev(ctx), to_dev(h), to_dev(c), to_dev(kernel), to_dev(rec),
                               to_dev(bias), emb=to_dev(emb))
    if use_ids:
        # same step with the token-only part of the gates pre-multiplied
        emb_proj = ops.dense(to_dev(table), to_dev(np.ascontiguousarray(kernel[C:])), None)
        h2, c2 = ops.lstm_step(to_dev(ctx), to_dev(h), to_dev(c), to_dev(kernel), to_dev(rec),
                               to_dev(bias), ids=to_dev(ids), emb_proj=emb_proj)
    x = np.concatenate([ctx, emb], axis=1)
    h_ref, (_, c_ref) = co.lstm_cell(x.astype(np.float64), h.astype(np.float64),
                                     c.astype(np.float64), kernel.astype(np.float64),
                                     rec.astype(np.float64), bias.astype(np.float64))
    assert_close(np_of(c1), c_ref, what="lstm c")
    assert_close(np_of(h1), h_ref, what="lstm h")
    if use_ids:
        assert_close(np_of(c2), c_ref, what="lstm c (emb_proj)")
        assert_close(np_of(h2), h_ref, what="lstm h (emb_proj)")


def test_bad_arguments_raise(ops):
    from image_caption_amd._lib import IcapError
    x = torch.zeros(4, 8, device=dev())
    w = torch.zeros(8, 4, device=dev())
    with pytest.raises(ValueError):
        ops.dense(x, torch.zeros(9, 4, device=dev()))
    with pytest.raises(ValueError):
        ops.dense(x.cpu(), w)
    with pytest.raises(TypeError):
        ops.dense(x.double(), w)
    with pytest.raises(IcapError):
        ops.beam_step(torch.zeros(1, 17, 20, device=dev()), torch.zeros(1, 17, device=dev()),
                      torch.zeros(1, 17, dtype=torch.uint8, device=dev()),
                      torch.zeros(1, 17, dtype=torch.int32, device=dev()), 0, 0, 1.0)


@pytest.mark.parametrize("activation", ["relu", "sigmoid", "linear", None, "softsign_callable", "torch_relu"])
def test_bahdanau_attention_layer_with_another_activation(activation):
    """BahdanauAttention(activation=...) (attention/attention.py:31,47): the named ones inside
    the kernel, any other callable applied to the caller's tensor."""
    import torch
    from image_caption_amd.attention.attention import BahdanauAttention
    rng = np.random.default_rng(17)
    N, L, F, U, Hh = 6, 9, 24, 16, 12
    p = {"att_W1_kernel": (rng.standard_normal((F, U)) * 0.3).astype(np.float32),
         "att_W1_bias": (0.1 * rng.standard_normal(U)).astype(np.float32),
         "att_W2_kernel": (rng.standard_normal((Hh, U)) * 0.3).astype(np.float32),
         "att_W2_bias": (0.1 * rng.standard_normal(U)).astype(np.float32),
         "att_V_kernel": rng.standard_normal((U, 1)).astype(np.float32),
         "att_V_bias": np.asarray([0.2], np.float32)}
    f = rng.standard_normal((N, L, F)).astype(np.float32)
    h = rng.standard_normal((N, Hh)).astype(np.float32)
    np_act = {"relu": lambda x: np.maximum(x, 0), "torch_relu": lambda x: np.maximum(x, 0),
              "sigmoid": co.sigmoid, "linear": lambda x: x, None: lambda x: x,
              "softsign_callable": lambda x: x / (1 + np.abs(x))}[activation]
    arg = {"softsign_callable": torch.nn.functional.softsign, "torch_relu": torch.relu}.get(activation, activation)
    layer = BahdanauAttention(U, activation=arg, device=dev())
    layer.W1.set_weights(p["att_W1_kernel"], p["att_W1_bias"], dev())
    layer.W2.set_weights(p["att_W2_kernel"], p["att_W2_bias"], dev())
    layer.V.set_weights(p["att_V_kernel"], p["att_V_bias"], dev())
    ctx, w = layer(to_dev(f), to_dev(h))
    ctx_ref, w_ref = co.bahdanau_attention(f, h, p, activation=np_act)
    assert tuple(w.shape) == (N, L, 1)
    assert_close(np_of(w), w_ref, what="weights")
    assert_close(np_of(ctx), ctx_ref, what="context")


def test_operand_devices_left_by_a_failed_call_do_not_leak_into_the_next():
    """The operators note the device of every checked operand per call.  A call whose argument
    list raises after a check never resolves its set; the next call (another frame) must drop
    it instead of reporting operands on different devices."""
    from image_caption_amd import ops
    x = to_dev(np.ones((4, 8), np.float32))
    k = to_dev(np.ones((8, 8), np.float32))
    ops._pending().add(torch.device("cuda", 1))      # what an interrupted call on another GPU leaves
    ops._tls.frame = object()
    y = ops.dense(x, k)
    torch.cuda.synchronize()
    assert_close(np_of(y), np.full((4, 8), 8.0, np.float32), what="dense after a stale operand set")
    t = ops.position_table(6, 8, dev())
    assert t.device == dev() and not ops._pending()
