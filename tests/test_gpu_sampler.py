"""Sequence sampler of the policy-gradient trainers (reference utils/agent_utils.py:34-132).

The reference draws with tf.random.categorical, whose random stream cannot be reproduced
without TensorFlow: parity unpinned for the individual draws.  What is checked: the draw is the
inverse of the cumulative distribution at the caller's uniform numbers (against the oracle's
float64 restatement), masked columns are never drawn, the forcing by position is exact, and
the frequencies over many draws follow the softmax."""
import numpy as np
import pytest
import torch

from oracle import caption_oracle as co
from tests.helpers import dev, np_of, to_dev
from tests.test_gpu_searcher import _params, _searcher_modules, _sequences

pytestmark = pytest.mark.gpu


def _near_boundary(logits, u, a, b, toks):
    """True when u * total lies within fp32 rounding of the cumulative mass between columns a
    and b (the two sides may then fall on different sides of it)."""
    x = logits.astype(np.float64).copy()
    for t in toks:
        if t >= 0:
            x[t] = -np.inf
    e = np.exp(x - x.max())
    cdf = np.cumsum(e)
    lo, hi = min(a, b), max(a, b)
    return bool(np.any(np.abs(cdf[lo:hi] - u * cdf[-1]) <= 2e-5 * cdf[-1]))


@pytest.mark.parametrize("N,V", [(7, 5), (64, 37), (50, 1000), (33, 10000), (9, 10001)])
@pytest.mark.parametrize("masked", [False, True])
def test_sample_is_the_inverse_cdf_at_u(N, V, masked):
    from image_caption_amd import ops
    rng = np.random.default_rng(N + V)
    logits = (rng.standard_normal((N, V)) * 3).astype(np.float32)
    u = rng.random(N).astype(np.float32)
    u[0], u[-1] = 0.0, np.float32(1.0 - 2.0 ** -24)
    toks = (3, 4, 0) if masked else (-1, -1, -1)
    got = np_of(ops.sample_categorical(to_dev(logits), to_dev(u), start_token=toks[0],
                                       eos_token=toks[1], pad_token=toks[2]))
    ref = co.sample_categorical(logits, u, start_token=toks[0], eos_token=toks[1], pad_token=toks[2])
    assert got.dtype == np.int32 and got.min() >= 0 and got.max() < V
    for n in np.nonzero(got != ref)[0]:
        assert _near_boundary(logits[n], float(u[n]), int(got[n]), int(ref[n]), toks), (n, got[n], ref[n])
    assert (got != ref).sum() <= 1
    if masked:
        assert not np.isin(got, toks).any()


def test_masked_columns_are_never_drawn_and_extremes():
    from image_caption_amd import ops
    V = 12
    logits = np.zeros((1000, V), np.float32)
    logits[:, [0, 3, 4]] = 50.0                      # the masked columns would dominate
    u = np.linspace(0.0, 1.0, 1000, endpoint=False).astype(np.float32)
    got = np_of(ops.sample_categorical(to_dev(logits), to_dev(u), start_token=3, eos_token=4,
                                       pad_token=0))
    assert not np.isin(got, [0, 3, 4]).any()
    assert got[0] == 1 and got[-1] == V - 1          # first / last column with mass
    assert (np.diff(got) >= 0).all()                 # the inverse of a CDF is monotone in u
    counts = np.bincount(got, minlength=V)[[1, 2, 5, 6, 7, 8, 9, 10, 11]]
    assert counts.min() >= 110 and counts.max() <= 112    # nine equal shares of 1000


def test_forcing_by_position():
    from image_caption_amd import ops
    rng = np.random.default_rng(1)
    N, V = 8, 20
    logits = rng.standard_normal((N, V)).astype(np.float32)
    u = rng.random(N).astype(np.float32)
    row_len = np.array([1, 2, 3, 4, 5, 6, 7, 8], np.int32)
    for step in range(1, 8):
        got = np_of(ops.sample_categorical(to_dev(logits), to_dev(u), step=step, row_len=to_dev(row_len),
                                           start_token=3, eos_token=4, pad_token=0))
        ref = co.sample_categorical(logits, u, step=step, row_len=row_len, start_token=3, eos_token=4,
                                    pad_token=0)
        np.testing.assert_array_equal(got, ref)
        assert (got[row_len == step + 1] == 4).all() and (got[row_len <= step] == 0).all()


def test_frequencies_follow_the_softmax():
    from image_caption_amd import ops
    V, N = 16, 40000
    rng = np.random.default_rng(2)
    row = (rng.standard_normal(V) * 1.5).astype(np.float32)
    p = np.exp(row - row.max())
    p /= p.sum()
    g = torch.Generator(device=dev())
    g.manual_seed(7)
    u = torch.rand(N, dtype=torch.float32, device=dev(), generator=g)
    got = np_of(ops.sample_categorical(to_dev(np.tile(row, (N, 1))), u))
    freq = np.bincount(got, minlength=V) / N
    sigma = np.sqrt(p * (1 - p) / N)
    assert (np.abs(freq - p) <= 5 * sigma + 1e-4).all(), (freq, p)


@pytest.mark.parametrize("cell", ["lstm", "gru"])
def test_sequence_sampler_class_follows_the_oracle_loop(cell):
    from image_caption_amd.utils.agent_utils import SequenceSampler
    dims = dict(F=24, D=8, E=8, H=32, V=40)
    p = _params(8, dims, cell, bias_seed=9)
    rng = np.random.default_rng(10)
    B, L, T = 6, 4, 9
    feats = np.abs(rng.standard_normal((B, L, dims["D"]))).astype(np.float32)
    seqs = _sequences(rng, B, T)
    uniforms = rng.random((T - 1, B)).astype(np.float32)
    ref = co.sequence_sampler(p, feats, seqs, uniforms, 3, 4, 0, cell=cell)
    net = _searcher_modules(p, dims, cell)
    sampler = SequenceSampler(net, 3, 4, uniform_fn=lambda t, b: uniforms[t - 1])
    got = sampler((to_dev(feats), to_dev(seqs)), training=False)
    assert got.dtype == torch.int32 and tuple(got.shape) == (B, T)
    np.testing.assert_array_equal(np_of(got), ref)
    lens = (seqs > 0).sum(axis=1)
    for b in range(B):
        assert ref[b, 0] == 3 and ref[b, lens[b] - 1] == 4 and (ref[b, lens[b]:] == 0).all()
        assert not np.isin(ref[b, 1:lens[b] - 1], [0, 3, 4]).any()
    # the default random source: reproducible from a seeded generator, tokens in range
    g = torch.Generator(device=dev())
    g.manual_seed(3)
    a = np_of(SequenceSampler(net, 3, 4, generator=g)((to_dev(feats), to_dev(seqs)), training=False))
    g.manual_seed(3)
    b2 = np_of(SequenceSampler(net, 3, 4, generator=g)((to_dev(feats), to_dev(seqs)), training=False))
    np.testing.assert_array_equal(a, b2)
    assert a.min() >= 0 and a.max() < dims["V"]
