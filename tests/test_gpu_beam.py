"""Fused beam-search kernel vs the oracle, through the reference-shaped module API.
Integer outputs (ids, parents, lengths, finished) are bit-exact; scores and
log-probs within 1e-5 relative.  Includes the reference's own golden vectors."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import caption_oracle as co
from tests.helpers import assert_close, dev, np_of, to_dev

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def bs():
    from image_caption_amd.beam_search import beam_search as _bs
    return _bs


def _golden(golden_dir):
    with open(os.path.join(golden_dir, "beam_search_reference_goldens.json")) as f:
        return json.load(f)


def test_reference_golden_single(bs, golden_dir):
    g = _golden(golden_dir)["single"]
    cfg = bs.BeamSearchConfig(g["beam_width"], g["vocab_size"], g["eos_token"],
                              g["length_penalty_weight"], bs.choose_top_k)
    state = bs.create_initial_beam_state(g["beam_width"], device=dev())
    seq = []
    for t, lg in enumerate(g["logits"]):
        out, state = bs.beam_search_step(t, to_dev(np.asarray(lg, np.float32)), state, cfg)
        seq.append((out, state))
    result = [int(e.item()) for e in bs.loc_optimal_beam_path(seq)]
    assert result == g["expected_sequence"]


def test_reference_golden_batch(bs, golden_dir):
    g = _golden(golden_dir)["batch"]
    cfg = bs.BeamSearchConfig(g["beam_width"], g["vocab_size"], g["eos_token"],
                              g["length_penalty_weight"], bs.choose_top_k)
    state = bs.create_initial_beam_state_batch(g["batch_size"], g["beam_width"], device=dev())
    seq = []
    for t, lg in enumerate(g["logits"]):
        out, state = bs.beam_search_step_batch(t, to_dev(np.asarray(lg, np.float32)), state, cfg)
        seq.append((out, state))
    path = bs.loc_optimal_beam_path_batch(seq)
    result = [np.reshape(np_of(e), (-1)) for e in path]
    assert result[0].tolist() == g["expected_time_major"][0]
    assert result[1].tolist() == g["expected_time_major"][1]
    assert path[0].dtype == torch.float32 and tuple(path[0].shape) == (2, 1)
    im = g["intermediates"]
    np.testing.assert_allclose(np_of(seq[0][0].scores), np.asarray(im["step0_scores"], np.float32),
                               rtol=2e-6)
    assert np_of(seq[1][0].predicted_ids).tolist() == im["step1_ids"]
    assert np_of(seq[1][0].beam_parent_ids).tolist() == im["step1_parents"]
    assert np_of(seq[1][1].finished).tolist() == im["step1_finished"]
    assert np_of(seq[1][1].lengths).tolist() == im["step1_lengths"]


def _random_state(rng, B, W, t):
    if t == 0:
        return co.create_initial_beam_state_batch(B, W)
    lp = -np.abs(rng.standard_normal((B, W))).astype(np.float32) * 3
    fin = rng.random((B, W)) < 0.3
    ln = rng.integers(1, t + 1, size=(B, W)).astype(np.int32)
    return co.BeamSearchState(lp, fin, ln)


def _check_step(bs, logits, state, t, W, V, eos, lpw, start_offset=0, prob_fn=None,
                exact=True):
    cfg_o = co.BeamSearchConfig(W, V, eos, lpw, co.choose_top_k)
    cfg_g = bs.BeamSearchConfig(W, V, eos, lpw, bs.choose_top_k)
    prob_o = None
    prob_g = None
    if prob_fn:
        prob_o = lambda x: co.log_softmax(x * np.float32(0.5))
        prob_g = lambda x: torch.log_softmax(x * 0.5, dim=-1)
    out_o, st_o = co.beam_search_step_batch(t, logits, state, cfg_o, prob_fn=prob_o,
                                            start_offset=start_offset)
    st_g = bs.BeamSearchState(to_dev(state.log_probs), to_dev(state.finished),
                              to_dev(state.lengths))
    out_g, nst_g = bs.beam_search_step_batch(t, to_dev(logits), st_g, cfg_g, prob_fn=prob_g,
                                             start_offset=start_offset)
    ids_g, par_g = np_of(out_g.predicted_ids), np_of(out_g.beam_parent_ids)
    if exact:
        np.testing.assert_array_equal(ids_g, out_o.predicted_ids)
        np.testing.assert_array_equal(par_g, out_o.beam_parent_ids)
        np.testing.assert_array_equal(np_of(nst_g.finished), st_o.finished)
        np.testing.assert_array_equal(np_of(nst_g.lengths), st_o.lengths)
        with np.errstate(over="ignore"):
            assert_close(np.maximum(np_of(out_g.scores), -1e30), np.maximum(out_o.scores, -1e30),
                         what="scores")
            assert_close(np.maximum(np_of(nst_g.log_probs), -1e30),
                         np.maximum(st_o.log_probs, -1e30), what="log_probs")
    assert out_g.predicted_ids.dtype == torch.int32 and nst_g.finished.dtype == torch.bool
    assert nst_g.lengths.dtype == torch.int32 and out_g.scores.dtype == torch.float32
    return out_o, st_o, out_g, nst_g


@pytest.mark.parametrize("B,W,V,t,lpw", [
    (1, 1, 5, 0, 1.0), (2, 2, 5, 1, 1.2), (3, 3, 11, 0, 1.0), (3, 3, 11, 2, 1.0),
    (4, 5, 37, 3, 0.0), (2, 10, 50, 4, 0.7), (2, 16, 64, 5, 1.0), (5, 3, 1000, 2, 1.0),
    (3, 3, 10000, 0, 1.0), (3, 3, 10000, 7, 1.0),     # LDS-staged production shape
    (2, 5, 10000, 3, 1.0),                            # 200 KB: re-read path
    (2, 10, 50000, 2, 1.0),                           # large-vocab stress shape
    (2, 3, 10001, 2, 1.0), (2, 4, 4099, 1, 1.0)])     # V % 4 != 0
def test_beam_step_random(bs, B, W, V, t, lpw):
    rng = np.random.default_rng(B * 1000 + W * 100 + V + t)
    logits = rng.standard_normal((B, W, V)).astype(np.float32) * 2
    state = _random_state(rng, B, W, t)
    eos = int(rng.integers(0, V))
    _check_step(bs, logits, state, t, W, V, eos, lpw)


@pytest.mark.parametrize("V", [5, 37, 1000, 10000, 10001, 50000, 70000])
@pytest.mark.parametrize("t", [0, 3])
def test_beam_step_greedy_in_one_launch(bs, V, t):
    """Beam width 1 with one row per image takes the row kernel that also finishes the step
    (finished rows, gathers, next state): against the oracle for every row-length form of the
    kernel, with finished rows, eos at the first / last column and a tie on the maximum."""
    rng = np.random.default_rng(V + t)
    B = 24
    logits = rng.standard_normal((B, 1, V)).astype(np.float32) * 2
    logits[3, 0, V - 1] = logits[3, 0, 1] = logits[3, 0].max() + 1.0    # a tie: the lower column wins
    state = _random_state(rng, B, 1, t)
    for eos in (0, V - 1, int(rng.integers(0, V))):
        out_o, st_o, _, _ = _check_step(bs, logits, state, t, 1, V, eos, 1.0)
        if t > 0:
            assert state.finished.any() and not state.finished.all()
        if eos not in (1, V - 1):       # (the eos column is scored with its own length penalty)
            assert out_o.predicted_ids[3, 0] == (eos if state.finished[3, 0] else 1)


def test_beam_step_exact_ties_prefer_lower_flat_index(bs):
    """Identical rows and quantised logits give bit-identical scores on both sides;
    tf.nn.top_k order (lower flat index first) must be reproduced."""
    rng = np.random.default_rng(5)
    B, W, V = 4, 3, 2000
    row = np.round(rng.standard_normal((B, 1, V)) * 2).astype(np.float32) / 2   # many duplicates
    logits = np.repeat(row, W, axis=1)
    lp = np.repeat(-np.abs(rng.standard_normal((B, 1))).astype(np.float32), W, axis=1)
    state = co.BeamSearchState(lp, np.zeros((B, W), bool), np.full((B, W), 2, np.int32))
    out_o, _, _, _ = _check_step(bs, logits, state, 3, W, V, 7, 1.0)
    assert (out_o.scores[:, 0] == out_o.scores[:, 1]).all()    # the case really has ties


def test_beam_step_all_finished_and_partly_finished(bs):
    rng = np.random.default_rng(6)
    B, W, V = 3, 3, 50
    logits = rng.standard_normal((B, W, V)).astype(np.float32)
    lp = -np.abs(rng.standard_normal((B, W))).astype(np.float32)
    fin = np.array([[True, True, True], [True, False, True], [False, False, False]])
    ln = np.array([[2, 3, 1], [1, 4, 2], [3, 3, 3]], np.int32)
    for eos in (0, 4, 49):
        _check_step(bs, logits, co.BeamSearchState(lp, fin, ln), 4, W, V, eos, 1.0)


def test_beam_step_start_offset_and_custom_prob_fn(bs):
    rng = np.random.default_rng(7)
    B, W, V = 3, 4, 23
    logits = rng.standard_normal((B, W, V)).astype(np.float32)
    state = co.create_initial_beam_state_batch(B, W)
    _check_step(bs, logits, state, 0, W, V, 2, 1.0, start_offset=1)
    _check_step(bs, logits, _random_state(rng, B, W, 3), 3, W, V, 2, 1.0, prob_fn=True)


@pytest.mark.parametrize("B,R,W,V,t", [
    (3, 4, 3, 23, 0), (3, 4, 3, 23, 2), (2, 6, 5, 1000, 3), (4, 2, 1, 40, 1), (2, 1, 3, 50, 2),
    (2, 11, 10, 300, 4), (3, 4, 3, 10000, 5), (2, 16, 3, 64, 2), (2, 3, 16, 64, 1)])
def test_beam_step_rows_differ_from_beam_width(bs, B, R, W, V, t):
    """Gold + search rows of the beam-search-optimisation trainer (utils/train_utils.py:
    275-279: R = W + 1 rows in, W successors out, start_offset = 1 at time 0) and other row
    counts: parents index the R input rows."""
    rng = np.random.default_rng(B * 100 + R * 10 + W + V + t)
    logits = rng.standard_normal((B, R, V)).astype(np.float32) * 2
    state = _random_state(rng, B, R, t)
    eos = int(rng.integers(0, V))
    off = 1 if R > 1 else 0
    out_o, _, out_g, nst_g = _check_step(bs, logits, state, t, W, V, eos, 1.2, start_offset=off)
    assert tuple(out_g.scores.shape) == (B, W) and tuple(nst_g.lengths.shape) == (B, W)
    if t > 0:
        assert int(np_of(out_g.beam_parent_ids).max()) < R
    # the trainer's gold step: one row, a mask added to the log-softmax (GoldProbFn, :70-78)
    mask = np.full((V,), np.float32(-1e9))
    mask[int(rng.integers(0, V))] = 0.0
    gold_o = lambda x: co.log_softmax(x) + mask
    gold_g = lambda x: torch.log_softmax(x, dim=-1) + to_dev(mask)
    st1 = _random_state(rng, B, 1, t)
    cfg_o = co.BeamSearchConfig(1, V, eos, 1.2, co.choose_top_k)
    cfg_g = bs.BeamSearchConfig(1, V, eos, 1.2, bs.choose_top_k)
    o_o, s_o = co.beam_search_step_batch(t, logits[:, :1], st1, cfg_o, prob_fn=gold_o)
    o_g, s_g = bs.beam_search_step_batch(
        t, to_dev(logits[:, :1].copy()),
        bs.BeamSearchState(to_dev(st1.log_probs), to_dev(st1.finished), to_dev(st1.lengths)),
        cfg_g, prob_fn=gold_g)
    np.testing.assert_array_equal(np_of(o_g.predicted_ids), o_o.predicted_ids)
    with np.errstate(over="ignore"):
        assert_close(np.maximum(np_of(s_g.log_probs), -1e30), np.maximum(s_o.log_probs, -1e30),
                     what="gold log_probs")


def test_beam_step_monotone_logits_worst_case_insertion(bs):
    """Increasing logits make every element a new maximum (the slow insertion path)."""
    B, W, V = 2, 3, 3000
    logits = np.tile(np.linspace(-3, 3, V, dtype=np.float32), (B, W, 1))
    logits[:, 1] *= 1.01      # distinct distributions per beam (a constant shift would
    logits[:, 2] *= 0.98      # leave the log-softmax, and so the scores, tied)
    state = co.BeamSearchState(np.zeros((B, W), np.float32), np.zeros((B, W), bool),
                               np.ones((B, W), np.int32))
    _check_step(bs, logits, state, 2, W, V, 5, 1.0)


def test_multi_step_search_and_backtrack_match_oracle(bs):
    rng = np.random.default_rng(8)
    B, W, V, T, eos = 6, 3, 40, 9, 1
    cfg_o = co.BeamSearchConfig(W, V, eos, 1.0, co.choose_top_k)
    cfg_g = bs.BeamSearchConfig(W, V, eos, 1.0, bs.choose_top_k)
    st_o = co.create_initial_beam_state_batch(B, W)
    st_g = bs.create_initial_beam_state_batch(B, W, device=dev())
    seq_o, seq_g = [], []
    for t in range(T):
        logits = rng.standard_normal((B, W, V)).astype(np.float32) * 1.5
        logits[:, :, eos] += 1.0                      # make EOS likely so beams finish
        out_o, st_o = co.beam_search_step_batch(t, logits, st_o, cfg_o)
        out_g, st_g = bs.beam_search_step_batch(t, to_dev(logits), st_g, cfg_g)
        np.testing.assert_array_equal(np_of(out_g.predicted_ids), out_o.predicted_ids)
        np.testing.assert_array_equal(np_of(out_g.beam_parent_ids), out_o.beam_parent_ids)
        np.testing.assert_array_equal(np_of(st_g.finished), st_o.finished)
        seq_o.append((out_o, st_o))
        seq_g.append((out_g, st_g))
    assert st_o.finished.any()
    path_o = np.concatenate(co.loc_optimal_beam_path_batch(seq_o), axis=1)
    path_g = np.concatenate([np_of(e) for e in bs.loc_optimal_beam_path_batch(seq_g)], axis=1)
    np.testing.assert_array_equal(path_g, path_o)


def test_choose_top_k_direct(bs):
    x = np.array([[1.0, 3.0, 3.0, 2.0, 3.0, -1.0], [0.5, 0.25, 4.0, 4.0, -2.0, 0.0]], np.float32)
    v, i = bs.choose_top_k(to_dev(x), 3)
    v_o, i_o = co.choose_top_k(x, 3)
    np.testing.assert_array_equal(np_of(i), i_o)
    np.testing.assert_array_equal(np_of(v), v_o)


def test_empty_sequence_backtrack(bs):
    assert bs.loc_optimal_beam_path_batch([]) == []
    assert bs.loc_optimal_beam_path([]) == []


# ---- unfused step: caller-supplied successor choice, and the stand-alone helpers ----

def _even_top_k_np(scores_flat, k):
    """A deliberately odd successor choice: the best k among even flat indices."""
    masked = np.where(np.arange(scores_flat.shape[-1]) % 2 == 0, scores_flat, -np.inf)
    return co.choose_top_k(masked, k)


def _even_top_k_torch(scores_flat, k):
    n = scores_flat.shape[-1]
    even = (torch.arange(n, device=scores_flat.device) % 2 == 0)
    masked = torch.where(even, scores_flat, torch.full_like(scores_flat, float("-inf")))
    v, i = torch.topk(masked, k, dim=-1)
    return v, i


@pytest.mark.parametrize("B,W,V,t,start_offset", [
    (2, 3, 11, 0, 0), (2, 3, 11, 0, 2), (3, 3, 40, 2, 0), (2, 5, 10000, 3, 0), (2, 4, 4099, 1, 0)])
def test_custom_successor_fn_step(bs, B, W, V, t, start_offset):
    rng = np.random.default_rng(100 + B * 7 + W * 3 + V + t)
    logits = rng.standard_normal((B, W, V)).astype(np.float32) * 2
    state = _random_state(rng, B, W, t)
    eos, lpw = 4, 1.0
    out_o, st_o = co.beam_search_step_batch(
        t, logits, state, co.BeamSearchConfig(W, V, eos, lpw, _even_top_k_np), start_offset=start_offset)
    st_g = bs.BeamSearchState(to_dev(state.log_probs), to_dev(state.finished), to_dev(state.lengths))
    out_g, nst_g = bs.beam_search_step_batch(
        t, to_dev(logits), st_g, bs.BeamSearchConfig(W, V, eos, lpw, _even_top_k_torch),
        start_offset=start_offset)
    np.testing.assert_array_equal(np_of(out_g.predicted_ids), out_o.predicted_ids)
    np.testing.assert_array_equal(np_of(out_g.beam_parent_ids), out_o.beam_parent_ids)
    np.testing.assert_array_equal(np_of(nst_g.finished), st_o.finished)
    np.testing.assert_array_equal(np_of(nst_g.lengths), st_o.lengths)
    with np.errstate(over="ignore"):
        assert_close(np.maximum(np_of(out_g.scores), -1e30), np.maximum(out_o.scores, -1e30), what="scores")
        assert_close(np.maximum(np_of(nst_g.log_probs), -1e30), np.maximum(st_o.log_probs, -1e30),
                     what="log_probs")
    assert out_g.predicted_ids.dtype == torch.int32 and nst_g.finished.dtype == torch.bool


def test_custom_successor_fn_equals_fused_when_it_is_top_k(bs):
    rng = np.random.default_rng(5)
    B, W, V, t = 3, 3, 1000, 2
    logits = rng.standard_normal((B, W, V)).astype(np.float32)
    state = _random_state(rng, B, W, t)
    st_g = bs.BeamSearchState(to_dev(state.log_probs), to_dev(state.finished), to_dev(state.lengths))
    topk = lambda s, k: torch.topk(s, k, dim=-1)
    out_a, st_a = bs.beam_search_step_batch(t, to_dev(logits), st_g,
                                            bs.BeamSearchConfig(W, V, 4, 1.0, topk))
    out_b, st_b = bs.beam_search_step_batch(t, to_dev(logits), st_g,
                                            bs.BeamSearchConfig(W, V, 4, 1.0, bs.choose_top_k))
    np.testing.assert_array_equal(np_of(out_a.predicted_ids), np_of(out_b.predicted_ids))
    np.testing.assert_array_equal(np_of(out_a.beam_parent_ids), np_of(out_b.beam_parent_ids))
    np.testing.assert_array_equal(np_of(st_a.lengths), np_of(st_b.lengths))
    assert_close(np.maximum(np_of(st_a.log_probs), -1e30), np.maximum(np_of(st_b.log_probs), -1e30))


def test_custom_successor_fn_unbatched_and_bad_index(bs):
    rng = np.random.default_rng(6)
    W, V = 3, 13
    logits = rng.standard_normal((W, V)).astype(np.float32)
    st_o = co.BeamSearchState(-np.abs(rng.standard_normal(W)).astype(np.float32),
                              np.array([False, True, False]), np.array([2, 1, 2], np.int32))
    out_o, nst_o = co.beam_search_step(1, logits, st_o, co.BeamSearchConfig(W, V, 4, 1.0, _even_top_k_np))
    seen = {}

    def fn(scores_flat, k):
        seen["shape"] = tuple(scores_flat.shape)
        return _even_top_k_torch(scores_flat, k)
    st_g = bs.BeamSearchState(to_dev(st_o.log_probs), to_dev(st_o.finished), to_dev(st_o.lengths))
    out_g, nst_g = bs.beam_search_step(1, to_dev(logits), st_g, bs.BeamSearchConfig(W, V, 4, 1.0, fn))
    assert seen["shape"] == (W * V,)
    np.testing.assert_array_equal(np_of(out_g.predicted_ids), out_o.predicted_ids)
    np.testing.assert_array_equal(np_of(out_g.beam_parent_ids), out_o.beam_parent_ids)
    np.testing.assert_array_equal(np_of(nst_g.lengths), nst_o.lengths)
    # an index outside [0, W*V): no exception (no host round trip on the step path) - the
    # result is what the reference's gathers give on a GPU, zeros from the out-of-range rows
    bad = lambda s, k: (s[..., :k], torch.full_like(s[..., :k], W * V + 5).to(torch.int32))
    out_b, nst_b = bs.beam_search_step(1, to_dev(logits), st_g, bs.BeamSearchConfig(W, V, 4, 1.0, bad))
    np.testing.assert_array_equal(np_of(out_b.predicted_ids), np.full(W, 5))
    np.testing.assert_array_equal(np_of(out_b.beam_parent_ids), np.full(W, W))
    np.testing.assert_array_equal(np_of(nst_b.log_probs), np.zeros(W, np.float32))
    np.testing.assert_array_equal(np_of(nst_b.lengths), np.ones(W, np.int32))
    assert not np_of(nst_b.finished).any()


def test_helper_functions_match_oracle(bs):
    rng = np.random.default_rng(8)
    N, V, eos = 6, 37, 4
    probs = co.log_softmax(rng.standard_normal((N, V)).astype(np.float32))
    fin = rng.random(N) < 0.5
    got = np_of(bs.mask_probs(to_dev(probs), eos, to_dev(fin)))
    np.testing.assert_array_equal(got, co.mask_probs(probs, eos, fin))
    lens = rng.integers(0, 90, size=(4, 5)).astype(np.int32)     # beyond the 64-entry table too
    for w in (0.0, 0.7, 1.0, 1.2):
        assert_close(np_of(bs.length_penalty(to_dev(lens), w)), co.length_penalty(lens, w), rtol=2e-6)
        lp = -np.abs(rng.standard_normal((4, 5))).astype(np.float32) * 10
        assert_close(np_of(bs.hyp_score(to_dev(lp), to_dev(lens), w)), co.hyp_score(lp, lens, w),
                     rtol=2e-6)
    # small lengths come from a host libm table: bit-exact against numpy there
    small = np.arange(0, 40, dtype=np.int32)[None]
    np.testing.assert_array_equal(np_of(bs.length_penalty(to_dev(small), 1.0)),
                                  co.length_penalty(small, 1.0))
    lp3 = rng.standard_normal((2, 3, 1)).astype(np.float32)
    ln3 = rng.integers(0, 9, size=(2, 3, 7)).astype(np.int32)
    assert_close(np_of(bs.hyp_score(to_dev(lp3), to_dev(ln3), 1.0)), co.hyp_score(lp3, ln3, 1.0),
                 rtol=2e-6)
    a = rng.standard_normal((5, 9)).astype(np.float32)
    idx = rng.integers(0, 9, size=(5, 4)).astype(np.int32)
    g = bs.batch_gather(to_dev(a), to_dev(idx))
    assert g.dtype == torch.float32
    np.testing.assert_array_equal(np_of(g), co.batch_gather(a, idx))
    ai = rng.integers(0, 50, size=(5, 9)).astype(np.int32)
    np.testing.assert_array_equal(np_of(bs.batch_gather(to_dev(ai), to_dev(idx))), co.batch_gather(ai, idx))
    # out of range: 0, like tf.gather on a GPU
    oob = np_of(bs.batch_gather(to_dev(a), to_dev(np.array([[9], [-1], [0], [8], [100]], np.int32))))
    np.testing.assert_array_equal(oob[:, 0], [0.0, 0.0, a[2, 0], a[3, 8], 0.0])


# ---- fused vocab projection + beam step (no logits in memory) ----

@pytest.mark.parametrize("W,boost", [(3, 6.0), (5, 6.0), (3, -6.0)])
def test_records_step_with_eos_inside_or_outside_the_recorded_best(bs, W, boost):
    """The eos column is scored apart from a row's recorded logits: as the best logit of its
    half tile it leaves a hole in the middle of a record's sorted list, as the worst it is
    not recorded at all; both must rank like the step on full logits."""
    from image_caption_amd import ops
    B, K, V, eos, t = 6, 64, 3000, 70, 3
    rng = np.random.default_rng(17 + W)
    N = B * W
    x = rng.standard_normal((N, K)).astype(np.float32)
    w = (rng.standard_normal((K, V)) * 0.3).astype(np.float32)
    bias = rng.standard_normal(V).astype(np.float32) * 0.1
    bias[eos] += boost
    bias[eos + 1] += abs(boost) * 0.5      # a strong neighbour in the same half tile
    state = _random_state(rng, B, W, t)
    lp, fin, ln = to_dev(state.log_probs), to_dev(state.finished.astype(np.uint8)), to_dev(state.lengths)
    top_k = 3 if W <= 3 else 5
    logits = ops.dense(to_dev(x), to_dev(w), to_dev(bias))
    ref = ops.beam_step(logits.reshape(B, W, V), lp, fin, ln, t, eos, 1.0)
    rec, eos_x = ops.dense_topk(to_dev(x), to_dev(w), to_dev(bias), eos, top_k)
    got = ops.beam_step_from_records(rec, eos_x, lp, fin, ln, t, eos, 1.0, V)
    for name, g, r in zip(["scores", "ids", "parents", "next_log_probs", "next_finished", "next_lengths"], got, ref):
        g, r = np_of(g), np_of(r)
        if g.dtype == np.float32:
            with np.errstate(over="ignore"):
                assert_close(np.maximum(g, -1e30), np.maximum(r, -1e30), rtol=2e-6, what=name)
        else:
            np.testing.assert_array_equal(g, r, err_msg=name)
    if boost > 0:
        assert (np_of(got[1]) == eos).any()    # eos does get chosen somewhere


@pytest.mark.parametrize("B,W,K,V,t,eos", [
    (4, 3, 64, 10000, 2, 4), (3, 3, 64, 10000, 0, 4), (5, 1, 32, 1000, 3, 4), (3, 5, 96, 5000, 4, 4),
    (2, 3, 32, 132, 1, 4),        # last 128-column tile has 4 valid columns, its second half none
    (2, 2, 32, 64, 2, 70),        # eos outside the vocabulary
    (70, 3, 64, 1028, 5, 1027),   # several row tiles, eos in the last column
    (3, 4, 64, 5000, 3, 4),       # beam narrower than the records are deep
    (2, 3, 32, 20000, 2, 4),      # five records per lane of a row's wave
    (2, 3, 32, 40000, 2, 4)])     # more than 512 records per row: the one-workgroup form
def test_projection_topk_records_equal_unfused_step(bs, B, W, K, V, t, eos):
    from image_caption_amd import ops
    rng = np.random.default_rng(B * 31 + W * 7 + V + t)
    N = B * W
    x = rng.standard_normal((N, K)).astype(np.float32)
    w = (rng.standard_normal((K, V)) * 0.3).astype(np.float32)
    w[:, 7] = w[:, 3]                       # an exact tie inside one half tile
    if V > 200:
        w[:, 150] = w[:, 3]                 # and across tiles
    bias = rng.standard_normal(V).astype(np.float32) * 0.1
    bias[7] = bias[3]
    if V > 200:
        bias[150] = bias[3]
    state = _random_state(rng, B, W, t)
    lp, fin, ln = to_dev(state.log_probs), to_dev(state.finished.astype(np.uint8)), to_dev(state.lengths)
    top_k = 1 if W == 1 else (3 if W <= 3 else 5)
    logits = ops.dense(to_dev(x), to_dev(w), to_dev(bias))
    ref = ops.beam_step(logits.reshape(B, W, V), lp, fin, ln, t, eos, 1.0)
    rec, eos_x = ops.dense_topk(to_dev(x), to_dev(w), to_dev(bias), eos, top_k)
    got = ops.beam_step_from_records(rec, eos_x, lp, fin, ln, t, eos, 1.0, V)
    names = ["scores", "ids", "parents", "next_log_probs", "next_finished", "next_lengths"]
    for name, g, r in zip(names, got, ref):
        g, r = np_of(g), np_of(r)
        if g.dtype == np.float32:
            with np.errstate(over="ignore"):
                assert_close(np.maximum(g, -1e30), np.maximum(r, -1e30), rtol=2e-6, what=name)
        else:
            np.testing.assert_array_equal(g, r, err_msg=name)
    # the records themselves: max and best logit of every half tile against the logits
    lg = np_of(logits)
    recs = np_of(rec)
    ht = recs.shape[1]
    pad = np.full((N, ht * 64), -np.inf, np.float32)
    pad[:, :V] = lg
    tiles = pad.reshape(N, ht, 64)
    # (ops.dense may pick a split-K kernel for this shape, so its sums can differ from the
    # projection's by an ulp or two: compare values with the fp32 bar, columns through values)
    tmax = tiles.max(axis=2)
    live = np.isfinite(tmax)
    assert_close(np.where(live, recs[:, :, 0], 0), np.where(live, tmax, 0), what="record max")
    np.testing.assert_array_equal(recs[:, :, 2], recs[:, :, 0])
    assert (np.isneginf(recs[:, :, 0]) == ~live).all()
    cols = recs[:, :, 2 + top_k].view(np.int32)
    assert (cols[~live] == 0x7fffffff).all()
    cl = np.where(live, cols, 0)
    assert ((cl >= 0) & (cl < V)).all() and ((cl // 64) == np.arange(ht)[None])[live].all()
    at_col = np.take_along_axis(lg, cl.reshape(N, -1), axis=1).reshape(N, ht)
    assert_close(np.where(live, at_col, 0), np.where(live, tmax, 0), what="logit at the recorded column")
    ssum = np.where(live, np.exp(tiles - np.where(live, tmax, 0)[..., None]).sum(axis=2), 0)
    assert_close(np.where(live, recs[:, :, 1], 0), ssum, what="record sum exp")
    if 0 <= eos < V:
        assert_close(np_of(eos_x), lg[:, eos], what="eos logit")
    if not 0 <= eos < V:
        return
    # and against the oracle
    out_o, st_o = co.beam_search_step_batch(t, lg.reshape(B, W, V), state,
                                            co.BeamSearchConfig(W, V, eos, 1.0, co.choose_top_k))
    np.testing.assert_array_equal(np_of(got[1]), out_o.predicted_ids)
    np.testing.assert_array_equal(np_of(got[2]), out_o.beam_parent_ids)
    np.testing.assert_array_equal(np_of(got[5]), st_o.lengths)


def test_projection_topk_rejects_unsupported(bs):
    from image_caption_amd import ops
    x = torch.zeros(4, 32, device="cuda")
    with pytest.raises(ValueError):
        ops.dense_topk(x, torch.zeros(32, 64, device="cuda"), None, 4, 2)
    with pytest.raises(RuntimeError, match="N % 4"):
        ops.dense_topk(x, torch.zeros(32, 65, device="cuda"), None, 4, 3)
