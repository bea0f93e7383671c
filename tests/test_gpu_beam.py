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
