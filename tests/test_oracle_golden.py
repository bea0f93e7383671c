"""The oracle against the reference's own golden vectors
(beam_search/beam_search_test.py:36-108).  CPU only."""
import json
import os

import numpy as np

from oracle import caption_oracle as co


def _load(golden_dir):
    with open(os.path.join(golden_dir, "beam_search_reference_goldens.json")) as f:
        return json.load(f)


def test_single_batch_golden(golden_dir):
    g = _load(golden_dir)["single"]
    cfg = co.BeamSearchConfig(g["beam_width"], g["vocab_size"], g["eos_token"],
                              g["length_penalty_weight"], co.choose_top_k)
    state = co.create_initial_beam_state(g["beam_width"])
    seq = []
    for t, lg in enumerate(g["logits"]):
        out, state = co.beam_search_step(t, np.asarray(lg, np.float32), state, cfg)
        seq.append((out, state))
    result = [int(x) for x in co.loc_optimal_beam_path(seq)]
    assert result == g["expected_sequence"]


def test_multiple_batch_golden(golden_dir):
    g = _load(golden_dir)["batch"]
    cfg = co.BeamSearchConfig(g["beam_width"], g["vocab_size"], g["eos_token"],
                              g["length_penalty_weight"], co.choose_top_k)
    state = co.create_initial_beam_state_batch(g["batch_size"], g["beam_width"])
    seq = []
    for t, lg in enumerate(g["logits"]):
        out, state = co.beam_search_step_batch(t, np.asarray(lg, np.float32), state, cfg)
        seq.append((out, state))
    path = co.loc_optimal_beam_path_batch(seq)
    result = [np.reshape(e, (-1)) for e in path]
    assert result[0].tolist() == g["expected_time_major"][0]
    assert result[1].tolist() == g["expected_time_major"][1]
    assert path[0].dtype == np.float32 and path[0].shape == (2, 1)
    im = g["intermediates"]
    np.testing.assert_allclose(seq[0][0].scores, np.asarray(im["step0_scores"], np.float32),
                               rtol=2e-6)
    assert seq[1][0].predicted_ids.tolist() == im["step1_ids"]
    assert seq[1][0].beam_parent_ids.tolist() == im["step1_parents"]
    assert seq[1][1].finished.tolist() == im["step1_finished"]
    assert seq[1][1].lengths.tolist() == im["step1_lengths"]


def test_batch_of_one_matches_unbatched(golden_dir):
    g = _load(golden_dir)["single"]
    cfg = co.BeamSearchConfig(g["beam_width"], g["vocab_size"], g["eos_token"],
                              g["length_penalty_weight"], co.choose_top_k)
    s1 = co.create_initial_beam_state(g["beam_width"])
    sb = co.create_initial_beam_state_batch(1, g["beam_width"])
    for t, lg in enumerate(g["logits"]):
        lg = np.asarray(lg, np.float32)
        o1, s1 = co.beam_search_step(t, lg, s1, cfg)
        ob, sb = co.beam_search_step_batch(t, lg[None], sb, cfg)
        assert ob.predicted_ids[0].tolist() == o1.predicted_ids.tolist()
        assert ob.beam_parent_ids[0].tolist() == o1.beam_parent_ids.tolist()
        np.testing.assert_array_equal(ob.scores[0], o1.scores)
        np.testing.assert_array_equal(sb.log_probs[0], s1.log_probs)


def test_mask_probs_finished_row():
    probs = np.log(np.full((2, 4), 0.25, np.float32))
    out = co.mask_probs(probs, 2, np.array([False, True]))
    np.testing.assert_array_equal(out[0], probs[0])
    assert out[1, 2] == 0.0
    assert (out[1, [0, 1, 3]] == co.F32_MIN).all()


def test_top_k_tie_prefers_lower_index():
    v, i = co.choose_top_k(np.array([[1.0, 3.0, 3.0, 2.0, 3.0]], np.float32), 3)
    assert i.tolist() == [[1, 2, 4]]
    assert v.tolist() == [[3.0, 3.0, 3.0]]
