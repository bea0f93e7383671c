"""Committed whole-sequence vectors (tests/golden/e2e_tiny_fixtures.json, written by
tests/golden/make_e2e_fixtures.py): inputs, weights, tokens and every per-step beam record of
tiny decodes.  The oracle (CPU, every run) and the HIP engine (GPU) are both held to them:
integer outputs exact, floats 1e-5 (the stored margins are 1e-4 or more, so no decision is a
near-tie).  The reference holds no fixture for this loop; these vectors pin drift, not the
reference."""
import json
import os

import numpy as np
import pytest

from oracle import caption_oracle as co

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "e2e_tiny_fixtures.json")) as f:
    FIX = json.load(f)
DIMS = FIX["dims"]
START, EOS, PAD = FIX["start_id"], FIX["eos_id"], FIX["pad_id"]
CASES = sorted(FIX["cases"])


def arr(d):
    return np.array(d["data"], dtype=np.float32 if d["dtype"] == "float32" else np.int32).reshape(d["shape"])


def load_case(name):
    c = FIX["cases"][name]
    p = {k: arr(v) for k, v in c["params"].items()}
    inputs = {k: arr(v) for k, v in c["inputs"].items()}
    steps = [{k: arr(v) for k, v in st.items()} for st in c["steps"]]
    return c, p, inputs, arr(c["tokens"]), steps


def close(got, ref, what, rtol=1e-5):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    big = ref < -1e30                       # masked candidates of finished beams (about -3.4e38)
    assert ((got < -1e30) == big).all(), what
    err = np.abs(got - ref)[~big]
    tol = rtol * np.maximum(np.abs(ref[~big]), 1.0)
    assert not np.isnan(got).any() and (err <= tol).all(), "%s: max err %.3e" % (what, err.max())


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_the_committed_vectors(name):
    c, p, inputs, tokens, steps = load_case(name)
    assert c["min_topk_margin"] >= 1e-4
    if c["loop"] == "searcher":
        got = co.baseline_sequence_beam_search(p, inputs["features"], inputs["sequences"], c["W"],
                                               DIMS["V"], 0, START, EOS, PAD, cell=c["cell"])
        np.testing.assert_array_equal(got, tokens)
        return
    got, trace = co.predict_image_caption_sequence(
        p, inputs["img"], c["W"], c["T"], START, EOS, c["length_penalty_weight"], c["tile_mode"],
        return_trace=True, cell=c["cell"])
    np.testing.assert_array_equal(got, tokens)
    for t, (st, ref) in enumerate(zip(trace, steps)):
        np.testing.assert_array_equal(st["out"].predicted_ids, ref["ids"])
        np.testing.assert_array_equal(st["out"].beam_parent_ids, ref["parents"])
        np.testing.assert_array_equal(st["state"].finished.astype(np.int32), ref["finished"])
        np.testing.assert_array_equal(st["state"].lengths, ref["lengths"])
        close(st["out"].scores, ref["scores"], "scores step %d" % t, rtol=2e-6)
        close(st["state"].log_probs, ref["log_probs"], "log_probs step %d" % t, rtol=2e-6)
        close(st["logits"][0, 0], ref["logits_beam0_image0"], "logits step %d" % t, rtol=2e-6)
    # the hoisted form of the oracle (used for the large parity runs) gives the same records
    got_h = co.predict_image_caption_sequence(
        p, inputs["img"], c["W"], c["T"], START, EOS, c["length_penalty_weight"], c["tile_mode"],
        hoist_w1=True, cell=c["cell"])
    np.testing.assert_array_equal(got_h, tokens)


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_engine_reproduces_the_committed_vectors(name):
    import torch
    from image_caption_amd.engine import CaptionConfig, CaptionEngine
    from tests.helpers import dev, np_of, to_dev
    c, p, inputs, tokens, steps = load_case(name)
    searcher = c["loop"] == "searcher"
    cfg = CaptionConfig(B=c["B"], W=c["W"], L=c["L"], F=DIMS["D"] if searcher else DIMS["F"],
                        D=DIMS["D"], E=DIMS["E"], H=DIMS["H"], U=DIMS["H"], V=DIMS["V"], T=c["T"],
                        start_id=START, eos_id=EOS, length_penalty_weight=c["length_penalty_weight"],
                        tile_mode=c["tile_mode"], cell=c["cell"])
    if searcher:
        p = dict(p, enc_kernel=np.zeros((DIMS["D"], DIMS["D"]), np.float32))
        eng = CaptionEngine(cfg, p, device=dev())
        seq_len = (inputs["sequences"] > PAD).sum(axis=1).astype(np.int32)
        tok, _ = eng.search_baseline(to_dev(inputs["features"]), seq_len=to_dev(seq_len), pad_token=PAD)
        got = np.concatenate([np.full((c["B"], 1), START, np.int32), np_of(tok)], axis=1)
        np.testing.assert_array_equal(got, tokens)
        eng.close()
        return
    eng = CaptionEngine(cfg, p, device=dev())
    tok, _ = eng.decode(to_dev(inputs["img"]))
    tr = {k: np_of(v) for k, v in eng.trace().items()}
    np.testing.assert_array_equal(np_of(tok), tokens)
    for t, ref in enumerate(steps):
        np.testing.assert_array_equal(tr["ids"][t], ref["ids"])
        np.testing.assert_array_equal(tr["parents"][t], ref["parents"])
        close(tr["scores"][t], ref["scores"], "scores step %d" % t)
        close(tr["log_probs"][t], ref["log_probs"], "log_probs step %d" % t)
    eng.close()
    torch.cuda.synchronize()
