#!/usr/bin/env python3
"""Generates tests/golden/e2e_tiny_fixtures.json: whole-sequence decodes at tiny dimensions
(inputs, weights and every per-step beam record), produced by the CPU oracle.

The reference itself cannot run here (TensorFlow is not installed) and holds no fixture for
the decode loop (recurrent_example.py:477-562), so these vectors do not pin the oracle to
the reference; they pin the oracle AND the HIP engine to one committed set of numbers, so
that neither can drift unnoticed.  Every case stores its smallest relative top-k margin
(over all steps and images, between neighbours among the best W+1 candidates): the cases are
chosen with margins far above fp32 noise, so the integer outputs are exact on any machine.

Run from the repository root:  python tests/golden/make_e2e_fixtures.py"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import caption_oracle as co  # noqa: E402

TINY = dict(F=8, D=4, E=4, H=8, V=11)   # (the attention kernel wants H % 4 == 0)
CASES = [
    # name, cell, tile_mode, B, W, L, T, length_penalty_weight, loop
    # (B = 4: with B == W the reference tile order gives every image the same feature rows)
    ("lstm_reference_w3", "lstm", "reference", 4, 3, 5, 6, 1.0, "decode"),
    ("lstm_repeat_w3", "lstm", "repeat", 3, 3, 5, 6, 1.0, "decode"),
    ("lstm_reference_w1", "lstm", "reference", 3, 1, 5, 6, 1.0, "decode"),
    ("lstm_reference_w2_lp12", "lstm", "reference", 3, 2, 5, 6, 1.2, "decode"),
    ("gru_reference_w3", "gru", "reference", 4, 3, 5, 6, 1.0, "decode"),
    ("lstm_searcher_w3", "lstm", "reference", 4, 3, 5, 6, 1.0, "searcher"),
]
MIN_MARGIN = 1e-4


def tiny_params(seed, cell):
    """Keras-default weights at the tiny dimensions, scaled up so that the 11-word
    distributions are not flat and the end token (4) is reached in some captions."""
    p = co.perturb_biases(co.make_params(seed, **TINY), seed + 1, scale=0.3)
    f = np.float32
    p["embedding"] = p["embedding"] * f(20)
    p["fc2_kernel"] = p["fc2_kernel"] * f(6)
    p["lstm_kernel"] = p["lstm_kernel"] * f(2)
    for k in ("init0_kernel", "init1_kernel", "att_V_kernel", "att_W1_kernel"):
        p[k] = p[k] * f(4)
    if cell == "gru":
        p = co.add_gru_params(p, seed)
        p["gru_kernel"] = p["gru_kernel"] * f(2)
        p["gru_bias"] = (p["gru_bias"] + np.random.default_rng(seed + 2).normal(
            0, 0.3, p["gru_bias"].shape)).astype(np.float32)
    return p


def run_case(cell, tile_mode, B, W, L, T, lpw, loop, seed):
    p = tiny_params(seed, cell)
    # images with their own brightness and their own dead feature channels, so that the
    # captions depend on the image (iid ReLU-Gaussian maps all pool to the same vector)
    rng_i = np.random.default_rng(seed + 77)
    img = co.make_images(seed + 50, B, L, TINY["F"])
    img = img * rng_i.uniform(0.2, 4.0, size=(B, 1, 1)).astype(np.float32)
    img = (img * (rng_i.random((B, 1, TINY["F"])) < 0.6)).astype(np.float32)
    margins = []

    def choose(scores_flat, k):
        kk = min(k + 1, scores_flat.shape[-1])
        v, i = co.choose_top_k(scores_flat, kk)
        s = v.astype(np.float64)
        fin = np.isfinite(s) & (s > -1e30)
        gaps = np.abs(np.diff(s, axis=1)) / np.maximum(np.abs(s[:, :-1]), 1e-30)
        margins.append(np.where(fin[:, 1:] & fin[:, :-1], gaps, np.inf).min())
        return v[:, :k], i[:, :k]

    if loop == "decode":
        tokens, trace = co.predict_image_caption_sequence(
            p, img, W, T, 3, 4, lpw, tile_mode, return_trace=True, cell=cell,
            choose_successors_fn=choose)
        steps = [{"ids": st["out"].predicted_ids, "parents": st["out"].beam_parent_ids,
                  "scores": st["out"].scores, "log_probs": st["state"].log_probs,
                  "finished": st["state"].finished.astype(np.int32), "lengths": st["state"].lengths,
                  "logits_beam0_image0": st["logits"][0, 0]} for st in trace]
        inputs = {"img": img}
    else:
        feats = co.mlp_encoder(img, p["enc_kernel"], p["enc_bias"])
        rng = np.random.default_rng(seed + 9)
        seqs = np.zeros((B, T + 1), np.int32)
        for b in range(B):
            n = int(rng.integers(3, T + 2))
            seqs[b, :n] = rng.integers(5, TINY["V"], size=n)
        tokens = co.baseline_sequence_beam_search(p, feats, seqs, W, TINY["V"], 0, 3, 4, 0, cell=cell)
        steps = []
        inputs = {"features": feats, "sequences": seqs}
        margins.append(np.inf)
    return p, inputs, tokens, steps, float(min(margins))


def to_json(x):
    if isinstance(x, dict):
        return {k: to_json(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [to_json(v) for v in x]
    if isinstance(x, np.ndarray):
        if x.dtype.kind == "f":
            return {"dtype": "float32", "shape": list(x.shape),
                    "data": [float(v) for v in x.astype(np.float32).ravel()]}
        return {"dtype": "int32", "shape": list(x.shape), "data": [int(v) for v in x.ravel()]}
    return x


def main():
    out = {"dims": TINY, "start_id": 3, "eos_id": 4, "pad_id": 0, "cases": {}}
    for name, cell, tile_mode, B, W, L, T, lpw, loop in CASES:
        for seed in range(400):
            p, inputs, tokens, steps, margin = run_case(cell, tile_mode, B, W, L, T, lpw, loop, seed)
            # a useful vector: the end token is reached but not at once, the images decode to
            # different captions, several words occur
            varied = (bool((tokens == 4).any()) and not bool((tokens[:, 0] == 4).any()) and
                      len({tuple(r) for r in tokens.tolist()}) >= 2 and len(np.unique(tokens)) >= 4)
            if margin >= MIN_MARGIN and varied:
                break
        else:
            raise SystemExit("no seed with a clear margin for " + name)
        out["cases"][name] = to_json({
            "cell": cell, "tile_mode": tile_mode, "B": B, "W": W, "L": L, "T": T,
            "length_penalty_weight": lpw, "loop": loop, "seed": seed, "min_topk_margin": margin,
            "params": p, "inputs": inputs, "tokens": tokens, "steps": steps})
        print("%-26s seed %2d  min margin %.2e  tokens %s" % (name, seed, margin, tokens.tolist()))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "e2e_tiny_fixtures.json")
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
