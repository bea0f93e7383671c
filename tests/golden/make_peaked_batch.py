#!/usr/bin/env python3
"""Builds the image batch of the exact-token parity test at the headline size
(tests/golden/peaked_batch_seeds.json): 512 per-image seeds such that, on the peaked synthetic
weights (oracle.make_peaked_params(0)), EVERY top-k decision of the oracle's 20-step beam-3
decode - in both tile orders - has a relative score margin of at least TARGET between
neighbours among the best W+1 candidates.

Why: both the oracle and the HIP engine are fp32 with different summation orders (their
scores differ by up to a few 1e-6 relative on these weights), so a decision closer than that
can legitimately come out either way.  On a batch without such decisions the engine's tokens
and beam records must equal the oracle's bit for bit, with no tie rule at all; the test
re-checks the margins at run time, so the screening is verified, not assumed.

Images are drawn per seed (oracle.make_images(seed, 1, L, F)); an image that takes part in a
low-margin decision is re-drawn.  In the reference tile order beam k of image b reads the
features of image (b*W + k) % B, so all W of them are re-drawn.  Takes a few minutes of CPU.

Run from the repository root:  python tests/golden/make_peaked_batch.py"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import caption_oracle as co  # noqa: E402

DIMS = dict(F=2048, D=256, E=256, H=512, V=10000)
B, W, L, T = 512, 3, 64, 20
TARGET = 4e-5        # the test asserts 3e-5


def batch_from_seeds(seeds, L=L, F=DIMS["F"]):
    return np.concatenate([co.make_images(int(s), 1, L, F) for s in seeds], axis=0)


def image_margins(p, img, W, T, tile_mode):
    """Per image: the smallest relative gap between neighbours among its best W+1 candidates,
    over all steps of the free-running oracle."""
    per_image = np.full(img.shape[0], np.inf)

    def choose(scores_flat, k):
        v, i = co.choose_top_k(scores_flat, k + 1)
        s = v.astype(np.float64)
        ok = s > -1e30
        gaps = np.abs(np.diff(s, axis=1)) / np.maximum(np.abs(s[:, :-1]), 1e-30)
        gaps = np.where(ok[:, 1:] & ok[:, :-1], gaps, np.inf)
        np.minimum(per_image, gaps.min(axis=1), out=per_image)
        return v[:, :k], i[:, :k]

    co.predict_image_caption_sequence(p, img, W, T, 3, 4, 1.0, tile_mode, hoist_w1=True,
                                      choose_successors_fn=choose)
    return per_image


def main():
    p = co.make_peaked_params(0, **DIMS)
    seeds = np.arange(1000, 1000 + B)
    fresh = 1000 + B
    for it in range(12):
        img = batch_from_seeds(seeds)
        redo = set()
        report = {}
        for mode in ("reference", "repeat"):
            m = image_margins(p, img, W, T, mode)
            report[mode] = float(m.min())
            for b in np.nonzero(m < TARGET)[0]:
                redo.update(((b * W + k) % B for k in range(W)) if mode == "reference" else [b])
        print("iteration %d: min margins %s, %d images to re-draw" % (it, report, len(redo)), flush=True)
        if not redo:
            break
        for b in sorted(redo):
            seeds[b] = fresh
            fresh += 1
    else:
        raise SystemExit("did not converge")
    out = {"param_seed": 0, "dims": DIMS, "B": B, "W": W, "L": L, "T": T, "target_margin": TARGET,
           "min_margin": report, "image_seeds": [int(s) for s in seeds]}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "peaked_batch_seeds.json")
    with open(path, "w") as f:
        json.dump(out, f)
    print("wrote", path)


if __name__ == "__main__":
    main()
