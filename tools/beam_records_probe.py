#!/usr/bin/env python3
"""Times the records beam step alone: behind an idle spin (records warm in the caches) and
behind the projection that has just written them (what the decode loop sees).

    python tools/beam_records_probe.py [--B 512 --W 5]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_caption_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=512)
ap.add_argument("--W", type=int, default=5)
ap.add_argument("--V", type=int, default=10000)
a = ap.parse_args()
dev = torch.device("cuda", 0)
B, W, V, H = a.B, a.W, a.V, 512
N = B * W
g = torch.Generator(device="cpu").manual_seed(0)
rn = lambda *s: torch.randn(*s, generator=g).to(dev)
x, w, b = rn(N, H), rn(H, V) * 0.05, rn(V)
top_k = 1 if W == 1 else (3 if W <= 3 else 5)
lp = -rn(B, W).abs()
fin = torch.zeros(B, W, dtype=torch.uint8, device=dev)
ln = torch.full((B, W), 3, dtype=torch.int32, device=dev)
rec, eos_x = ops.dense_topk(x, w, b, 4, top_k)
print("records: %.1f MB" % (rec.numel() * 4 / 1e6))


def run(behind_projection, iters=40):
    ts = []
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for i in range(iters + 5):
        global rec, eos_x
        if behind_projection:
            rec, eos_x = ops.dense_topk(x, w, b, 4, top_k)
        else:
            torch.cuda._sleep(400000)
        s.record()
        ops.beam_step_from_records(rec, eos_x, lp, fin, ln, 3, 4, 1.0, V)
        e.record()
        e.synchronize()
        if i >= 5:
            ts.append(s.elapsed_time(e) * 1e3)
    ts = np.array(ts)
    return float(np.median(ts)), float(ts.min())


for bp in (False, True):
    med, mn = run(bp)
    print("behind %-10s median %6.1f us  min %6.1f us" % ("projection" if bp else "spin", med, mn), flush=True)
