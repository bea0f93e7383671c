#!/usr/bin/env python3
"""Phase timeline of the one-launch greedy loop (experiment build, tools/build_timeline.sh):
what workgroup 0 spends in every phase and at every grid barrier.

    ICAP_LIB=image_caption_amd/libicap_tl.so python tools/small_loop_timeline.py [B]
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_caption_amd import _lib  # noqa: E402

if os.environ.get("ICAP_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["ICAP_LIB"])
from image_caption_amd import synthetic  # noqa: E402
from image_caption_amd.engine import CaptionConfig, CaptionEngine  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T = 20
dev = torch.device("cuda", 0)
params = synthetic.make_params(0, F=2048, D=256, E=256, H=512, V=10000)
eng = CaptionEngine(CaptionConfig(B=B, W=1, T=T), params, device=dev)
img = torch.from_numpy(synthetic.make_images(1, B, 64, 2048)).to(dev)
for _ in range(5):
    eng.decode(img)
torch.cuda.synchronize()
lib = _lib.load()
n = 8 + 6 * T
out = np.zeros(n, dtype=np.int64)
lib.icap_debug_small_loop_stamps.argtypes = [C.c_void_p, C.c_int]
assert lib.icap_debug_small_loop_stamps(out.ctypes.data_as(C.c_void_p), n) == 0
t = (out - out[0]) / 100.0
print("prologue: query %.1f, barrier %.1f, attention %.1f, barrier %.1f us" % (
    t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3]))
s = t[8:].reshape(T, 6)
prev_end = np.concatenate([[t[4]], s[:-1, 5]])
ph = np.stack([s[:, 0] - prev_end, s[:, 1] - s[:, 0], s[:, 2] - s[:, 1], s[:, 3] - s[:, 2],
               s[:, 4] - s[:, 3], s[:, 5] - s[:, 4]], axis=1)
names = ("cell", "barrier", "query+vocab", "barrier", "attention+beam", "barrier")
print("per step (median over %d steps, workgroup 0):" % T)
for k, nm in enumerate(names):
    print("  %-16s %6.1f us  (min %.1f, max %.1f)" % (nm, np.median(ph[:, k]), ph[:, k].min(), ph[:, k].max()))
print("  step total       %6.1f us; loop %.1f us" % (np.median(ph.sum(axis=1)), s[-1, 5] - t[0]))
