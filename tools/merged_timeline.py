#!/usr/bin/env python3
"""Device timeline of the merged projection + attention launch at the headline shape
(experiment build, tools/build_timeline.sh): start and end of its attention workgroups (the
first 256) and of its tiles.

    ICAP_LIB=image_caption_amd/libicap_tl.so python tools/merged_timeline.py [B W]
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

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
W = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda", 0)
params = synthetic.make_params(0, F=2048, D=256, E=256, H=512, V=10000)
eng = CaptionEngine(CaptionConfig(B=B, W=W, L=64, F=2048, D=256, E=256, H=512, U=512, V=10000, T=20), params, device=dev)
assert eng.merged_launch
img = torch.from_numpy(synthetic.make_images(1, B, 64, 2048)).to(dev)
for _ in range(5):
    eng.decode(img)
torch.cuda.synchronize()
att = min(256, B)
tiles = ((B * W + 127) // 128) * ((10000 + 127) // 128)
nq = 0 if os.environ.get("ICAP_MERGED_QTILES") == "0" else ((B * W + 63) // 64) * 4
nq = (nq + 7) // 8 * 8
n = nq + att + tiles
out = np.zeros((n, 8), dtype=np.int64)
lib = _lib.load()
lib.icap_debug_timeline_f32.argtypes = [C.c_void_p, C.c_int]
assert lib.icap_debug_timeline_f32(out.ctypes.data_as(C.c_void_p), n) == 0
t = (out - out[:, 0].min()) / 100.0
idx = np.arange(n)
if nq:
    # grid order of dense_topk_attq_kernel: query tiles, tiles_before vocab tiles, the
    # attention workgroups, the other vocab tiles
    tb = int(os.environ.get("ICAP_ATTQ_TILES_BEFORE", 512 + (512 - nq) // 2)) & ~7
    tb = min(tb, tiles & ~7)
    is_q = idx < nq
    is_att = (idx >= nq + tb) & (idx < nq + tb + att)
    print("  query tiles: end p50 %.1f max %.1f us; attention workgroups past their wait %.1f us (p50) after they start" % (
        np.median(t[is_q, 1]), t[is_q, 1].max(), np.median(t[is_att, 2] - t[is_att, 0])))
else:
    is_q = idx < 0
    is_att = idx < att
    print("  attention phases (us since the workgroup started, medians): query done %.1f, image 1 scores done %.1f, "
          "image 2 starts %.1f, end %.1f" % tuple(np.median(t[is_att, k] - t[is_att, 0]) for k in (2, 5, 4, 1)))
is_tile = ~is_q & ~is_att
print("launch span %.1f us; %d attention workgroups, %d tiles" % (t[:, 1].max(), att, tiles))
for name, sel in (("attention", is_att), ("tiles", is_tile)):
    s_, e_ = t[sel, 0], t[sel, 1]
    d = e_ - s_
    q = lambda a: tuple(np.percentile(a, [0, 10, 50, 90, 100]))
    print("  %-9s start  min %6.1f p10 %6.1f p50 %6.1f p90 %6.1f max %6.1f" % ((name,) + q(s_)))
    print("  %-9s end    min %6.1f p10 %6.1f p50 %6.1f p90 %6.1f max %6.1f" % ((name,) + q(e_)))
    print("  %-9s length min %6.1f p10 %6.1f p50 %6.1f p90 %6.1f max %6.1f" % ((name,) + q(d)))
