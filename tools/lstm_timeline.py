#!/usr/bin/env python3
"""Device timeline of the exact-f32 LSTM step (lstm_f32_wide_kernel) at the headline shape
(experiment build, tools/build_timeline.sh): per workgroup the time to the first staged tile,
through the K loop and through the cell update, on the 100 MHz device clock.

    ICAP_LIB=image_caption_amd/libicap_tl.so python tools/lstm_timeline.py [B W]
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
from image_caption_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
W = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda", 0)
N, H, D, E, V = B * W, 512, 256, 256, 10000
g = torch.Generator(device="cpu").manual_seed(0)
rn = lambda *s: torch.randn(*s, generator=g).to(dev)
ctx, h, c = rn(N, D), rn(N, H) * 0.3, rn(N, H) * 0.3
kern, rec, bias = rn(D + E, 4 * H) * 0.05, rn(H, 4 * H) * 0.05, rn(4 * H) * 0.1
emb_proj = rn(V, 4 * H) * 0.1
ids = torch.randint(0, V, (N,), dtype=torch.int32, generator=g).to(dev)
fn = lambda: ops.lstm_step(ctx, h, c, kern, rec, bias, ids=ids, emb_proj=emb_proj)
for _ in range(10):
    fn()
torch.cuda.synchronize()
n = ((N + 95) // 96) * (H // 32)
out = np.zeros((n, 8), dtype=np.int64)
lib = _lib.load()
lib.icap_debug_timeline_f32.argtypes = [C.c_void_p, C.c_int]
assert lib.icap_debug_timeline_f32(out.ctypes.data_as(C.c_void_p), n) == 0
t = (out - out[:, 0].min()) / 100.0
print("lstm_f32_wide_kernel: %d workgroups; span %.1f us" % (n, t[:, 1].max()))
q = lambda a: tuple(np.percentile(a, [0, 10, 50, 90, 100]))
for name, col in (("start", 0), ("first tile staged", 2), ("K loop done", 3), ("end", 1)):
    print("  %-18s min %6.1f p10 %6.1f p50 %6.1f p90 %6.1f max %6.1f" % ((name,) + q(t[:, col])))
for name, a, b in (("prologue", 0, 2), ("K loop", 2, 3), ("cell update", 3, 1)):
    print("  per workgroup %-12s p10 %5.2f p50 %5.2f p90 %5.2f us" % ((name,) + tuple(np.percentile(t[:, b] - t[:, a], [10, 50, 90]))))
