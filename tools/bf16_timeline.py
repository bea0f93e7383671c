#!/usr/bin/env python3
"""Device timeline of gemm_bf16_kernel's workgroups (experiment build with -DICAP_TIMELINE,
see tools/build_timeline.sh): when every workgroup started, finished its first stage, its K
loop and its epilogue, on the 100 MHz device clock.

    ICAP_LIB=image_caption_amd/libicap_tl.so python tools/bf16_timeline.py lstm|fc1|dense_vocab
    (built with ICAP_TIMELINE=2: ... bf16_timeline.py decode [overlap] - the last vocab
    projection with its record epilogue inside a bf16 decode at B=512, beam 5)
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

which = sys.argv[1] if len(sys.argv) > 1 else "lstm"
dev = torch.device("cuda", 0)
B, W, V, H, D = 512, 5, 10000, 512, 256
N = B * W
g = torch.Generator(device="cpu").manual_seed(0)
rn = lambda *s: torch.randn(*s, generator=g).to(dev)
if which == "decode":
    from image_caption_amd import synthetic
    from image_caption_amd.engine import CaptionConfig, CaptionEngine
    params = synthetic.make_params(0, F=2048, D=256, E=256, H=512, V=V)
    cfg = CaptionConfig(B=B, W=W, L=64, F=2048, D=256, E=256, H=512, U=512, V=V, T=20, precision="bf16")
    eng = CaptionEngine(cfg, params, device=dev)
    eng.set_overlap(len(sys.argv) > 2 and sys.argv[2] == "overlap")
    img = torch.from_numpy(synthetic.make_images(1, B, 64, 2048)).to(dev)
    fn = lambda: eng.decode(img)
    n_wg = ((N + 127) // 128) * ((V + 127) // 128)
elif which == "lstm":
    ctx, h, c = rn(N, D), rn(N, H) * 0.3, rn(N, H) * 0.3
    kern, rec, bias = rn(D, 4 * H) * 0.05, rn(H, 4 * H) * 0.05, rn(4 * H) * 0.1
    emb_proj = rn(V, 4 * H) * 0.1
    ids = torch.randint(0, V, (N,), dtype=torch.int32, generator=g).to(dev)
    pk = ops.PackedWeight(D + H, 4 * H, dev, gate_units=H).fill(kern, 0).fill(rec, D)
    fn = lambda: ops.lstm_step_bf16(ctx, h, c, pk, bias, emb_proj, ids)
    n_wg = ((N + 63) // 64) * (4 * H // 128)
    if n_wg > 768:
        n_wg = ((N + 127) // 128) * (4 * H // 128)
else:
    Nn = H if which == "fc1" else V
    x, w, b = rn(N, H), rn(H, Nn) * 0.05, rn(Nn)
    pw = ops.pack_weight(w)
    fn = lambda: ops.dense_bf16(x, pw, b)
    bm = 128 if ((N + 127) // 128) * ((Nn + 127) // 128) >= 256 else 64
    n_wg = ((N + bm - 1) // bm) * ((Nn + 127) // 128)
for _ in range(10):
    fn()
torch.cuda.synchronize()
lib = _lib.load()
out = np.zeros((min(n_wg, 8192), 4), dtype=np.int64)
rc = lib.icap_debug_timeline(out.ctypes.data_as(C.c_void_p), out.shape[0])
assert rc == 0, rc
t = (out - out[:, 0].min()) / 100.0  # us
print("%s: %d workgroups; kernel span %.1f us" % (which, n_wg, t[:, 3].max()))
for name, col in (("start", 0), ("first stage in LDS", 1), ("K loop done", 2), ("end", 3)):
    q = np.percentile(t[:, col], [0, 10, 50, 90, 100])
    print("  %-20s min %6.1f  p10 %6.1f  p50 %6.1f  p90 %6.1f  max %6.1f" % ((name,) + tuple(q)))
for name, a, b_ in (("prologue", 0, 1), ("K loop", 1, 2), ("epilogue", 2, 3), ("whole", 0, 3)):
    d = t[:, b_] - t[:, a]
    q = np.percentile(d, [10, 50, 90])
    print("  per workgroup %-10s p10 %6.2f  p50 %6.2f  p90 %6.2f us" % ((name,) + tuple(q)))
