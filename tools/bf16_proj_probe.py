#!/usr/bin/env python3
"""bf16 beam-5 decode at B = 512: the vocab projection's time alone (timing brackets serialise
the kernels) and the decode rate, for the environment it is started under (ICAP_BF16_RESIDENT,
experiment builds through ICAP_LIB)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_caption_amd import _lib  # noqa: E402

if os.environ.get("ICAP_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["ICAP_LIB"])
from image_caption_amd import synthetic  # noqa: E402
from image_caption_amd.engine import CaptionConfig, CaptionEngine  # noqa: E402

B, W = int(os.environ.get("PROBE_B", 512)), int(os.environ.get("PROBE_W", 5))
dev = torch.device("cuda", 0)
p = {k: torch.from_numpy(v).to(dev) for k, v in synthetic.make_params(0, 2048, 256, 256, 512, 10000).items()}
eng = CaptionEngine(CaptionConfig(B=B, W=W, precision="bf16"), p, device=dev)
img = torch.from_numpy(synthetic.make_images(1, B, 64, 2048)).to(dev)
for _ in range(10):
    eng.decode(img)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    eng.decode(img)
torch.cuda.synchronize()
rate = B * 50 / (time.perf_counter() - t0)
eng.set_timing(True)
for _ in range(3):
    eng.decode(img)
torch.cuda.synchronize()
t = eng.kernel_times()
print("RESIDENT=%s %s: %.0f captions/s; alone per launch: %s" % (
    os.environ.get("ICAP_BF16_RESIDENT", "-"), os.environ.get("ICAP_LIB", "libicap.so"), rate,
    ", ".join("%s %.1f us" % (k, 1e3 * ms / max(n, 1)) for k, (ms, n) in t.items() if n and k in ("vocab", "lstm", "fc1", "beam", "attention", "attn_query"))), flush=True)
eng.close()
