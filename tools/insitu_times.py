"""Per-class kernel times measured in situ (side-stream attention running) against the same
classes on one stream: shows what the overlap costs each kernel on the critical path.

    python tools/insitu_times.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from image_caption_amd import synthetic  # noqa: E402
from image_caption_amd.engine import CaptionConfig, CaptionEngine  # noqa: E402

dev = torch.device("cuda:0")
params = synthetic.make_params(0, F=2048, D=256, E=256, H=512, V=10000)
cfg = CaptionConfig(B=512, W=3, L=64, F=2048, D=256, E=256, H=512, U=512, V=10000, T=20)
eng = CaptionEngine(cfg, params, device=dev)
img = torch.from_numpy(synthetic.make_images(1, 512, 64, 2048)).to(dev)
main = ["lstm", "fc1", "vocab", "beam"]
for overlap in (True, False):
    eng.set_overlap(overlap)
    for classes in ([c] for c in main):
        eng.set_timing(classes)
        for _ in range(30):
            eng.decode(img)
        torch.cuda.synchronize()
        eng.decode(img)
        torch.cuda.synchronize()
        times = eng.kernel_times()
        ms, n = times[classes[0]]
        print("overlap=%-5s %-6s %7.1f us per launch (%d launches)" % (
            overlap, classes[0], 1e3 * ms / max(n, 1), n), flush=True)
eng.close()
