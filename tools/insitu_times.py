"""Per-class kernel times measured in situ (side-stream attention running) against the same
classes on one stream: shows what the overlap costs each kernel on the critical path.

    python tools/insitu_times.py [--B 512 --W 3 --precision fp32 --classes lstm,fc1,vocab,beam]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from image_caption_amd import synthetic  # noqa: E402
from image_caption_amd.engine import CaptionConfig, CaptionEngine  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=512)
ap.add_argument("--W", type=int, default=3)
ap.add_argument("--precision", default="fp32")
ap.add_argument("--classes", default="lstm,fc1,vocab,beam")
a = ap.parse_args()
dev = torch.device("cuda:0")
params = synthetic.make_params(0, F=2048, D=256, E=256, H=512, V=10000)
cfg = CaptionConfig(B=a.B, W=a.W, L=64, F=2048, D=256, E=256, H=512, U=512, V=10000, T=20,
                    precision=a.precision)
eng = CaptionEngine(cfg, params, device=dev)
img = torch.from_numpy(synthetic.make_images(1, a.B, 64, 2048)).to(dev)
main = a.classes.split(",")
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
