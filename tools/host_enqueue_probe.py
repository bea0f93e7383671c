#!/usr/bin/env python3
"""How long the host takes to enqueue one decode against how long the GPU takes to run it
(headline shape): if the two are close, the decode loop is bound by the launch rate."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_caption_amd import synthetic  # noqa: E402
from image_caption_amd.engine import CaptionConfig, CaptionEngine  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    B, W, L, T = 512, 3, 64, 20
    dims = dict(F=2048, D=256, E=256, H=512, V=10000)
    p = synthetic.make_params(0, **dims)
    cfg = CaptionConfig(B=B, W=W, L=L, T=T, U=dims["H"], **dims)
    eng = CaptionEngine(cfg, {k: torch.as_tensor(v, device=dev) for k, v in p.items()}, device=dev)
    img = torch.randn(B, L, dims["F"], device=dev)
    for _ in range(5):
        eng.decode(img)
    torch.cuda.synchronize()
    n = 20
    t0 = time.perf_counter()
    for _ in range(n):
        eng.decode(img)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("host enqueue %.3f ms per decode, total %.3f ms per decode" % ((t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
    # one decode at a time (the queue is empty at every start)
    t0 = time.perf_counter()
    for _ in range(n):
        eng.decode(img)
        torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("synchronised after every decode: %.3f ms per decode" % ((t2 - t0) / n * 1e3))


if __name__ == "__main__":
    main()
