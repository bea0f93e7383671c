"""Whole-decode HIP graph probe: captures one icap_decode call (main-stream launches plus the
side-stream attention, joined by events) into a graph and compares replay against direct
launches - captions/s and identical tokens.

    python tools/graph_probe.py [--configs beam3_b512 greedy_b32 ...]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from image_caption_amd import synthetic  # noqa: E402
from image_caption_amd.engine import CaptionConfig, CaptionEngine  # noqa: E402

CASES = {
    "beam3_b512": dict(B=512, W=3),
    "beam3_b128": dict(B=128, W=3),
    "beam3_b32": dict(B=32, W=3),
    "greedy_b32": dict(B=32, W=1),
    "greedy_b1": dict(B=1, W=1),
    "beam5_b512_bf16": dict(B=512, W=5, precision="bf16"),
}


def rate(fn, reps, B):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return B * reps / (time.perf_counter() - t0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", nargs="*", default=list(CASES))
    ap.add_argument("--reps", type=int, default=40)
    ap.add_argument("--no-overlap", action="store_true",
                    help="keep every launch on one stream (a linear graph)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    params = synthetic.make_params(0, F=2048, D=256, E=256, H=512, V=10000)
    for name in args.configs:
        kw = dict(CASES[name])
        B = kw["B"]
        cfg = CaptionConfig(L=64, F=2048, D=256, E=256, H=512, U=512, V=10000, T=20, **kw)
        eng = CaptionEngine(cfg, params, device=dev)
        if args.no_overlap:
            eng.set_overlap(False)
        img = torch.from_numpy(synthetic.make_images(1, B, 64, 2048)).to(dev)
        out = torch.empty(B, 20, dtype=torch.int32, device=dev)
        sc = torch.empty(B, dtype=torch.float32, device=dev)
        for _ in range(5):
            eng.decode(img, out, sc)
        torch.cuda.synchronize()
        ref = out.clone()
        direct = rate(lambda: eng.decode(img, out, sc), args.reps, B)
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            eng.decode(img, out, sc)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        out.zero_()
        with torch.cuda.graph(g):
            eng.decode(img, out, sc)
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        same = bool((out == ref).all())
        graph = rate(g.replay, args.reps, B)
        direct2 = rate(lambda: eng.decode(img, out, sc), args.reps, B)
        print("%-18s direct %9.1f / %9.1f  graph %9.1f captions/s  (x%.3f)  tokens %s" % (
            name, direct, direct2, graph, graph / max(direct, direct2), "same" if same else "DIFFER"),
            flush=True)
        del g
        eng.close()


if __name__ == "__main__":
    main()
