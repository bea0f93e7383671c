"""Probe: several recurrent engines decoding concurrently on their own streams against one
engine on the whole batch (does the latency-bound part of one decode - LSTM, fc1, beam step,
launch gaps - hide under the other's vocab projection?).

    python tools/overlap_probe_lstm.py [precision [beam width]]
"""
import sys
import time

import torch

sys.path.insert(0, ".")
from image_caption_amd import synthetic  # noqa: E402
from image_caption_amd.engine import CaptionConfig, CaptionEngine  # noqa: E402

dev = torch.device("cuda", 0)
precision = sys.argv[1] if len(sys.argv) > 1 else "fp32"
W = int(sys.argv[2]) if len(sys.argv) > 2 else 3
params = synthetic.make_params(0, F=2048, D=256, E=256, H=512, V=10000)
params = {k: torch.from_numpy(v).to(dev) for k, v in params.items()}


def run(nb, B, reps=20, fused=None):
    engs = [CaptionEngine(CaptionConfig(B=B, W=W, L=64, F=2048, D=256, E=256, H=512, U=512, V=10000,
                                        T=20, precision=precision), params, device=dev)
            for _ in range(nb)]
    if fused is not None:
        for e in engs:
            e.set_fused_vocab(fused)
    streams = [torch.cuda.Stream() for _ in range(nb)]
    imgs = [torch.from_numpy(synthetic.make_images(i, B, 64, 2048)).to(dev) for i in range(nb)]
    outs = [(torch.empty(B, 20, dtype=torch.int32, device=dev), torch.empty(B, device=dev)) for _ in range(nb)]

    def once():
        for e, s, im, (o, sc) in zip(engs, streams, imgs, outs):
            with torch.cuda.stream(s):
                e.decode(im, o, sc)

    for _ in range(3):
        once()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        once()
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / reps
    print("%d engine(s) x B=%d fused=%s: %.2f ms per round, %.0f captions/s" % (
        nb, B, fused, el * 1e3, nb * B / el), flush=True)
    for e in engs:
        e.close()


run(1, 512)
run(2, 256)
run(2, 256, fused=True)
run(4, 128)
run(2, 512)
run(3, 512)
run(1, 1024)
run(2, 1024)
