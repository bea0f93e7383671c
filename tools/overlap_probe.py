"""Probe: does running two half-batch transformer engines on two streams beat one full-batch
engine?  (Measures how much of the HBM-bound attention time hides under the MFMA-bound
projections when the GPU has both kinds of work in flight.)"""
import sys
import time

import torch

sys.path.insert(0, ".")
from image_caption_amd import synthetic  # noqa: E402
from image_caption_amd.tengine import TransformerConfig, TransformerEngine  # noqa: E402

dev = torch.device("cuda", 0)
dims = dict(L=64, F=2048, D=512, hidden=512, heads=512, key_dim=8, filter=2048, layers=6, V=10000, T=20)
p = synthetic.make_transformer_params(0, dims["F"], dims["D"], dims["V"], dims["hidden"], dims["layers"],
                                      dims["heads"], dims["key_dim"], dims["filter"])
p = {k: torch.from_numpy(v).to(dev) for k, v in p.items()}


def run(nb, B, reps=3):
    engs = [TransformerEngine(TransformerConfig(B=B, W=3, tile_mode="repeat", **dims), p, device=dev)
            for _ in range(nb)]
    streams = [torch.cuda.Stream() for _ in range(nb)]
    imgs = [torch.from_numpy(synthetic.make_images(i, B, dims["L"], dims["F"])).to(dev) for i in range(nb)]
    for e, s, im in zip(engs, streams, imgs):
        with torch.cuda.stream(s):
            e.decode(im)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        for e, s, im in zip(engs, streams, imgs):
            with torch.cuda.stream(s):
                e.decode(im)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / reps
    print("%d engine(s) x B=%d: %.1f ms per round, %.0f captions/s" % (nb, B, el * 1e3, nb * B / el), flush=True)
    for e in engs:
        e.close()


run(1, 1024)
run(2, 512)
run(4, 256)
