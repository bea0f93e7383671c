"""Decode throughput against batch size, fused and unfused vocab paths (fp32, beam 3)."""
import sys
import time

import torch

sys.path.insert(0, ".")
from image_caption_amd import synthetic  # noqa: E402
from image_caption_amd._lib import IcapError  # noqa: E402
from image_caption_amd.engine import CaptionConfig, CaptionEngine  # noqa: E402

dev = torch.device("cuda", 0)
p = {k: torch.from_numpy(v).to(dev) for k, v in synthetic.make_params(0, 2048, 256, 256, 512, 10000).items()}
W = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for B in (8, 32, 64, 96, 128, 192, 256, 384, 512):
    eng = CaptionEngine(CaptionConfig(B=B, W=W), p, device=dev)
    img = torch.from_numpy(synthetic.make_images(1, B, 64, 2048)).to(dev)
    default = eng.fused_vocab
    out = []
    for fused in (True, False):
        try:
            eng.set_fused_vocab(fused)
        except IcapError:
            out.append("   n/a")
            continue
        for _ in range(3):
            eng.decode(img)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 20
        for _ in range(reps):
            eng.decode(img)
        torch.cuda.synchronize()
        out.append("%7.0f" % (B * reps / (time.perf_counter() - t0)))
    print("B=%4d W=%d  fused %s  unfused %s captions/s  (default: %s)" % (B, W, out[0], out[1], "fused" if default else "unfused"), flush=True)
    eng.close()
