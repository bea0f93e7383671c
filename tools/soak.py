"""Soak test of the decode schedules: many decodes at several shapes (merged launch, wide
merged launch, bf16 side stream), every result compared with the first one of its shape and
no -1 token (the error word of the merged launch's bounded waits) in any of them."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from image_caption_amd import synthetic  # noqa: E402
from image_caption_amd.engine import CaptionConfig, CaptionEngine  # noqa: E402
dev = torch.device("cuda", 0)
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
shapes = [dict(B=512, W=3), dict(B=96, W=5), dict(B=64, W=10), dict(B=512, W=5, precision="bf16"),
          dict(B=1000, W=1)]
p = {k: torch.from_numpy(v).to(dev) for k, v in synthetic.make_params(0, 2048, 256, 256, 512, 10000).items()}
for sh in shapes:
    eng = CaptionEngine(CaptionConfig(**sh), p, device=dev)
    img = torch.from_numpy(synthetic.make_images(1, sh["B"], 64, 2048)).to(dev)
    ref = eng.decode(img)[0].cpu().numpy().copy()
    assert (ref >= 0).all()
    t0, n = time.perf_counter(), 0
    while time.perf_counter() - t0 < secs:
        for _ in range(20):
            tok, _ = eng.decode(img)
        got = tok.cpu().numpy()
        n += 20
        assert np.array_equal(got, ref), "decode %d of %s differs from the first" % (n, sh)
    print("%s merged=%s fused=%s: %d decodes identical" % (sh, eng.merged_launch, eng.fused_vocab, n), flush=True)
    eng.close()
