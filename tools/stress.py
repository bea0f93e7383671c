"""Engine life-cycle stress: repeated create / decode / destroy must not leak device memory,
and repeated decodes are deterministic."""
import sys

import torch

sys.path.insert(0, ".")
from image_caption_amd import synthetic  # noqa: E402
from image_caption_amd.engine import CaptionConfig, CaptionEngine  # noqa: E402

dev = torch.device("cuda", 0)
p = {k: torch.from_numpy(v).to(dev) for k, v in synthetic.make_params(0, 2048, 256, 256, 512, 10000).items()}
img = torch.from_numpy(synthetic.make_images(1, 64, 64, 2048)).to(dev)
ref = None
for rnd in range(4):
    for i in range(100):
        eng = CaptionEngine(CaptionConfig(B=64, W=3), p, device=dev)
        tok, _ = eng.decode(img)
        eng.close()
    torch.cuda.synchronize()
    t = tok.cpu().numpy()
    ref = t if ref is None else ref
    assert (t == ref).all()
    print("after %d engines: free %.1f MB" % (100 * (rnd + 1), torch.cuda.mem_get_info()[0] / 1e6), flush=True)
