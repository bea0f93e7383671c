"""Decode throughput by shape (fp32): run once per schedule, e.g.
    ICAP_MERGED=0 python tools/merged_sweep.py; ICAP_MERGED=1 python tools/merged_sweep.py"""
import sys, time, torch
sys.path.insert(0, ".")
from image_caption_amd import synthetic
from image_caption_amd.engine import CaptionConfig, CaptionEngine
dev = torch.device("cuda", 0)
p = {k: torch.from_numpy(v).to(dev) for k, v in synthetic.make_params(0, 2048, 256, 256, 512, 10000).items()}
for B, W in ((256, 3), (384, 3), (512, 3), (768, 3), (1024, 3), (256, 5), (512, 5), (1024, 1), (2048, 1)):
    eng = CaptionEngine(CaptionConfig(B=B, W=W), p, device=dev)
    img = torch.from_numpy(synthetic.make_images(1, B, 64, 2048)).to(dev)
    for _ in range(3): eng.decode(img)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 20
    for _ in range(n): eng.decode(img)
    torch.cuda.synchronize()
    print("B=%d W=%d fused=%s %.0f captions/s" % (B, W, eng.fused_vocab, B * n / (time.perf_counter() - t0)), flush=True)
    eng.close()
