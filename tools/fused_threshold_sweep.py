import sys, time, torch
sys.path.insert(0, ".")
from image_caption_amd import synthetic
from image_caption_amd.engine import CaptionConfig, CaptionEngine
dev = torch.device("cuda", 0)
p = {k: torch.from_numpy(v).to(dev) for k, v in synthetic.make_params(0, 2048, 256, 256, 512, 10000).items()}
import os
SHAPES = eval(os.environ.get("SHAPES", "((64, 3), (128, 3), (192, 3), (256, 3), (320, 3), (128, 5), (192, 5), (512, 1), (768, 1), (1024, 1))"))
PREC = os.environ.get("PREC", "fp32")
for B, W in SHAPES:
    for fused in (False, True):
        eng = CaptionEngine(CaptionConfig(B=B, W=W, precision=PREC), p, device=dev)
        try:
            eng.set_fused_vocab(fused)
        except Exception as ex:
            print("B=%d W=%d fused=%s not served" % (B, W, fused)); eng.close(); continue
        img = torch.from_numpy(synthetic.make_images(1, B, 64, 2048)).to(dev)
        for _ in range(3): eng.decode(img)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 20
        for _ in range(n): eng.decode(img)
        torch.cuda.synchronize()
        print("B=%d W=%d fused=%s merged=%s %.0f captions/s" % (B, W, eng.fused_vocab, eng.merged_launch, B * n / (time.perf_counter() - t0)), flush=True)
        eng.close()
