import sys, time, torch
sys.path.insert(0, ".")
from image_caption_amd import synthetic
from image_caption_amd.engine import CaptionConfig, CaptionEngine
dev = torch.device("cuda", 0)
p = {k: torch.from_numpy(v).to(dev) for k, v in synthetic.make_params(0, 2048, 256, 256, 512, 10000).items()}
for B, W in ((512, 3), (512, 5), (256, 5), (1024, 3)):
    eng = CaptionEngine(CaptionConfig(B=B, W=W, precision="bf16"), p, device=dev)
    img = torch.from_numpy(synthetic.make_images(1, B, 64, 2048)).to(dev)
    for _ in range(5): eng.decode(img)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(40): eng.decode(img)
    torch.cuda.synchronize()
    print("B=%d W=%d bf16 %.0f captions/s" % (B, W, B * 40 / (time.perf_counter() - t0)), flush=True)
    eng.close()
