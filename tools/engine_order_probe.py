"""Un-instrumented captions/s of engines created one after another in one process.

    [PROBE_B=512 PROBE_W=3] python tools/engine_order_probe.py fp32 fp32_split bf16 ...
"""
import os, sys, time, torch
sys.path.insert(0, ".")
from image_caption_amd import synthetic
from image_caption_amd.engine import CaptionConfig, CaptionEngine
dev = torch.device("cuda:0")
p = synthetic.make_params(0, 2048, 256, 256, 512, 10000, 512)
B = int(os.environ.get("PROBE_B", "512")); W = int(os.environ.get("PROBE_W", "3"))
img = torch.from_numpy(synthetic.make_images(1, B, 64, 2048)).to(dev)
tok = torch.empty(B, 20, dtype=torch.int32, device=dev); sc = torch.empty(B, device=dev)
def run(mode, keep=False):
    e = CaptionEngine(CaptionConfig(B=B, W=W, L=64, F=2048, D=256, E=256, H=512, U=512, V=10000, T=20, precision=mode), p, device=dev)
    for _ in range(10): e.decode(img, tok, sc)
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(20): e.decode(img, tok, sc)
    torch.cuda.synchronize(); r = 20*B/(time.perf_counter()-t)
    print(mode, round(r), flush=True)
    if not keep: e.close()
    return e
for m in sys.argv[1:]:
    run(m)
