"""Config 5 (B=256, beam 10, 196 locations, 50k vocabulary): per-class kernel times inside
the decode (side stream active) and the decode time, e.g. under ICAP_FOLD_FC=0|1."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_caption_amd import synthetic  # noqa: E402
from image_caption_amd.engine import CaptionConfig, CaptionEngine  # noqa: E402
dev = torch.device("cuda:0")
d = dict(B=256, W=10, L=196, F=512, D=256, E=256, H=512, U=512, V=50000, T=20)
params = synthetic.make_params(0, F=d["F"], D=d["D"], E=d["E"], H=d["H"], V=d["V"])
eng = CaptionEngine(CaptionConfig(precision="fp32", **d), params, device=dev)
img = torch.from_numpy(synthetic.make_images(1, d["B"], d["L"], d["F"])).to(dev)
for _ in range(3): eng.decode(img)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): eng.decode(img)
torch.cuda.synchronize()
print("decode %.2f ms (%.0f us per step)" % ((time.perf_counter() - t0) / 5 * 1e3, (time.perf_counter() - t0) / 5 / 20 * 1e6))
for cls in ("lstm", "fc1", "vocab", "beam", "attn_query", "attention"):
    eng.set_timing([cls])
    for _ in range(3): eng.decode(img)
    torch.cuda.synchronize()
    eng.decode(img); torch.cuda.synchronize()
    ms, n = eng.kernel_times()[cls]
    print("  %-10s %8.1f us per launch (%d launches)" % (cls, 1e3 * ms / max(n, 1), n), flush=True)
eng.close()
