"""Captions/s of the headline shape (or --B --W --precision) through the library named by
ICAP_LIB (default: the in-tree libicap.so): for A/B runs of experiment builds."""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_caption_amd import _lib  # noqa: E402
if os.environ.get("ICAP_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["ICAP_LIB"])
from image_caption_amd import synthetic  # noqa: E402
from image_caption_amd.engine import CaptionConfig, CaptionEngine  # noqa: E402
ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=512)
ap.add_argument("--W", type=int, default=3)
ap.add_argument("--precision", default="fp32")
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
dev = torch.device("cuda", 0)
p = {k: torch.from_numpy(v).to(dev) for k, v in synthetic.make_params(0, 2048, 256, 256, 512, 10000).items()}
eng = CaptionEngine(CaptionConfig(B=a.B, W=a.W, precision=a.precision), p, device=dev)
img = torch.from_numpy(synthetic.make_images(1, a.B, 64, 2048)).to(dev)
for _ in range(20): eng.decode(img)
rates = []
for _ in range(a.reps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 100
    for _ in range(n): eng.decode(img)
    torch.cuda.synchronize()
    rates.append(a.B * n / (time.perf_counter() - t0))
print("%s B=%d W=%d %s merged=%s: %s captions/s" % (os.environ.get("ICAP_LIB", "libicap.so"), a.B, a.W, a.precision,
      eng.merged_launch, " ".join("%.0f" % r for r in rates)), flush=True)
eng.close()
