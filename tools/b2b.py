"""Back-to-back launch timing of the vocab projection kernels (no per-launch events)."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from image_caption_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
N, H, V = 1536, 512, 10000
rng = np.random.default_rng(0)
x = torch.from_numpy(rng.standard_normal((N, H)).astype(np.float32)).to(dev)
w = torch.from_numpy((rng.standard_normal((H, V)) / 22).astype(np.float32)).to(dev)
b = torch.zeros(V, device=dev)
out = torch.empty(N, V, device=dev)


def timeit(name, fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print("%-24s %7.1f us  %6.1f TF/s" % (name, us, 2.0 * N * H * V / us / 1e6), flush=True)


timeit("dense vocab (store)", lambda: ops.dense(x, w, b, out=out))
timeit("topk vocab", lambda: ops.dense_topk(x, w, b, 4, 3))
