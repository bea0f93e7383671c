"""LSTM step time against the number of rows (run with ICAP_LSTM_FORM=wide|narrow|unset)."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from image_caption_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
C, E, H, V = 256, 256, 512, 10000
g = torch.Generator().manual_seed(0)
rn = lambda *s: torch.randn(*s, generator=g).to(dev)
kern, rec, bias = rn(C + E, 4 * H) / 22, rn(H, 4 * H) / 22, rn(4 * H) * 0.1
table = rn(V, E) * 0.05
ep = ops.dense(table, kern[C:].contiguous(), None)
for N in (32, 96, 128, 192, 384, 768, 1152, 1536, 2560):
    ctx, h, c = rn(N, C), rn(N, H) * 0.5, rn(N, H) * 0.5
    ids = torch.randint(0, V, (N,), generator=g).to(torch.int32).to(dev)
    fn = lambda: ops.lstm_step(ctx, h, c, kern, rec, bias, ids=ids, emb_proj=ep)
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print("N=%5d  %6.1f us" % (N, e0.elapsed_time(e1) * 10), flush=True)
