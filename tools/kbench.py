#!/usr/bin/env python3
"""Per-kernel micro-benchmarks at the headline shapes (B=512, W=3, V=10000, ...).
Times each libicap operator alone with HIP events on the current stream."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_caption_amd import ops  # noqa: E402


def timeit(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(iters):
        s.record()
        fn()
        e.record()
        e.synchronize()
        ts.append(s.elapsed_time(e) * 1e3)
    ts = np.array(ts)
    return float(np.median(ts)), float(ts.min())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=512)
    ap.add_argument("--W", type=int, default=3)
    ap.add_argument("--V", type=int, default=10000)
    ap.add_argument("--L", type=int, default=64)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    B, W, V, L = a.B, a.W, a.V, a.L
    N, H, U, D, E, F = B * W, 512, 512, 256, 256, 2048
    g = torch.Generator(device="cpu").manual_seed(0)
    rn = lambda *s: torch.randn(*s, generator=g).to(dev)
    res = {}

    def run(name, fn, flops=None, bytes_=None):
        if a.only and a.only not in name:
            return
        med, mn = timeit(fn)
        msg = "%-28s median %8.1f us  min %8.1f us" % (name, med, mn)
        if flops:
            msg += "  %7.1f TFLOP/s" % (flops / med / 1e6)
        if bytes_:
            msg += "  %7.1f GB/s" % (bytes_ / med / 1e3)
        print(msg, flush=True)
        res[name] = med

    # dense shapes
    for name, (M, K, Nn, act) in {
            "dense encode": (B * L, F, D, 1), "dense feat_proj": (B * L, D, U, 0),
            "dense attn_query": (N, H, U, 0), "dense fc1": (N, H, H, 0),
            "dense vocab": (N, H, V, 0)}.items():
        x, w, b = rn(M, K), rn(K, Nn) * 0.05, rn(Nn)
        out = torch.empty(M, Nn, device=dev)
        run(name, lambda: ops.dense(x, w, b, act=act, out=out), flops=2.0 * M * K * Nn)

    # vocab projection with the fused top-k epilogue (no logits written)
    x, w, b = rn(N, H), rn(H, V) * 0.05, rn(V)
    run("topk vocab", lambda: ops.dense_topk(x, w, b, 4, 1 if W == 1 else (3 if W <= 3 else 5)),
        flops=2.0 * N * H * V)

    # the same contractions with bf16 operands
    for name, (M, K, Nn, act) in {
            "bf16 encode": (B * L, F, D, 1), "bf16 fc1": (N, H, H, 0), "bf16 vocab": (N, H, V, 0)}.items():
        x, w, b = rn(M, K), rn(K, Nn) * 0.05, rn(Nn)
        pw = ops.pack_weight(w)
        run(name, lambda: ops.dense_bf16(x, pw, b, act=act), flops=2.0 * M * K * Nn,
            bytes_=K * Nn * 2.0 + M * K * 4.0 + M * Nn * 4.0)

    # split mode (fp32-accurate products on the bf16 pipe)
    for name, (M, K, Nn, act) in {
            "split encode": (B * L, F, D, 1), "split fc1": (N, H, H, 0), "split vocab": (N, H, V, 0)}.items():
        x, w, b = rn(M, K), rn(K, Nn) * 0.05, rn(Nn)
        pw = ops.pack_weight_split(w)
        run(name, lambda: ops.dense_split(x, pw, b, act=act), flops=2.0 * M * K * Nn)
    x, w, b = rn(N, H), rn(H, V) * 0.05, rn(V)
    pw = ops.pack_weight_split(w)
    run("split topk vocab", lambda: ops.dense_topk_split(x, pw, b, 4, 1 if W == 1 else (3 if W <= 3 else 5)),
        flops=2.0 * N * H * V)

    # lstm
    ctx, h, c = rn(N, D), rn(N, H) * 0.3, rn(N, H) * 0.3
    kern, rec, bias = rn(D + E, 4 * H) * 0.05, rn(H, 4 * H) * 0.05, rn(4 * H) * 0.1
    table = rn(V, E) * 0.05
    ids = torch.randint(0, V, (N,), dtype=torch.int32, generator=g).to(dev)
    run("lstm_step", lambda: ops.lstm_step(ctx, h, c, kern, rec, bias, table=table, ids=ids),
        flops=2.0 * N * (D + E + H) * 4 * H)

    emb_proj = rn(V, 4 * H) * 0.1
    run("lstm_step emb_proj", lambda: ops.lstm_step(ctx, h, c, kern, rec, bias, ids=ids, emb_proj=emb_proj),
        flops=2.0 * N * (D + H) * 4 * H)
    pk = ops.PackedWeight(D + H, 4 * H, dev, gate_units=H).fill(kern[:D].contiguous(), 0).fill(rec, D)
    run("bf16 lstm_step", lambda: ops.lstm_step_bf16(ctx, h, c, pk, bias, emb_proj, ids),
        flops=2.0 * N * (D + H) * 4 * H)

    pks = ops.PackedSplitWeight(D + H, 4 * H, dev, gate_units=H).fill(kern[:D].contiguous(), 0).fill(rec, D)
    run("split lstm_step", lambda: ops.lstm_step_split(ctx, h, c, pks, bias, emb_proj, ids),
        flops=2.0 * N * (D + H) * 4 * H)

    # attention
    pf, feats, q, v = rn(B, L, U), rn(B, L, D).abs(), rn(N, U), rn(U) * 0.1
    gate_w = rn(H) * 0.05
    att_bytes = B * L * (U + D) * 4.0 + N * (U + D + L) * 4.0
    run("attention repeat", lambda: ops.attention(pf, feats, q, v, 0.1, gate_state=c, gate_w=gate_w,
                                                  rows_per_group=W, row_stride_g=W, row_stride_k=1,
                                                  want_weights=False), bytes_=att_bytes)
    run("attention reference", lambda: ops.attention(pf, feats, q, v, 0.1, gate_state=c,
                                                     gate_w=gate_w, rows_per_group=W,
                                                     row_stride_g=1, row_stride_k=B,
                                                     want_weights=False), bytes_=att_bytes)

    # beam step
    logits = rn(B, W, V) * 2
    lp = -rn(B, W).abs()
    fin0 = torch.zeros(B, W, dtype=torch.uint8, device=dev)
    fin1 = torch.ones(B, W, dtype=torch.uint8, device=dev)
    ln = torch.full((B, W), 3, dtype=torch.int32, device=dev)
    nb = N * V * 4.0
    run("beam t>0", lambda: ops.beam_step(logits, lp, fin0, ln, 3, 4, 1.0), bytes_=nb)
    run("beam t=0", lambda: ops.beam_step(logits, lp, fin0, ln, 0, 4, 1.0), bytes_=nb / W)
    run("beam all finished", lambda: ops.beam_step(logits, lp, fin1, ln, 3, 4, 1.0))
    run("beam logprob input", lambda: ops.beam_step(logits, lp, fin0, ln, 3, 4, 1.0,
                                                    input_is_log_prob=True), bytes_=nb)
    # transformer decoder kernels at the config-4 shapes (1024 images x beam 3, 512 heads x 8)
    Bt, Nt, Hd, heads, kd, pos = 1024, 3072, 512, 512, 8, 10
    HD = heads * kd
    if not a.only or any(k in a.only for k in ("mha", "layernorm", "gru")):
        qt = rn(Nt, HD)
        kv_self = torch.randn(pos + 1, Nt, 2 * HD, generator=g).to(dev)
        run("mha_self_step pos=10", lambda: ops.mha_self_step(qt, kv_self, heads, kd, pos, 0),
            bytes_=(pos + 1) * Nt * 2 * HD * 4.0 + 2 * Nt * HD * 4.0)
        kv_cross = torch.randn(Bt, L, 2 * HD, generator=g).to(dev)
        run("mha_cross_step", lambda: ops.mha_cross_step(qt, kv_cross, Bt, 3, 1, Bt, heads, kd),
            bytes_=Bt * L * 2 * HD * 4.0 + 2 * Nt * HD * 4.0)
        xt, ot, gam, bet = rn(Nt, Hd), rn(Nt, Hd), rn(Hd), rn(Hd)
        run("residual_layernorm", lambda: ops.residual_layernorm(xt, ot, gam, bet),
            bytes_=4 * Nt * Hd * 4.0)
        # GRU step at the headline shape
        k3 = rn(D + E, 3 * H) / np.sqrt(D + E)
        r3 = rn(H, 3 * H) / np.sqrt(H)
        b3 = torch.zeros(2, 3 * H, device=dev)
        k4, r4, b4 = ops.gru_pack(k3, r3, b3)
        table = rn(V, E) * 0.05
        ids = torch.randint(0, V, (N,), generator=g).to(torch.int32).to(dev)
        ep = ops.dense(table, k4[D:].contiguous(), None)
        ctxg, hg = rn(N, D), rn(N, H) * 0.5
        run("gru_step emb_proj", lambda: ops.gru_step(ctxg, hg, k4, r4, b4, ids=ids, emb_proj=ep),
            flops=2.0 * N * (D + H) * 4 * H)
    return res


if __name__ == "__main__":
    main()
