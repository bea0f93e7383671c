#!/usr/bin/env python3
"""Headline benchmark: captions/sec of the whole-sequence beam-search decode
(beam 3, 20 steps, B=512 images per GPU, fp32) on N MI355X GPUs of one node.

One pass of the hot path over one batch of 512 images: encoder -> tile -> reset_state
-> 20 x (attention, LSTM step, fc1, vocab projection, fused beam step) -> back-track,
with the synthetic CNN features already resident in HBM.  A pass takes about 5 ms, so a
bench "step" is `--inner` passes back to back over that many different resident batches
(default 10 for the recurrent configurations: 20 steps time about one second instead of a
tenth of one); `value` counts every caption of every pass.  With N > 1 every rank decodes
its own batches (weak scaling, weights replicated, no per-step communication) and the decoded
sequences are all-gathered over RCCL at the end of every pass.

`python bench.py --gpus N` with N > 1 starts its own N ranks (child processes, one per GPU,
started before anything touches a GPU) unless it already runs under torch.distributed.run
(WORLD_SIZE set), in which case it is one of the ranks.

Prints ONE JSON line on rank 0."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    # BASELINE.json configs[1]: the configuration the metric is quoted on
    "beam3_b512_fp32": dict(B=512, W=3, L=64, F=2048, D=256, E=256, H=512, U=512, V=10000, T=20,
                            precision="fp32"),
    "greedy_b32_fp32": dict(B=32, W=1, L=64, F=2048, D=256, E=256, H=512, U=512, V=10000, T=20,
                            precision="fp32"),
    # the headline shape with fp32-accurate products on the bf16 matrix pipe (opt-in mode)
    "beam3_b512_fp32split": dict(B=512, W=3, L=64, F=2048, D=256, E=256, H=512, U=512, V=10000,
                                 T=20, precision="fp32_split"),
    # BASELINE.json configs[2], per-GPU share (512 of the 4096 images), bf16 operands
    "beam5_b512_bf16": dict(B=512, W=5, L=64, F=2048, D=256, E=256, H=512, U=512, V=10000, T=20,
                            precision="bf16"),
    # the headline shape over the GRU sibling decoder (decoder/recurrent_decoder.py)
    "beam3_b512_fp32_gru": dict(B=512, W=3, L=64, F=2048, D=256, E=256, H=512, U=512, V=10000,
                                T=20, precision="fp32", cell="gru"),
    # the baseline beam searcher of the sequence-level trainers (utils/agent_utils.py:134-262)
    # at the headline shape: encoded features in, start-token mask, time from 1, 20 steps
    "searcher_b512_fp32": dict(B=512, W=3, L=64, F=256, D=256, E=256, H=512, U=512, V=10000, T=20,
                               precision="fp32", loop="searcher"),
    # BASELINE.json configs[3]: transformer decoder, 6 layers, d = 512, beam 3, batch 1024
    # (as the reference is written: 512 heads x key_dim 8, filter 2048 - the reference has no
    # params dict for it, 2048 and 8 are the "Attention is all you need" base values)
    "transformer_b1024_fp32": dict(B=1024, W=3, L=64, F=2048, D=512, hidden=512, heads=512, key_dim=8,
                                   filter=2048, layers=6, V=10000, T=20, precision="fp32",
                                   loop="transformer"),
    # the same stack in the conventional layout the engine's flags expose: 8 heads of 64, a
    # causal mask, self-attention keys / values from the normalised input
    "transformer_b1024_fp32_conventional": dict(B=1024, W=3, L=64, F=2048, D=512, hidden=512, heads=8,
                                                key_dim=64, filter=2048, layers=6, V=10000, T=20,
                                                precision="fp32", loop="transformer",
                                                mask_mode="causal", kv_from_normed=True),
    "beam10_v50k_fp32": dict(B=256, W=10, L=196, F=512, D=256, E=256, H=512, U=512, V=50000, T=20,
                             precision="fp32"),
}
# passes per bench step (see the module docstring): sized so that 20 steps time about a second
DEFAULT_INNER = {"transformer_b1024_fp32": 1, "transformer_b1024_fp32_conventional": 1,
                 "beam10_v50k_fp32": 2}
PEAK_F32_MFMA_TFLOPS = 157.3   # dense f32-input MFMA peak of MI355X
PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense bf16 MFMA peak
PEAK_HBM_GBS = 8000.0


def measured_traffic(workload):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes
    (profiles/r*/traffic*.json: 2*FETCH_SIZE + WRITE_SIZE, collected in separate rocprofv3
    runs of this same command); None when no profile of this workload is present."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "traffic*.json")), reverse=True):
        try:
            with open(path) as f:
                d = json.load(f)
            if d.get("config") == workload:
                return d["kernels"]["vocab_projection"]["hbm_bytes_per_launch"]
        except Exception:
            continue
    return None


def cpu_baseline(cfg, budget_s=12.0):
    """The CPU oracle (op-for-op numpy restatement of the reference loop, including the
    un-hoisted W1(features) every step) on a bounded sample of the same workload:
    B=32 images per decode instead of 512, same beam width / dims / steps."""
    from oracle import caption_oracle as co
    try:
        from threadpoolctl import threadpool_info
        threads = max([i.get("num_threads", 1) for i in threadpool_info()] or [1])
    except Exception:
        threads = int(os.environ.get("OMP_NUM_THREADS", "1"))
    Bs = 32
    cell = cfg.get("cell", "lstm")
    if cfg.get("loop") == "searcher":
        return cpu_baseline_searcher(cfg, budget_s)
    p = co.make_params(0, cfg["F"], cfg["D"], cfg["E"], cfg["H"], cfg["V"])
    if cell == "gru":
        p = co.add_gru_params(p, 0)
    img = co.make_images(1, Bs, cfg["L"], cfg["F"])
    co.predict_image_caption_sequence(p, img[:4], cfg["W"], 2, 3, 4, cell=cell)  # warm the BLAS pool
    def rate(hoist, budget):
        reps, t0 = 0, time.perf_counter()
        while True:
            co.predict_image_caption_sequence(p, img, cfg["W"], cfg["T"], 3, 4, 1.0, "reference",
                                              hoist_w1=hoist, cell=cell)
            reps += 1
            el = time.perf_counter() - t0
            if el >= budget or reps >= 50:
                return reps, el
    reps, el = rate(False, budget_s)
    reps_h, el_h = rate(True, budget_s / 2)       # W1(features) hoisted out of the step loop
    return {"value": round(Bs * reps / el, 2), "unit": "captions/s", "cores": threads,
            "kind": "port",
            "sample": "%d decodes of B=%d images (beam %d, %d steps, V=%d), numpy oracle, %.1f s"
                      % (reps, Bs, cfg["W"], cfg["T"], cfg["V"], el),
            "hoisted_value": round(Bs * reps_h / el_h, 2),
            "hoisted_note": "same oracle with the step-invariant W1(features) computed once per "
                            "decode, so that the ratio is not inflated by that inefficiency of "
                            "the reference loop"}


def cpu_baseline_config1_torch(cfg):
    """The CPU figure of the greedy configuration (BASELINE.json configs[0]): B = 32 images,
    beam 1, same dims / vocabulary / steps, on the torch-CPU op-for-op restatement of the
    reference loop (oracle/torch_cpu_reference.py: W1(features) every step, tiled features,
    full [B,W,V] masks, per-row gathers); 3 warm-up + 10 timed whole-batch decodes, median."""
    import statistics
    from oracle import caption_oracle as co
    from oracle import torch_cpu_reference as tr
    threads = max(1, min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "0") or 0) or 16))
    before = torch.get_num_threads()
    torch.set_num_threads(threads)
    try:
        p = tr.to_torch(co.make_params(0, cfg["F"], cfg["D"], cfg["E"], cfg["H"], cfg["V"]))
        img = torch.from_numpy(co.make_images(1, 32, cfg["L"], cfg["F"]))
        times = []
        for i in range(13):
            t0 = time.perf_counter()
            tr.predict_image_caption_sequence(p, img, 1, cfg["T"], 3, 4)
            if i >= 3:
                times.append(time.perf_counter() - t0)
    finally:
        torch.set_num_threads(before)
    return {"value": round(32 / statistics.median(times), 2), "unit": "captions/s", "cores": threads,
            "kind": "port",
            "sample": "greedy, B=32, %d steps, V=%d: 3 warm-up + 10 timed decodes, median; torch-CPU "
                      "op-for-op restatement of the reference loop" % (cfg["T"], cfg["V"])}


def cpu_baseline_searcher(cfg, budget_s):
    """The oracle's restatement of the searcher loop on B=32 encoded feature maps."""
    from oracle import caption_oracle as co
    try:
        from threadpoolctl import threadpool_info
        threads = max([i.get("num_threads", 1) for i in threadpool_info()] or [1])
    except Exception:
        threads = int(os.environ.get("OMP_NUM_THREADS", "1"))
    Bs = 32
    p = co.make_params(0, cfg["F"], cfg["D"], cfg["E"], cfg["H"], cfg["V"])
    feats = co.mlp_encoder(co.make_images(1, Bs, cfg["L"], cfg["F"]), p["enc_kernel"], p["enc_bias"])
    seqs = np.zeros((Bs, cfg["T"] + 1), np.int32)
    reps, t0 = 0, time.perf_counter()
    while True:
        co.baseline_sequence_beam_search(p, feats, seqs, cfg["W"], cfg["V"], cfg["T"] + 1, 3, 4, 0)
        reps += 1
        el = time.perf_counter() - t0
        if el >= budget_s or reps >= 50:
            break
    return {"value": round(Bs * reps / el, 2), "unit": "captions/s", "cores": threads, "kind": "port",
            "sample": "%d searches of B=%d feature maps (beam %d, %d steps, V=%d), numpy oracle, %.1f s"
                      % (reps, Bs, cfg["W"], cfg["T"], cfg["V"], el)}


def transformer_main(args, cfg, device, world, rank):
    """bench line of the transformer decoder path (same contract; dominant kernels: the
    q / k|v / output projections of the layer stack on the exact-f32 MFMA kernels)."""
    import torch.distributed as dist
    from image_caption_amd import synthetic
    from image_caption_amd.distributed import all_gather_sequences
    from image_caption_amd.tengine import TransformerConfig, TransformerEngine
    p = synthetic.make_transformer_params(0, cfg["F"], cfg["D"], cfg["V"], cfg["hidden"], cfg["layers"],
                                          cfg["heads"], cfg["key_dim"], cfg["filter"])
    eng = TransformerEngine(TransformerConfig(tile_mode=args.tile_mode, **cfg), p, device=device)
    img = torch.from_numpy(synthetic.make_images(100 + rank, cfg["B"], cfg["L"], cfg["F"])).to(device)
    tokens = torch.empty(cfg["B"], cfg["T"], dtype=torch.int32, device=device)
    scores = torch.empty(cfg["B"], dtype=torch.float32, device=device)

    def step():
        eng.decode(img, out=tokens, scores=scores)
        return all_gather_sequences(tokens) if world > 1 else tokens

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    eng.set_timing(["proj"])
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    sync()
    elapsed = time.perf_counter() - t0
    proj_ms, proj_n = eng.kernel_times()["proj"]
    eng.set_timing(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert out.shape[0] == cfg["B"] * world
    eng.set_timing(True)
    diag = min(args.steps, 2)
    for _ in range(diag):
        step()
    sync()
    ktimes = eng.kernel_times()
    eng.set_timing(False)
    if rank == 0:
        N, H, HD = cfg["B"] * cfg["W"], cfg["hidden"], cfg["heads"] * cfg["key_dim"]
        passes = cfg["T"] + 1
        # per pass and layer: q (H x HD), k|v (H x 2HD), o (HD x H), cross q (H x HD), cross o
        proj_flops = 2.0 * N * H * HD * 6 * cfg["layers"] * passes * args.steps
        achieved = proj_flops / (proj_ms * 1e-3) / 1e12 if proj_ms else 0.0
        line = {
            "metric": "captions/sec (beam=%d, len %d)" % (cfg["W"], cfg["T"]),
            "value": round(cfg["B"] * world * args.steps / elapsed, 1), "unit": "captions/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.config, "images_per_gpu": cfg["B"], "beam_width": cfg["W"],
                       "decode_steps": cfg["T"], "layers": cfg["layers"], "hidden": H,
                       "heads": cfg["heads"], "key_dim": cfg["key_dim"], "filter": cfg["filter"],
                       "vocab": cfg["V"], "tile_mode": args.tile_mode,
                       "mask_mode": cfg.get("mask_mode", "as_written"),
                       "kv_from_normed": bool(cfg.get("kv_from_normed", False)),
                       "sharding": "images split over ranks, all-gather of sequences"},
            "roofline": {"bound": "mfma", "kernel": "attention projections dense_f32 (q, k|v, output; "
                         "5 launches per layer and pass)", "achieved": round(achieved, 2),
                         "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK_F32_MFMA_TFLOPS, 4), "traffic": None,
                         "avg_launch_us": round(proj_ms / max(proj_n, 1) * 1e3, 2), "launches": proj_n},
            "kernel_ms_per_step": {k: round(v[0] / diag, 4) for k, v in ktimes.items()},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline_transformer(cfg)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline_transformer(cfg, budget_s=12.0):
    """The oracle (full-prefix recompute every step, as the reference's decoder does) on a
    bounded sample: B=4 images, same layer shapes, beam width and steps."""
    from oracle import caption_oracle as co
    try:
        from threadpoolctl import threadpool_info
        threads = max([i.get("num_threads", 1) for i in threadpool_info()] or [1])
    except Exception:
        threads = int(os.environ.get("OMP_NUM_THREADS", "1"))
    Bs = 4
    p = co.make_transformer_params(0, cfg["F"], cfg["D"], cfg["V"], cfg["hidden"], cfg["layers"],
                                   cfg["heads"], cfg["key_dim"], cfg["filter"])
    img = co.make_images(1, Bs, cfg["L"], cfg["F"])
    reps, t0 = 0, time.perf_counter()
    while True:
        co.predict_transformer_caption_sequence(p, img, cfg["W"], cfg["T"], 3, 4, cfg["layers"],
                                                cfg["heads"], cfg["key_dim"],
                                                mask_mode=cfg.get("mask_mode", "as_written"),
                                                kv_from_normed=cfg.get("kv_from_normed", False))
        reps += 1
        el = time.perf_counter() - t0
        if el >= budget_s or reps >= 20:
            break
    return {"value": round(Bs * reps / el, 3), "unit": "captions/s", "cores": threads, "kind": "port",
            "sample": "%d decodes of B=%d images (beam %d, %d steps, %d layers, V=%d), numpy oracle, "
                      "%.1f s" % (reps, Bs, cfg["W"], cfg["T"], cfg["layers"], cfg["V"], el)}


def spawn_ranks(n):
    """Starts this script once per rank (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the
    environment, rendezvous on 127.0.0.1) and waits for all of them.  The parent has not
    touched a GPU: it only starts children and returns the worst exit code; rank 0 prints the
    JSON line on the inherited stdout."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rcs = [p.wait() for p in procs]
    worst = max(rcs, key=abs)
    if worst:
        raise SystemExit("bench rank exit codes: %s" % rcs)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="beam3_b512_fp32", choices=sorted(CONFIGS))
    ap.add_argument("--tile-mode", default="reference", choices=["reference", "repeat"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--inner", type=int, default=0,
                    help="decode passes per bench step (0: the configuration's default)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo + --ranks-share-gpu rehearses the multi-rank path on one GPU")
    ap.add_argument("--gather", default="every", choices=["every", "end"],
                    help="with N > 1: all-gather the decoded sequences after every decode pass "
                         "(default) or once, after the last pass of a step - separates decode time "
                         "from collective time in a scaling run")
    ap.add_argument("--ranks-share-gpu", action="store_true",
                    help="every rank uses cuda:0 (rehearsal on a one-GPU box, gloo only)")
    args = ap.parse_args()
    if args.inner <= 0:
        args.inner = DEFAULT_INNER.get(args.config, 10)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args.gpus)

    import torch.distributed as dist
    from image_caption_amd import synthetic
    from image_caption_amd.distributed import all_gather_sequences
    from image_caption_amd.engine import CaptionConfig, CaptionEngine

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE is %d" % (args.gpus, world))
    if args.ranks_share_gpu:
        if args.backend != "gloo":
            raise SystemExit("--ranks-share-gpu needs --backend gloo (RCCL wants one GPU per rank)")
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group("gloo")

    cfg = dict(CONFIGS[args.config])
    precision = cfg.pop("precision")
    loop = cfg.pop("loop", "decode")
    if loop == "transformer":
        return transformer_main(args, cfg, device, world, rank)
    p = synthetic.make_params(0, cfg["F"], cfg["D"], cfg["E"], cfg["H"], cfg["V"], cfg["U"])
    if cfg.get("cell") == "gru":
        p = synthetic.add_gru_params(p, 0)
    ccfg = CaptionConfig(start_id=3, eos_id=4, length_penalty_weight=1.0, tile_mode=args.tile_mode,
                         precision=precision, **cfg)
    eng = CaptionEngine(ccfg, p, device=device)
    # `inner` different batches per rank, all resident in HBM before anything is timed
    inner = args.inner
    imgs = [torch.from_numpy(synthetic.make_images(100 + rank * 1000 + j, cfg["B"], cfg["L"],
                                                   cfg["F"])).to(device) for j in range(inner)]
    img = imgs[0]
    tokens = torch.empty(cfg["B"], cfg["T"], dtype=torch.int32, device=device)
    scores = torch.empty(cfg["B"], dtype=torch.float32, device=device)

    feats = None
    if loop == "searcher":      # the searcher starts from encoded features [B,L,D], resident in HBM
        from image_caption_amd import ops
        feats = [ops.dense(im.reshape(-1, cfg["F"]), eng.weights["enc_kernel"], eng.weights["enc_bias"],
                           act=ops.ACT_RELU).reshape(cfg["B"], cfg["L"], cfg["D"]).contiguous()
                 for im in imgs]

    def step():
        for j in range(inner):
            if loop == "searcher":
                eng.search_baseline(feats[j], out=tokens, scores=scores)
            else:
                eng.decode(imgs[j], out=tokens, scores=scores)
            if world > 1 and (args.gather == "every" or j == inner - 1):
                out = all_gather_sequences(tokens)
            else:
                out = tokens
        return out

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    # ---- timed region: exactly `steps` steps.  The launches of the dominant kernel (the
    # vocab projection) are bracketed by HIP events on the decode stream inside it - in every
    # fifth decode, an event pair costs the stream a few microseconds per launch; they feed
    # the roofline object. ----
    eng.set_timing(["vocab"], every=5 if args.steps * inner >= 10 else 1)   # a sample of the decodes
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    sync()
    elapsed = time.perf_counter() - t0
    eng.raise_if_invalid()      # (outside the timed region: a decode whose in-launch wait ran out is no result)
    vocab_ms, vocab_n = eng.kernel_times()["vocab"]
    merged_launch = getattr(eng, "merged_launch", False)
    what_ran = {"fold_fc": eng.fold_fc, "merged_launch": bool(merged_launch), "att_table": eng.att_table,
                "fused_vocab": eng.fused_vocab}
    eng.set_timing(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64,
                         device=device if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert out.shape[0] == cfg["B"] * world

    # untimed diagnostic pass: every kernel class bracketed, everything on one stream so
    # that each class is timed alone (not part of `value`)
    eng.set_overlap(False)
    eng.set_timing(True)
    for _ in range(min(args.steps, 5) if inner == 1 else 1):
        step()
    sync()
    diag_steps = min(args.steps, 5) if inner == 1 else inner     # decodes in the diagnostic pass
    ktimes = eng.kernel_times()
    eng.set_timing(False)
    eng.set_overlap(True)

    if rank == 0:
        N = cfg["B"] * cfg["W"]
        captions = cfg["B"] * world * args.steps * inner
        value = captions / elapsed
        # dominant kernel: the vocab projection  [N,H] x [H,V].  On f32 MFMA it is bound by the
        # matrix pipe; with bf16 operands the matrix work shrinks 16x and the fp32 logits
        # write (N*V*4) plus the operand reads bound it by HBM instead.
        n_vocab = vocab_n
        avg_s = vocab_ms / max(vocab_n, 1) * 1e-3
        flops = 2.0 * N * cfg["H"] * cfg["V"]
        if precision == "fp32" and eng.fused_vocab and loop == "decode" and cfg["W"] > 1:
            # the time-0 launch of a decode projects beam 0 only (B rows instead of B*W):
            # algorithmic flops per launch, averaged over the T launches of a decode
            flops = 2.0 * cfg["H"] * cfg["V"] * (N * (cfg["T"] - 1) + cfg["B"]) / cfg["T"]
        if precision == "bf16" and not eng.fused_vocab:
            vbytes = cfg["H"] * cfg["V"] * 2.0 + N * cfg["H"] * 4.0 + N * cfg["V"] * 4.0
            achieved = vbytes / avg_s / 1e9 if avg_s > 0 else 0.0
            roof = {"bound": "hbm", "kernel": "vocab projection gemm_bf16 [N,H]x[H,V] (fp32 logits out)",
                    "achieved": round(achieved, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": round(achieved / PEAK_HBM_GBS, 4)}
        elif precision == "bf16":
            # fused with the top-k epilogue nothing of size N*V is written: matrix-pipe bound
            achieved = flops / avg_s / 1e12 if avg_s > 0 else 0.0
            roof = {"bound": "mfma", "kernel": "vocab projection gemm_bf16 [N,H]x[H,V] + top-k records",
                    "achieved": round(achieved, 2), "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(achieved / PEAK_BF16_MFMA_TFLOPS, 4)}
        elif precision == "fp32_split":
            # six bf16 MFMA products per fp32 product: priced against the bf16 matrix peak
            achieved = 6.0 * flops / avg_s / 1e12 if avg_s > 0 else 0.0
            roof = {"bound": "mfma", "kernel": "vocab projection gemm_split [N,H]x[H,V] (6 bf16 "
                    "products per fp32 product) + top-k records",
                    "achieved": round(achieved, 2), "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(achieved / PEAK_BF16_MFMA_TFLOPS, 4),
                    "fp32_equivalent_TFLOPs": round(flops / avg_s / 1e12, 2) if avg_s > 0 else 0.0}
        else:
            label = "vocab projection dense_f32 [N,H]x[H,V]" + (" + top-k records" if eng.fused_vocab else "")
            if merged_launch:
                # T-1 of a decode's T launches also carry the next step's query projection
                # [N,H]x[H,U] (vector FMAs) and its attention (tanh / softmax / context: no
                # matrix work, not counted) in their first workgroups
                flops += 2.0 * N * cfg["H"] * cfg["U"] * (cfg["T"] - 1) / cfg["T"]
                label += ", carrying the next step's query projection and attention in the same launch"
            achieved = flops / avg_s / 1e12 if avg_s > 0 else 0.0
            roof = {"bound": "mfma", "kernel": label,
                    "achieved": round(achieved, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(achieved / PEAK_F32_MFMA_TFLOPS, 4)}
            if merged_launch:
                # the projection launched alone (diagnostic pass, nothing beside it)
                ms_v, n_v = ktimes["vocab"]
                alone_s = ms_v / max(n_v, 1) * 1e-3
                pflops = 2.0 * cfg["H"] * cfg["V"] * (N * (cfg["T"] - 1) + cfg["B"]) / cfg["T"]
                roof["projection_alone"] = {"avg_launch_us": round(alone_s * 1e6, 2),
                                            "frac": round(pflops / alone_s / 1e12 / PEAK_F32_MFMA_TFLOPS, 4) if alone_s > 0 else 0.0}
        roof.update({"traffic": measured_traffic(args.config), "avg_launch_us": round(avg_s * 1e6, 2),
                     "launches": n_vocab})
        ms_beam, n_beam = ktimes["beam"]
        ms_att, n_att = ktimes["attention"]
        # the beam step reads the [N,V] logits once (unfused kernels) or the projection's
        # top-k records, 8..12 floats per 64 logits (fused path: latency-bound, few bytes)
        fused = eng.fused_vocab
        if fused:
            tk = 1 if cfg["W"] == 1 else (3 if cfg["W"] <= 3 else 5)
            beam_bytes = N * (2 * ((cfg["V"] + 127) // 128)) * (2 + 2 * tk) * 4.0
        else:
            beam_bytes = N * cfg["V"] * 4.0
        beam_gbs = beam_bytes / (ms_beam / max(n_beam, 1) * 1e-3) / 1e9 if ms_beam else 0
        att_bytes = cfg["B"] * cfg["L"] * (cfg["U"] + cfg["D"]) * 4.0 + N * (cfg["U"] + cfg["D"] + cfg["L"]) * 4.0
        att_gbs = att_bytes / (ms_att / max(n_att, 1) * 1e-3) / 1e9 if ms_att else 0
        line = {
            "metric": "captions/sec (beam=%d, len %d)" % (cfg["W"], cfg["T"]), "value": round(value, 1),
            "unit": "captions/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": {"bf16": "bf16", "fp32_split": "f32 (3 x bf16 split operands, f32 accumulate)"}.get(precision, "f32"),
            "data": "synthetic",
            "config": {"workload": args.config, "images_per_gpu": cfg["B"],
                       "decode_passes_per_step": inner, "captions_per_step": cfg["B"] * inner * world,
                       "beam_width": cfg["W"],
                       "decode_steps": cfg["T"], "locations": cfg["L"], "feature_dim": cfg["F"],
                       "hidden": cfg["H"], "vocab": cfg["V"], "tile_mode": args.tile_mode,
                       "sharding": "images split over ranks, all-gather of sequences",
                       "gather": args.gather, **what_ran},
            "roofline": roof,
            "hbm_kernels": {"beam_step_GBps": round(beam_gbs, 1),
                            "beam_step_path": "top-k records of the fused projection" if fused
                            else "logits read once",
                            "beam_step_bytes": int(beam_bytes), "attention_GBps": round(att_gbs, 1),
                            "peak_GBps": PEAK_HBM_GBS},
            "kernel_ms_per_decode_pass": {k: round(v[0] / diag_steps, 4) for k, v in ktimes.items()},
            "ms_per_decode_pass": round(elapsed / args.steps / inner * 1e3, 4),
        }
        line["world_size_reported_by_backend"] = dist.get_world_size() if world > 1 else 1
        line["backend"] = args.backend if world > 1 else "none"
        if world == 1 and precision == "fp32" and loop == "decode" and not args.no_cpu_baseline:
            # informational, not `value`: the same workload with the opt-in precision modes
            # (measured before the CPU baseline, while the GPU clocks are still up)
            eng.close()
            other = {}
            for mode in ("fp32_split", "bf16"):
                e2 = CaptionEngine(CaptionConfig(start_id=3, eos_id=4, length_penalty_weight=1.0,
                                                 tile_mode=args.tile_mode, precision=mode, **cfg),
                                   p, device=device)
                for _ in range(10):
                    e2.decode(img, out=tokens, scores=scores)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(10):
                    e2.decode(img, out=tokens, scores=scores)
                torch.cuda.synchronize()
                other[mode + "_captions_per_s"] = round(10 * cfg["B"] / (time.perf_counter() - t1), 1)
                e2.close()
            line["other_precision_modes"] = other
            if what_ran["fold_fc"]:
                # the reference's op order for the output layer: fc1, then fc2
                # (decoder/lstm_recurrent_decoder.py:106-112) instead of the folded kernel
                e3 = CaptionEngine(ccfg, p, device=device)
                e3.set_fold_fc(False)
                for _ in range(10):
                    e3.decode(img, out=tokens, scores=scores)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(20):
                    e3.decode(imgs[_ % inner], out=tokens, scores=scores)
                torch.cuda.synchronize()
                line["fp32_two_products_captions_per_s"] = round(20 * cfg["B"] / (time.perf_counter() - t1), 1)
                line["fp32_two_products_note"] = ("same engine with fc1 and fc2 as two products, the "
                                                  "reference's op order (set_fold_fc(False)); `value` "
                                                  "runs the folded fc1 . fc2 kernel")
                e3.close()
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(dict(cfg, loop=loop))
            if loop == "decode" and cfg.get("cell", "lstm") == "lstm":
                line["cpu_baseline_config1_torch"] = cpu_baseline_config1_torch(cfg)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
