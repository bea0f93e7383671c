#!/usr/bin/env python3
"""Turns the rocprofv3 output of tools/collect_profiles.sh (gpurun_out/profiles/{stats,fetch,
write}) into the committed summaries: profiles/<round>/bench_kernel_stats[_<config>].csv and
profiles/<round>/traffic[_<config>].json (mean FETCH_SIZE / WRITE_SIZE per kernel class and HBM bytes per
launch = 2*FETCH + WRITE: gfx950's FETCH_SIZE counts half of a wide coalesced read)."""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def classify(name, grid, block):
    if "dense_topk_attq_kernel" in name or "dense_topk_attention_kernel" in name:
        return "vocab_projection"   # the merged launch: projection tiles + next step's attention
    if "dense_f32_fast_kernel<128" in name:
        topk = not name.rstrip(")").rstrip("(icap::(anonymous namespace)::DenseArgs").endswith(", 0>")
        return "vocab_projection" if (topk or grid // max(block, 1) == 948) else "dense128_other"
    m = re.search(r"gemm_bf16_kernel<\d+, (?:true|false), (\d+)", name)
    if m or "vocab_bf16_resident" in name:
        return "vocab_projection" if (not m or m.group(1) != "0") else "gemm_bf16"
    for key, cls in (("dense_f32_small_kernel", "small_dense"), ("dense_f32_shortk_kernel", "small_dense"), ("dense_f32_fast_kernel<64", "small_dense"),
                     ("lstm_f32", "lstm"), ("attention_f32", "attention"), ("beam_records", "beam_records"),
                     ("beam_row", "beam_row"), ("beam_merge", "beam_merge"), ("lstm_bf16", "lstm")):
        if key in name:
            return cls
    return None


def main():
    src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "profiles")
    rnd = sys.argv[2] if len(sys.argv) > 2 else "r2"
    config = sys.argv[3] if len(sys.argv) > 3 else "beam3_b512_fp32"
    suffix = "" if config == "beam3_b512_fp32" else "_" + config
    dst = os.path.join(ROOT, "profiles", rnd)
    os.makedirs(dst, exist_ok=True)
    newest = lambda paths: sorted(paths, key=os.path.getmtime)[-1:]   # merged runs pile up locally
    stats = newest(glob.glob(os.path.join(src, "stats", "*", "*kernel_stats.csv")))
    if stats:
        shutil.copy(stats[0], os.path.join(dst, "bench_kernel_stats%s.csv" % suffix))
    res = {}
    for sub, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        files = newest(glob.glob(os.path.join(src, sub, "*", "*counter_collection.csv")))
        if not files:
            continue
        agg = collections.defaultdict(list)
        with open(files[0]) as f:
            for r in csv.DictReader(f):
                if r["Counter_Name"] != counter:
                    continue
                cls = classify(r["Kernel_Name"], int(r["Grid_Size"]), int(r["Workgroup_Size"]))
                if cls:
                    agg[cls].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            res.setdefault(k, {})[counter + "_KB_mean"] = round(sum(v) / len(v), 1)
            res[k]["launches_" + sub] = len(v)
    for v in res.values():
        v["hbm_bytes_per_launch"] = int((2 * v.get("FETCH_SIZE_KB_mean", 0) + v.get("WRITE_SIZE_KB_mean", 0)) * 1024)
    out = {"command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 bench.py "
                      "--config %s --steps 3 --warmup 1 --no-cpu-baseline" % config,
           "config": config,
           "note": "hbm_bytes_per_launch = 2*FETCH_SIZE + WRITE_SIZE (KB counters x 1024)",
           "kernels": res}
    with open(os.path.join(dst, "traffic%s.json" % suffix), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
