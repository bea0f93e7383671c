#!/usr/bin/env python3
"""Timeline of one decode pass from a rocprofv3 --kernel-trace CSV: per kernel class the mean
duration, and the mean step period of the main stream's chain (LSTM start to LSTM start) with
the idle time between consecutive kernels of that chain.
usage: trace_timeline.py <dir with *kernel_trace.csv>"""
import collections
import csv
import glob
import os
import sys


def short(name):
    for key in ("dense_topk_beam_kernel", "dense_f32_fast_kernel<128, 128, 4, 2, 2, 3", "dense_f32_fast_kernel", "attention_f32",
                "lstm_f32_wide", "dense_f32_shortk_kernel<8", "dense_f32_shortk", "beam_records", "backtrack",
                "expand_rows", "pool_features", "fill"):
        if key in name:
            return key
    return name[:40]


def main():
    files = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
    rows = []
    with open(files[-1]) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "")))
    rows.sort()
    # the last complete decode: from the second-to-last backtrack to the last
    bt = [i for i, r in enumerate(rows) if r[2] == "backtrack"]
    seg = rows[bt[-2] + 1: bt[-1] + 1]
    print("decode: %.1f us, %d kernels" % ((seg[-1][1] - seg[0][0]) / 1e3, len(seg)))
    dur = collections.defaultdict(list)
    for s, e, n, q in seg:
        dur[n].append((e - s) / 1e3)
    for n, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        print("  %-48s n=%3d mean %7.1f us  total %8.1f" % (n, len(v), sum(v) / len(v), sum(v)))
    lstm = [r for r in seg if r[2] == "lstm_f32_wide"]
    per = [(b[0] - a[0]) / 1e3 for a, b in zip(lstm, lstm[1:])]
    print("step period (lstm start to lstm start): mean %.1f us over %d steps" % (sum(per) / len(per), len(per)))
    # chain on the main queue: everything in the queue of the lstm kernel
    q = lstm[0][3]
    chain = [r for r in seg if r[3] == q]
    gaps = collections.defaultdict(list)
    for a, b in zip(chain, chain[1:]):
        gaps[a[2] + " -> " + b[2]].append((b[0] - a[1]) / 1e3)
    for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1])):
        if len(v) >= 5:
            print("  gap %-70s n=%3d mean %6.1f us" % (k, len(v), sum(v) / len(v)))


if __name__ == "__main__":
    main()
