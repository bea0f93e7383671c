#!/usr/bin/env python3
"""Mean counter values per kernel from the passes of tools/pmc_kernel.sh:
python tools/pmc_summary.py <tag> <kernel-name substring>"""
import collections
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, pat = sys.argv[1], sys.argv[2]
vals = {}
for p in ("p1", "p2", "p3"):
    files = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "pmc_" + tag, p, "*", "*counter_collection.csv")),
                   key=os.path.getmtime)
    if not files:
        continue
    agg = collections.defaultdict(list)
    with open(files[-1]) as f:
        for r in csv.DictReader(f):
            if pat in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        vals[k] = sum(v) / len(v)
for k in sorted(vals):
    print("%-28s %.4g" % (k, vals[k]))
if "GRBM_GUI_ACTIVE" in vals and "SQ_VALU_MFMA_BUSY_CYCLES" in vals:
    cyc = vals["GRBM_GUI_ACTIVE"] / 8.0          # summed over the 8 XCDs
    print("kernel cycles per XCD        %.4g" % cyc)
    print("MFMA pipe utilisation        %.3f" % (vals["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc)))
    if "SQ_BUSY_CU_CYCLES" in vals:
        print("CU busy fraction             %.3f" % (vals["SQ_BUSY_CU_CYCLES"] / (256 * cyc)))
    if "SQ_WAVE_CYCLES" in vals:
        print("waves resident per SIMD      %.2f" % (4 * vals["SQ_WAVE_CYCLES"] / (1024 * cyc)))
