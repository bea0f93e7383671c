#!/bin/bash
# L2 fills (FETCH_SIZE) of the merged launch in config 5 under ICAP_BAND_WIDTH, one counter pass each
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for w in "$@"; do
  export ICAP_BAND_WIDTH=$w
  rm -rf gpurun_out/bw_fetch_$w
  timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/bw_fetch_$w -- python3 bench.py --config beam10_v50k_fp32 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/bw_fetch_$w.log 2>&1 || exit 1
  python3 - $w <<'PY'
import csv, glob, sys
w = sys.argv[1]
f = glob.glob("gpurun_out/bw_fetch_%s/*/*counter_collection.csv" % w)[0]
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "dense_topk_attq" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
print("ICAP_BAND_WIDTH=%s: merged launch FETCH_SIZE mean %.0f KB over %d launches -> 2*FETCH = %.0f MB" % (w, sum(v) / len(v), len(v), 2 * sum(v) / len(v) / 1024))
PY
done
