#!/bin/bash
# config 5 (and the headline) under ICAP_BAND_WIDTH: panels per band of the merged launch's tile order
cd "$GRAFT_REPO_ROOT" || exit 1
for w in 0 4 8 12 16 24; do
  echo "== ICAP_BAND_WIDTH=$w"
  ICAP_BAND_WIDTH=$w timeout -k 10 150 python bench.py --config beam10_v50k_fp32 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" || exit 1
done
