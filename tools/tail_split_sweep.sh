#!/bin/bash
# Sweep of the merged launch's tail split (ICAP_TAIL_SPLIT: last vocab tiles cut into two
# 64 x 128 workgroups) against the position of the attention workgroups in the grid
# (ICAP_ATTQ_TILES_BEFORE) at the headline shape; each point is its own process.
out=${1:-gpurun_out/r3/tail_split_sweep.log}
: > $out
for tb in ${TBS:-720 640 576}; do
  for sp in ${SPLITS:-0 64 128 192 224}; do
    echo "== tiles_before $tb split $sp" >> $out
    ICAP_ATTQ_TILES_BEFORE=$tb ICAP_TAIL_SPLIT=$sp timeout -k 5 120 python tools/headline_rate.py --reps 2 2>&1 | grep captions >> $out || exit 1
  done
done
