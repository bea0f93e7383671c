#!/bin/bash
# Config 5 (beam 10, 50k vocabulary) under the schedule knobs: captions/s per variant.
run() {
  echo "$*"
  env "$@" timeout -k 10 150 python bench.py --config beam10_v50k_fp32 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('   ', d['value'], d['roofline'].get('avg_launch_us'))
" || exit 1
}
run ICAP_X=0
run ICAP_FOLD_FC=0
run ICAP_SIDE_FQ=1
run ICAP_FOLD_FC=0 ICAP_SIDE_FQ=1
run ICAP_VOCAB_GATE=0
