#!/bin/bash
# A/B of one tuning knob on one bench configuration: tools/ab_env.sh CONFIG VAR VALUE...
# prints captions/s and the dominant kernel's in-situ launch time for every value.
cfg=$1; var=$2; shift 2
for v in "$@"; do
  echo -n "$var=$v: "
  env $var=$v timeout -k 10 120 python bench.py --config $cfg --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['avg_launch_us'], d['kernel_ms_per_decode_pass'])" || exit 1
done
