#!/bin/bash
# the conventional-layout transformer config under the small-product knobs of dense_f32
cd "$GRAFT_REPO_ROOT" || exit 1
run() { echo "== $*"; env "$@" timeout -k 10 200 python bench.py --config transformer_b1024_fp32_conventional --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['kernel_ms_per_step'])" || exit 1; }
run X=0
run ICAP_SHORTK=0
run ICAP_SMALL_KS=1
run ICAP_SMALL_KS=2
