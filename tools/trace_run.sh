#!/bin/bash
# kernel trace of a short bench run under an environment setting: tools/trace_run.sh NAME [VAR=VALUE...]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
name=$1; shift
for kv in "$@"; do export "$kv"; done
rm -rf gpurun_out/trace_$name
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_$name -- python3 bench.py --steps 1 --warmup 1 --inner 4 --no-cpu-baseline > gpurun_out/trace_$name.log 2>&1 || exit 1
python3 tools/trace_timeline.py gpurun_out/trace_$name
