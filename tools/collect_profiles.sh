#!/bin/bash
# Profiles a bench.py command on the GPU box (run through gpurun from the repo root):
#   bash tools/collect_profiles.sh [config [steps]]
# Counter passes are separate runs, as rocprofv3 on this pool requires.  Afterwards, here:
#   python tools/summarize_profiles.py gpurun_out/profiles[_config] <round> [config]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
CONFIG=${1:-beam3_b512_fp32}
STEPS=${2:-10}
OUT=gpurun_out/profiles
[ "$CONFIG" != beam3_b512_fp32 ] && OUT=gpurun_out/profiles_$CONFIG
rm -rf $OUT && mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --config $CONFIG --steps $STEPS --warmup 2 --no-cpu-baseline > $OUT/stats.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 bench.py --config $CONFIG --steps 3 --warmup 1 --no-cpu-baseline > $OUT/fetch.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 bench.py --config $CONFIG --steps 3 --warmup 1 --no-cpu-baseline > $OUT/write.log 2>&1 || exit 1
grep -h '"metric"' $OUT/stats.log
