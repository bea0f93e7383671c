#!/bin/bash
# Profiles the default bench.py command on the GPU box (run through gpurun from the repo
# root).  Counter passes are separate runs, as rocprofv3 on this pool requires.
# Afterwards: python tools/summarize_profiles.py  (copies the summaries into profiles/).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/profiles
rm -rf $OUT && mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/stats.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/fetch.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/write.log 2>&1 || exit 1
grep -h '"metric"' $OUT/stats.log
