#!/bin/bash
# PMC passes over a short bench.py run:  bash tools/pmc_bench.sh <tag> <bench.py arguments...>
# (run through gpurun; the counters of each pass stay within the per-block limits; summarise
# with  python tools/pmc_summary.py <tag> <kernel-name substring>)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
TAG="$1"; shift
O=gpurun_out/pmc_$TAG; rm -rf $O; mkdir -p $O
timeout -s KILL 150 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY --output-format csv -d $O/p1 -- python3 bench.py "$@" --steps 2 --warmup 1 --inner 2 --no-cpu-baseline > $O/p1.log 2>&1 || exit 1
timeout -s KILL 150 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INST_LEVEL_VMEM SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_BUSY_CU_CYCLES SQ_LDS_IDX_ACTIVE --output-format csv -d $O/p2 -- python3 bench.py "$@" --steps 2 --warmup 1 --inner 2 --no-cpu-baseline > $O/p2.log 2>&1 || exit 1
timeout -s KILL 150 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU --output-format csv -d $O/p3 -- python3 bench.py "$@" --steps 2 --warmup 1 --inner 2 --no-cpu-baseline > $O/p3.log 2>&1 || exit 1
echo done
