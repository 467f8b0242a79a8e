#!/bin/bash
# Run ON the GPU box (through gpurun): kernel-trace stats + separate PMC passes of the bench command.
# usage: bash scripts/collect_profiles.sh <tag>     -> gpurun_out/<tag>_{stats,FETCH_SIZE,WRITE_SIZE,sq}/
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-prof}
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-eigenray"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-eigenray > $R/gpurun_out/${TAG}_stats.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_$C -- $BENCH > $R/gpurun_out/${TAG}_$C.log 2>&1
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_nosave_$C -- $BENCH --no-save > $R/gpurun_out/${TAG}_nosave_$C.log 2>&1
done
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_sq -- $BENCH > $R/gpurun_out/${TAG}_sq.log 2>&1
echo collected $TAG
