#!/bin/bash
# Run ON the GPU box: PMC of ONE wave of the steepest rays alone on the chip, for two libraries.
# usage: bash scripts/lone_pmc.sh <tag> [<base.so>]
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-lone}
BASE=$2
cd /tmp && export TMPDIR=/tmp
ARGS="--rays 64 --amin -20 --amax -19.9748 --modes nosave --reps 3"
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY"
P2="SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY"
rocprofv3 --pmc $P1 --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_new1 -- python3 $R/scripts/kbench.py $ARGS > $R/gpurun_out/${TAG}_new1.log 2>&1
rocprofv3 --pmc $P2 --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_new2 -- python3 $R/scripts/kbench.py $ARGS > $R/gpurun_out/${TAG}_new2.log 2>&1
if [ -n "$BASE" ]; then
rocprofv3 --pmc $P1 --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_base1 -- python3 $R/scripts/kbench.py $ARGS --lib $R/$BASE > $R/gpurun_out/${TAG}_base1.log 2>&1
rocprofv3 --pmc $P2 --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_base2 -- python3 $R/scripts/kbench.py $ARGS --lib $R/$BASE > $R/gpurun_out/${TAG}_base2.log 2>&1
fi
echo collected $TAG
