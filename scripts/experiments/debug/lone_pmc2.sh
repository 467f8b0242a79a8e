#!/bin/bash
# Run ON the GPU box: latency-level counters of ONE lone steep wave.  usage: bash scripts/lone_pmc2.sh <tag> <lib.so>...
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
ARGS="--rays 64 --amin -20 --amax -19.9748 --modes nosave --reps 3"
P="SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
for L in "$@"; do
  N=$(basename $L .so)
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_$N -- python3 $R/scripts/kbench.py $ARGS --lib $R/$L > $R/gpurun_out/${TAG}_$N.log 2>&1 || exit 1
done
echo collected $TAG
