#!/bin/bash
# Run ON the GPU box: where a fan-kernel wave waits -- LDS / scalar-memory / vector-memory latencies (LEVEL / INSTS), FIFO
# stalls, branches -- for the lone steepest wave and for the fan, with trajectories and end state only.
# usage: bash scripts/collect_wait_counters.sh     -> gpurun_out/wt_*/
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for W in lone fan; do
 for M in sample nosave; do
  A="--modes $M --reps 3"; [ "$W" = "lone" ] && A="$A --rays 64 --amin -20 --amax -19.975"
  k=0
  for C in "SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY"; do
    k=$((k+1))
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/wt_${W}_${M}_$k -- python3 $R/scripts/kbench.py $A > $R/gpurun_out/wt_${W}_${M}_$k.log 2>&1 || echo "pass $W $M $k failed"
  done
 done
done
echo done
