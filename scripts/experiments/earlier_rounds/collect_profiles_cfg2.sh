#!/bin/bash
# Run ON the GPU box (through gpurun): the range-dependent fan of BASELINE configs[2] (1e5 rays, sofar
# axis sloping 2e-4, tables in HBM/L2) -- kernel-trace stats and separate PMC passes, with and without
# trajectories.   usage: bash scripts/collect_profiles_cfg2.sh <tag>
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-prof_cfg2}
cd /tmp && export TMPDIR=/tmp
K="python3 $R/scripts/kbench.py --slope 2e-4 --reps 3 --modes"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats -- $K nosave sample > $R/gpurun_out/${TAG}_stats.log 2>&1
for M in nosave sample; do
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_${M}_$C -- $K $M > $R/gpurun_out/${TAG}_${M}_$C.log 2>&1
  done
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_${M}_tcc -- $K $M > $R/gpurun_out/${TAG}_${M}_tcc.log 2>&1
  rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_${M}_tcp -- $K $M > $R/gpurun_out/${TAG}_${M}_tcp.log 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_${M}_sq -- $K $M > $R/gpurun_out/${TAG}_${M}_sq.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_${M}_sq2 -- $K $M > $R/gpurun_out/${TAG}_${M}_sq2.log 2>&1
done
echo collected $TAG
