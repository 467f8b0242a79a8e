#!/bin/bash
# Run ON the GPU box (through gpurun): round 5's additions to scripts/collect_profiles.sh -- the 1e6-ray leg (persistent
# waves), the lone steepest wave, and pr.shoot_rays on configs[2] (sample-blocked kernel + un-blocking pass).
# usage: bash scripts/collect_profiles_r05.sh <tag>   -> gpurun_out/<tag>_*/   then scripts/summarize_profiles.py <tag> r05
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-prof}
cd /tmp && export TMPDIR=/tmp
B6="python3 $R/bench.py --rays 1000000 --no-save --no-cpu-baseline --no-eigenray --no-legs"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_1e6_stats -- $B6 --steps 8 --warmup 2 > $R/gpurun_out/${TAG}_1e6_stats.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_1e6_$C -- $B6 --steps 2 --warmup 1 > $R/gpurun_out/${TAG}_1e6_$C.log 2>&1
done
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_1e6_sq -- $B6 --steps 2 --warmup 1 > $R/gpurun_out/${TAG}_1e6_sq.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_lone_stats -- python3 $R/scripts/kbench.py --rays 64 --amin -20 --amax -19.9748 --modes nosave sample --reps 10 > $R/gpurun_out/${TAG}_lone_stats.log 2>&1
for M in blocked rows; do
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_api2_${M}_$C -- python3 $R/scripts/api_cfg2_run.py $M > $R/gpurun_out/${TAG}_api2_${M}_$C.log 2>&1
  done
done
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_api2_stats -- python3 $R/scripts/api_cfg2_run.py blocked > $R/gpurun_out/${TAG}_api2_stats.log 2>&1
echo collected $TAG r05
