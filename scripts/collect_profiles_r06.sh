#!/bin/bash
# Run ON the GPU box (through gpurun): round 6's additions to scripts/collect_profiles.sh / collect_profiles_r05.sh -- the 1e6-ray fan
# WITH S = 1001 trajectories (pgr_fan_kernel<true, 4, 1, true>, 24 GB of samples: kernel stats and the HBM-traffic passes the
# verdict asked for) and the headline fan's end-state instance.
# usage: bash scripts/collect_profiles_r06.sh <tag>   -> gpurun_out/<tag>_*/   then scripts/summarize_profiles.py <tag> r06
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-prof}
cd /tmp && export TMPDIR=/tmp
B6="python3 $R/bench.py --rays 1000000 --no-cpu-baseline --no-eigenray --no-legs"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_1e6traj_stats -- $B6 --steps 6 --warmup 2 > $R/gpurun_out/${TAG}_1e6traj_stats.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_1e6traj_$C -- $B6 --steps 2 --warmup 1 > $R/gpurun_out/${TAG}_1e6traj_$C.log 2>&1
done
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_1e6traj_sq -- $B6 --steps 2 --warmup 1 > $R/gpurun_out/${TAG}_1e6traj_sq.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_nosave_stats -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-eigenray --no-legs --no-save > $R/gpurun_out/${TAG}_nosave_stats.log 2>&1
echo collected $TAG r06
