#!/bin/bash
# Run ON the GPU box (through gpurun): kernel-trace stats + separate PMC passes of the bench command, for the
# headline kernel and for the two other instances the bench line carries as legs (flat-earth default grid,
# configs[2] range-dependent tables).
# usage: bash scripts/collect_profiles.sh <tag>     -> gpurun_out/<tag>_*/   then scripts/summarize_profiles.py <tag> <round>
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-prof}
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-eigenray --no-legs"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-eigenray --no-legs > $R/gpurun_out/${TAG}_stats.log 2>&1
for V in "" "flatearth" "rangedep"; do
  F=""; [ "$V" = "flatearth" ] && F="--flat-earth"; [ "$V" = "rangedep" ] && F="--range-dependent"
  P=${V:+${V}_}
  if [ -n "$V" ]; then
    rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_${P}stats -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-eigenray --no-legs $F > $R/gpurun_out/${TAG}_${P}stats.log 2>&1
  fi
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_${P}$C -- $BENCH $F > $R/gpurun_out/${TAG}_${P}$C.log 2>&1
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_${P}nosave_$C -- $BENCH $F --no-save > $R/gpurun_out/${TAG}_${P}nosave_$C.log 2>&1
  done
  rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_${P}sq -- $BENCH $F > $R/gpurun_out/${TAG}_${P}sq.log 2>&1
done
# configs[2] with trajectories in the sample-blocked layout (PGR_SAMPLE_BLOCKED, what the range_dependent leg times)
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_rangedep_blocked_stats -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-eigenray --no-legs --range-dependent --blocked > $R/gpurun_out/${TAG}_rangedep_blocked_stats.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_rangedep_blocked_$C -- $BENCH --range-dependent --blocked > $R/gpurun_out/${TAG}_rangedep_blocked_$C.log 2>&1
done
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_rangedep_blocked_sq -- $BENCH --range-dependent --blocked > $R/gpurun_out/${TAG}_rangedep_blocked_sq.log 2>&1
python3 -c "import sys; sys.path.insert(0, '$R'); from pygenray_amd import _lib; import json; print(json.dumps({'device_code_sha256': _lib.device_code_sha256(), 'build': _lib.build_info()}))" > $R/gpurun_out/${TAG}_binary.json
echo collected $TAG
