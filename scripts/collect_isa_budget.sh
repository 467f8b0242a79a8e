#!/bin/bash
# Run ON the GPU box (through gpurun): PMC passes for the instruction budget of the fan kernels
# (profiles/rNN_isa_budget.json, scripts/summarize_isa_budget.py).   usage: bash scripts/collect_isa_budget.sh <tag>
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-budget}
cd /tmp && export TMPDIR=/tmp
for W in headline rangedep flatearth; do
  for F in full quiet steep; do
    for S in 0 1 3; do
      [ "$S" = "3" ] && [ "$W" != "rangedep" ] && continue
      [ "$F" = "steep" ] && [ "$W" != "headline" ] && continue
      K=${W}_${F}_$S
      rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv \
        -d $R/gpurun_out/${TAG}_$K -- python3 $R/scripts/isa_budget_run.py $W $F $S > $R/gpurun_out/${TAG}_$K.log 2>&1
      rocprofv3 --pmc SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_WAVES --kernel-trace --output-format csv \
        -d $R/gpurun_out/${TAG}_${K}_b -- python3 $R/scripts/isa_budget_run.py $W $F $S > $R/gpurun_out/${TAG}_${K}_b.log 2>&1 || true
      echo $K done
    done
  done
done
python3 -c "import sys; sys.path.insert(0, '$R'); from pygenray_amd import _lib; import json; print(json.dumps({'device_code_sha256': _lib.device_code_sha256(), 'build': _lib.build_info()}))" > $R/gpurun_out/${TAG}_binary.json
echo collected $TAG
