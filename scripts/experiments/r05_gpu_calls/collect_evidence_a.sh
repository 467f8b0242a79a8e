#!/bin/bash
# Evidence, part A (one gpurun call, ~12 min): the GPU suite, persistent-wave modes, every rocprofv3 pass, wave / service timing.
# usage: bash scripts/experiments/r05_gpu_calls/collect_evidence_a.sh <tag>      then: python scripts/summarize_profiles.py <tag> rNN ; python scripts/experiments/r05_gpu_calls/summarize_evidence.py <tag> rNN
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-ev}
cd $R
O=gpurun_out/${TAG}_a; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -n 3 $O/pytest.log
python scripts/persist_check.py --out $O/persist_check.json > $O/persist_check.log 2>&1; echo "persist_check rc $?"
bash scripts/collect_profiles.sh $TAG > $O/collect.log 2>&1; echo "collect_profiles rc $?"
bash scripts/collect_profiles_r05.sh $TAG > $O/collect_r05.log 2>&1; echo "collect_profiles_r05 rc $?"
python scripts/wave_times.py --persistent 0 --out $O/wave_times_static_1e6.json > $O/wt0.log 2>&1
python scripts/wave_times.py --persistent 1 --out $O/wave_times_persistent_1e6.json > $O/wt1.log 2>&1
python scripts/wave_times.py --rays 100000 --save --out $O/wave_times_1e5_trajectories.json > $O/wt2.log 2>&1
python scripts/service_times.py scripts/ab/svctiming.so --out=$O/service_times.json > $O/service_times.log 2>&1
python -c "
import sys; sys.path.insert(0, '.')
from pygenray_amd import _lib; import json
print(json.dumps({'product': _lib.device_code_sha256(), 'wavetimes': _lib.device_code_sha256('scripts/ab/wavetimes.so'), 'svctiming': _lib.device_code_sha256('scripts/ab/svctiming.so')}))" > $O/shas.json
cat $O/shas.json; grep -h persistent_ms $O/persist_check.log | cut -c1-330
