#!/bin/bash
# round 5, GPU call 3: rocprofv3 evidence with the round's final binary (kernel stats + PMC passes), wave / service timing
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
bash scripts/collect_profiles.sh p5 > gpurun_out/p5_collect.log 2>&1; echo "collect_profiles rc $?"
bash scripts/collect_profiles_r05.sh p5 > gpurun_out/p5_collect_r05.log 2>&1; echo "collect_profiles_r05 rc $?"
O=gpurun_out/r05c3; mkdir -p $O
python scripts/wave_times.py --persistent 0 --out $O/wave_times_static_1e6.json > $O/wt0.log 2>&1
python scripts/wave_times.py --persistent 1 --out $O/wave_times_persistent_1e6.json > $O/wt1.log 2>&1
python scripts/wave_times.py --rays 100000 --save --out $O/wave_times_1e5_trajectories.json > $O/wt2.log 2>&1
python scripts/service_times.py scripts/ab/svctiming.so --out=$O/service_times.json > $O/service_times.log 2>&1
python -c "
import sys; sys.path.insert(0, '.')
from pygenray_amd import _lib; import json
print(json.dumps({'product': _lib.device_code_sha256(), 'wavetimes': _lib.device_code_sha256('scripts/ab/wavetimes.so'), 'svctiming': _lib.device_code_sha256('scripts/ab/svctiming.so')}))" > $O/shas.json
ls gpurun_out | head -80; tail -n 3 gpurun_out/p5_collect.log gpurun_out/p5_collect_r05.log
