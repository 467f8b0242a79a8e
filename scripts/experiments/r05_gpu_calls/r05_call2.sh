#!/bin/bash
# round 5, GPU call 2: the whole GPU suite with the new tests, the service's cycle split, one full bench line
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r05c2; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -n 5 $O/pytest.log
python scripts/service_times.py scripts/ab/svctiming.so --out=$O/service_times.json > $O/service_times.log 2>&1; cat $O/service_times.log | grep -v amdgpu.ids
timeout -k 10 600 python bench.py > $O/bench_line.json 2> $O/bench.err; echo "bench rc $?"; python - <<'PY'
import json
d=json.load(open("gpurun_out/r05c2/bench_line.json"))
print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["frac"], d["roofline"]["kernel_ms"], d["lone_wave_ms"])
for k,v in d["legs"].items():
    print(k, json.dumps(v)[:600])
print(d["eigenray"])
PY
