#!/bin/bash
# round 5, GPU call 5: the bench lines committed under profiles/, the six-rank one-GPU rehearsal of the N > 1 path, the GPU suite
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r05c5; mkdir -p $O
timeout -k 10 600 python bench.py > $O/bench_line.json 2> $O/bench.err; echo "bench rc $?"
for V in "--range-dependent" "--range-dependent --blocked" "--flat-earth" "--rays 1000000 --no-save"; do
  N=$(echo $V | tr -d ' -' ); timeout -k 10 300 python bench.py $V --no-cpu-baseline --no-eigenray --no-legs > $O/bench_line_$N.json 2>> $O/bench.err; echo "bench $V rc $?"
done
T0=$(date +%s.%N)
PGR_BENCH_ONE_GPU=1 timeout -k 10 900 python bench.py --gpus 6 --backend gloo > $O/bench_line_6ranks_one_gpu_rehearsal.json 2> $O/rehearsal.err; RC=$?
T1=$(date +%s.%N)
echo "rehearsal rc $RC wall $(echo "$T1 - $T0" | bc) s"; echo "{\"wall_s\": $(echo "$T1 - $T0" | bc), \"rc\": $RC}" > $O/rehearsal_wall.json
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -n 3 $O/pytest.log
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05c5/bench_line.json"))
print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["frac"], d["roofline"]["kernel_ms"], d["roofline"]["traffic"], d.get("roofline_valu",{}).get("frac"), d["lone_wave_ms"])
for k,v in d["legs"].items():
    print(k, json.dumps(v)[:900])
print(d["eigenray"]); print(d["cpu_baseline"]["value"], d["cpu_baseline_c"]["value"])
r=json.load(open("gpurun_out/r05c5/bench_line_6ranks_one_gpu_rehearsal.json"))
print("rehearsal:", r["n_gpus"], r["ranks_joined"], r["value"], json.dumps(r["legs"]["config4"])[:1200], r["eigenray_sharded"])
PY
