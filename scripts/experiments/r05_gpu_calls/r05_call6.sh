#!/bin/bash
# round 5, GPU call 6: the GPU suite, the instruction budget's PMC passes, and LAST the five-rank one-GPU rehearsal of the N > 1 path
# (six ranks + their launcher are seven processes with the card open: the box's guard allows six)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r05c6; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -n 3 $O/pytest.log
bash scripts/collect_isa_budget.sh b5 > $O/isa_collect.log 2>&1; echo "isa budget rc $?"; tail -n 2 $O/isa_collect.log
T0=$(date +%s.%N)
PGR_BENCH_ONE_GPU=1 timeout -k 10 900 python bench.py --gpus 5 --backend gloo > $O/bench_line_5ranks_one_gpu_rehearsal.json 2> $O/rehearsal.err; RC=$?
T1=$(date +%s.%N)
echo "rehearsal rc $RC wall $(echo "$T1 - $T0" | bc) s"; echo "{\"wall_s\": $(echo "$T1 - $T0" | bc), \"rc\": $RC}" > $O/rehearsal_wall.json
python - <<'PY'
import json
r=json.load(open("gpurun_out/r05c6/bench_line_5ranks_one_gpu_rehearsal.json"))
print("rehearsal:", r["n_gpus"], r["ranks_joined"], r["value"], json.dumps(r["legs"]["config4"])[:1500], r["eigenray_sharded"])
PY
