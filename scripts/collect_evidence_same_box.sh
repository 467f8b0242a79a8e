#!/bin/bash
# The bench line AND the rocprofv3 passes of the same commands in ONE gpurun call (one box: boxes differ by a few per cent, and the
# committed kernel stats must agree with the committed bench line's HIP-event times).
# usage: bash scripts/collect_evidence_same_box.sh <tag>   then: summarize_profiles.py <tag> rNN ; copy gpurun_out/<tag>_c/bench_line*.json
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-ev}
cd $R
O=gpurun_out/${TAG}_c; mkdir -p $O
timeout -k 10 600 python bench.py > $O/bench_line.json 2> $O/bench.err; echo "bench rc $?"
timeout -k 10 300 python bench.py --range-dependent --no-cpu-baseline --no-eigenray --no-legs > $O/bench_line_config2.json 2>> $O/bench.err; echo "rc $?"
timeout -k 10 300 python bench.py --range-dependent --blocked --no-cpu-baseline --no-eigenray --no-legs > $O/bench_line_config2_blocked.json 2>> $O/bench.err; echo "rc $?"
timeout -k 10 300 python bench.py --flat-earth --no-cpu-baseline --no-eigenray --no-legs > $O/bench_line_flatearth.json 2>> $O/bench.err; echo "rc $?"
timeout -k 10 300 python bench.py --rays 1000000 --no-save --no-cpu-baseline --no-eigenray --no-legs > $O/bench_line_rays_1e6.json 2>> $O/bench.err; echo "rc $?"
bash scripts/collect_profiles.sh $TAG > $O/collect.log 2>&1; echo "collect_profiles rc $?"
bash scripts/collect_profiles_r05.sh $TAG > $O/collect_r05.log 2>&1; echo "collect_profiles_r05 rc $?"
bash scripts/collect_profiles_r06.sh $TAG > $O/collect_r06.log 2>&1; echo "collect_profiles_r06 rc $?"
timeout -k 10 300 python bench.py --rays 1000000 --no-cpu-baseline --no-eigenray --no-legs > $O/bench_line_rays_1e6_traj.json 2>> $O/bench.err; echo "rc $?"
python - <<'PY'
import json, glob, csv, os
d = json.load(open(os.path.join("gpurun_out", os.environ.get("TAGX", ""), "bench_line.json"))) if False else None
PY
for f in gpurun_out/${TAG}_stats/*/*_kernel_stats.csv gpurun_out/${TAG}_1e6_stats/*/*_kernel_stats.csv; do sed -n 2p $f | cut -c1-140; done
python -c "
import json
d=json.load(open('$O/bench_line.json')); print('bench:', d['roofline']['kernel_ms'], d['legs']['rays_1e6']['end_state']['kernel_ms'])"
