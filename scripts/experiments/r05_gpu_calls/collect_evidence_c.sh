#!/bin/bash
# Evidence, part C (one gpurun call, ~8 min): the bench lines committed under profiles/, the instruction budget's PMC passes, the
# random sweeps, and LAST the five-rank one-GPU rehearsal of the N > 1 path (ranks + launcher = six processes on the card: the limit).
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-ev}
cd $R
O=gpurun_out/${TAG}_c; mkdir -p $O
timeout -k 10 600 python bench.py > $O/bench_line.json 2> $O/bench.err; echo "bench rc $?"
timeout -k 10 300 python bench.py --range-dependent --no-cpu-baseline --no-eigenray --no-legs > $O/bench_line_config2.json 2>> $O/bench.err; echo "rc $?"
timeout -k 10 300 python bench.py --range-dependent --blocked --no-cpu-baseline --no-eigenray --no-legs > $O/bench_line_config2_blocked.json 2>> $O/bench.err; echo "rc $?"
timeout -k 10 300 python bench.py --flat-earth --no-cpu-baseline --no-eigenray --no-legs > $O/bench_line_flatearth.json 2>> $O/bench.err; echo "rc $?"
timeout -k 10 300 python bench.py --rays 1000000 --no-save --no-cpu-baseline --no-eigenray --no-legs > $O/bench_line_rays_1e6.json 2>> $O/bench.err; echo "rc $?"
bash scripts/collect_isa_budget.sh ${TAG}b > $O/isa_collect.log 2>&1; echo "isa budget rc $?"
SHA=$(python -c "import sys; sys.path.insert(0,'.'); from pygenray_amd import _lib; print(_lib.device_code_sha256())")
{ echo "# random sweeps (one gpurun call): scripts/fuzz_blocked.py; scripts/fuzz_bitparity.py 12000:12500; scripts/fuzz_bitparity.py 12000:12300 - flatearth";
  python scripts/fuzz_blocked.py 2>&1 | tail -n 1; python scripts/fuzz_bitparity.py 12000:12500 2>&1 | tail -n 2; python scripts/fuzz_bitparity.py 12000:12300 - flatearth 2>&1 | tail -n 2;
  echo "# device_code_sha256 $SHA"; } > $O/fuzz_sweeps.txt; cat $O/fuzz_sweeps.txt
START=$(date +%s)
PGR_BENCH_ONE_GPU=1 timeout -k 10 900 python bench.py --gpus 5 --backend gloo > $O/bench_line_5ranks_one_gpu_rehearsal.json 2> $O/rehearsal.err; RC=$?
echo "rehearsal rc $RC wall $(( $(date +%s) - START )) s"
# (A/B of the non-persistent instances against the previous build, when scripts/ab/prev.so is there)
if [ -f scripts/ab/prev.so ]; then for k in 1 2; do for L in scripts/ab/prev.so pygenray_amd/csrc/libpgr_hip.so; do echo $L; python scripts/kbench.py --lib $L --rays 100000 --modes nosave sample --reps 6 2>&1 | grep kernel | cut -c1-110; done; done > $O/ab_prev.log 2>&1; cat $O/ab_prev.log; fi
