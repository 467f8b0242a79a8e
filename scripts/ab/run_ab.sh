# A/B of library variants built by scripts/build_variants.py, ON the GPU box (through gpurun), two rounds each:
#   bash scripts/ab/run_ab.sh v0 v1 ...          (names under scripts/ab, without .so)
# the 1e5-ray headline fan and the lone wave of its steepest rays (whose latency is the fan's run time);
# KB_EXTRA="--slope 2e-4" switches both to the range-dependent tables of configs[2].   -> gpurun_out/ab_kb.log
set -e
mkdir -p gpurun_out
rm -f gpurun_out/ab_kb.log
for round in 1 2; do
for l in "$@"; do
  echo "LIB $l" >> gpurun_out/ab_kb.log
  python scripts/kbench.py --lib scripts/ab/$l.so --modes nosave sample --reps 7 $KB_EXTRA >> gpurun_out/ab_kb.log 2>&1
  python scripts/kbench.py --lib scripts/ab/$l.so --rays 64 --amin 19.7 --amax 19.8 --modes nosave sample --reps 7 $KB_EXTRA >> gpurun_out/ab_kb.log 2>&1
done
done
grep -v amdgpu.ids gpurun_out/ab_kb.log | awk '/^LIB/{printf "\n%s ", $2} /^mode/{printf "%s %s %s | ", $1, $4, $8}'; echo
