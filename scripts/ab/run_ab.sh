# usage: run_ab.sh lib1 lib2 ...   (names under scripts/ab, without .so); two rounds, lone steep wave + 1e5 fan
set -e
mkdir -p gpurun_out
rm -f gpurun_out/ab_kb.log
for round in 1 2; do
for l in "$@"; do
  echo "LIB $l" >> gpurun_out/ab_kb.log
  python scripts/kbench.py --lib scripts/ab/$l.so --modes nosave sample --reps 7 >> gpurun_out/ab_kb.log 2>&1
  python scripts/kbench.py --lib scripts/ab/$l.so --rays 64 --amin 19.7 --amax 19.8 --modes nosave sample --reps 7 >> gpurun_out/ab_kb.log 2>&1
done
done
