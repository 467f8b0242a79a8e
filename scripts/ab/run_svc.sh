set -e
mkdir -p gpurun_out
rm -f gpurun_out/svc_kb.log
python scripts/bitparity.py scripts/ab/svc3e.so 20 > gpurun_out/svc_bpe.log 2>&1
python scripts/bitparity.py scripts/ab/svc3e.so 20 2 > gpurun_out/svc_bp_c2.log 2>&1
for l in base svc2e svc3e base svc2e svc3e; do
  python scripts/kbench.py --lib scripts/ab/$l.so --modes nosave sample --reps 7 >> gpurun_out/svc_kb.log 2>&1
  python scripts/kbench.py --lib scripts/ab/$l.so --rays 64 --amin 19.7 --amax 19.8 --modes nosave sample --reps 7 >> gpurun_out/svc_kb.log 2>&1
done
python scripts/kbench.py --lib scripts/ab/base.so --rays 1000000 --modes nosave sample --reps 3 >> gpurun_out/svc_kb.log 2>&1
python scripts/kbench.py --lib scripts/ab/svc3e.so --rays 1000000 --modes nosave sample --reps 3 >> gpurun_out/svc_kb.log 2>&1
python scripts/kbench.py --lib scripts/ab/base.so --slope 2e-4 --modes nosave sample --reps 3 >> gpurun_out/svc_kb.log 2>&1
python scripts/kbench.py --lib scripts/ab/svc3e.so --slope 2e-4 --modes nosave sample --reps 3 >> gpurun_out/svc_kb.log 2>&1
grep "all ok" gpurun_out/svc_bpe.log gpurun_out/svc_bp_c2.log
