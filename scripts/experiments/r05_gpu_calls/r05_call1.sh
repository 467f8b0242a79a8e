#!/bin/bash
# round 5, first GPU call: persistent waves -- same bits as the static deal? where does the slot time go? A/B against round 4's binary
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r05c1; mkdir -p $O
python scripts/regress.py --check scripts/regress_ref.json > $O/regress.log 2>&1 && echo "regress OK" || { echo "regress FAILED"; tail -5 $O/regress.log; }
python scripts/persist_check.py --out $O/persist_check.json > $O/persist_check.log 2>&1 && echo "persist_check OK" || { echo "persist_check FAILED"; tail -8 $O/persist_check.log; }
python scripts/wave_times.py --persistent 0 --out $O/wt_static_1e6.json > $O/wt_static_1e6.log 2>&1
python scripts/wave_times.py --persistent 1 --out $O/wt_persist_1e6.json > $O/wt_persist_1e6.log 2>&1
python scripts/wave_times.py --rays 100000 --save --out $O/wt_1e5_save.json > $O/wt_1e5_save.log 2>&1
for L in scripts/ab/base.so pygenray_amd/csrc/libpgr_hip.so; do
  for k in 1 2; do
    python scripts/kbench.py --lib $L --rays 100000 --modes nosave sample --reps 5 >> $O/ab_1e5.log 2>&1
    python scripts/kbench.py --lib $L --rays 64 --amin -20 --amax -19.975 --modes nosave sample --reps 5 >> $O/ab_lone.log 2>&1
  done
  python scripts/kbench.py --lib $L --rays 1000000 --modes nosave --reps 3 >> $O/ab_1e6.log 2>&1
  python scripts/kbench.py --lib $L --rays 300000 --modes nosave sample --reps 3 >> $O/ab_3e5.log 2>&1
  python scripts/kbench.py --lib $L --rays 100000 --slope 2e-4 --modes nosave sample --reps 5 >> $O/ab_cfg2.log 2>&1
done
tail -n 3 $O/persist_check.log; cat $O/ab_1e5.log $O/ab_lone.log $O/ab_1e6.log $O/ab_3e5.log $O/ab_cfg2.log | cut -c1-150
