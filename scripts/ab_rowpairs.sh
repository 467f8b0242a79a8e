#!/bin/bash
# Round 6 A/B: configs[2] with rows i, i + 1 interleaved node by node (scripts/ab/rowpairs.so, -DPGR_ROW_PAIRS) against the product.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r06d}; mkdir -p $O
cd $R
for round in 1 2 3; do
  for lib in product rowpairs; do
    L=""; [ "$lib" = "rowpairs" ] && L="--lib scripts/ab/rowpairs.so"
    echo "== round $round $lib" >> $O/ab_rowpairs.txt
    timeout -k 10 300 python scripts/kbench.py --slope 2e-4 --modes nosave sample --reps 5 $L >> $O/ab_rowpairs.txt 2>&1 || exit 1
  done
done
python scripts/fuzz_bitparity.py 50000:50200 scripts/ab/rowpairs.so 2>&1 | tail -n 2 >> $O/ab_rowpairs.txt
grep "==\|kernel\|environments" $O/ab_rowpairs.txt | cut -c1-150
