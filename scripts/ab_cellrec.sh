#!/bin/bash
# Round 6 A/B (verdict item 3b): configs[2] with the four corner nodes of a cell as ONE 64-byte record (scripts/ab/cellrec.so,
# -DPGR_CELL_RECORDS) against the product's row-major {c, cp} table; kernel times (two rounds, alternating) and SQ_WAIT_ANY.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r06b}; mkdir -p $O
cd $R
for round in 1 2; do
  for lib in product cellrec; do
    L=""; [ "$lib" = "cellrec" ] && L="--lib scripts/ab/cellrec.so"
    echo "== round $round $lib" >> $O/ab_cellrec.txt
    timeout -k 10 300 python scripts/kbench.py --slope 2e-4 --modes nosave sample --reps 5 $L >> $O/ab_cellrec.txt 2>&1 || exit 1
  done
done
cd /tmp && export TMPDIR=/tmp
for lib in product cellrec; do
  L=""; [ "$lib" = "cellrec" ] && L="--lib $R/scripts/ab/cellrec.so"
  rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_$lib -- python3 $R/scripts/kbench.py --slope 2e-4 --modes sample --reps 2 $L > $O/pmc_$lib.log 2>&1 || exit 1
done
python3 - <<PY
import csv, glob
for lib in ("product", "cellrec"):
    f = glob.glob("$O/pmc_%s/*/*_counter_collection.csv" % lib)
    per = {}
    for r in csv.DictReader(open(f[0])):
        if "pgr_fan_kernel" in r["Kernel_Name"]:
            per.setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
            per[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    print(lib, {k: v[sorted(v, key=int)[-1]] for k, v in per.items()})
PY
