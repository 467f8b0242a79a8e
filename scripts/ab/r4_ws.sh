cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-eigenray --no-legs --range-dependent"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/ws_blk -- $BENCH --blocked > $R/gpurun_out/ws_blk.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/ws_row -- $BENCH > $R/gpurun_out/ws_row.log 2>&1
cd $R
python - <<'PY'
import csv, glob
for d in ("ws_blk", "ws_row"):
    f = glob.glob(f"gpurun_out/{d}/*/*_counter_collection.csv")[0]
    per = {}
    for r in csv.DictReader(open(f)):
        if "pgr_fan_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "WRITE_SIZE":
            per.setdefault(r["Dispatch_Id"], 0.0); per[r["Dispatch_Id"]] += float(r["Counter_Value"])
    k = sorted(per, key=int)[-1]
    print(d, "WRITE_SIZE GB", per[k] * 1024 / 1e9)
PY
tail -2 gpurun_out/ws_blk.log | cut -c1-600
