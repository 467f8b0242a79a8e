"""Per-call durations of the fan kernel from rocprofv3's kernel trace, beside the `--stats` averages.

rocprofv3's `*_kernel_stats.csv` averages EVERY call of a kernel, the bench's cold warm-up passes included (round 5: 5.764 ms
over 12 calls of which two are cold, against 5.60 ms from the bench's HIP events over its 20 timed steps), so a `frac`
recomputed from the stats average and the one in the bench line differ by the warm-up.  This lists the calls themselves
(dispatch order, ms) and the average over the warm ones -- the calls behind the bench's own timed region.

usage: python scripts/summarize_kernel_calls.py <tag> <round> [--warmup W]    e.g.  r06 r06
reads  gpurun_out/<tag>_*stats/*/*_kernel_trace.csv   writes  profiles/<round>_kernel_calls.json
"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, rnd = sys.argv[1], sys.argv[2]
warm_skip = int(sys.argv[sys.argv.index("--warmup") + 1]) if "--warmup" in sys.argv else 2
G = os.path.join(ROOT, "gpurun_out")
out = {"note": "rocprofv3 --kernel-trace: every dispatch of pgr_fan_kernel of the profiled command, in dispatch order (ms); "
               f"`warm_mean_ms` leaves out the first {warm_skip} (the command's --warmup passes, which include the cold one); "
               "`stats_mean_ms` is what rocprofv3 --stats averages (all calls)"}
for d in sorted(glob.glob(os.path.join(G, f"{tag}_*stats"))):
    name = os.path.basename(d)[len(tag) + 1:]
    traces = glob.glob(os.path.join(d, "*", "*_kernel_trace.csv"))
    if not traces:
        continue
    per = {}
    for r in csv.DictReader(open(traces[0])):
        k = r["Kernel_Name"]
        if "pgr_fan_kernel" in k:
            per.setdefault(k, []).append((int(r["Dispatch_Id"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6))
    for k, calls in per.items():
        ms = [round(v, 4) for _, v in sorted(calls)]
        warm = ms[warm_skip:] if len(ms) > warm_skip else ms
        out.setdefault(name, []).append({"kernel": k.split("(")[0], "calls_ms": ms, "stats_mean_ms": round(sum(ms) / len(ms), 4),
                                         "warm_mean_ms": round(sum(warm) / len(warm), 4), "min_ms": min(ms), "n_calls": len(ms)})
p = os.path.join(ROOT, "profiles", f"{rnd}_kernel_calls.json")
json.dump(out, open(p, "w"), indent=1)
print("wrote", p)
for k, v in out.items():
    if k != "note":
        for e in v:
            print(f"{k:28s} {e['kernel'][:46]:46s} n={e['n_calls']:3d} stats mean {e['stats_mean_ms']:.3f}  warm mean {e['warm_mean_ms']:.3f}  min {e['min_ms']:.3f}")
