"""rocprofv3 outputs of scripts/collect_profiles_cfg2.sh (merged back under gpurun_out/) ->
profiles/<round>_config2_counters.json and profiles/<round>_config2_kernel_stats.csv.
usage: python scripts/summarize_profiles_cfg2.py <tag> <round>     e.g.  pc2b r02"""
import csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, rnd = sys.argv[1], sys.argv[2]
G = os.path.join(ROOT, "gpurun_out")


def fan_counters(d):
    out = {}
    for f in glob.glob(os.path.join(G, d, "*", "*_counter_collection.csv")):
        per = {}
        for r in csv.DictReader(open(f)):
            if "pgr_fan_kernel" in r["Kernel_Name"]:
                per.setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
                per[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
        out.update({k: v[sorted(v, key=int)[-1]] for k, v in per.items()})   # last dispatch = a timed repetition
    return out


res = {"_note": f"{rnd}: BASELINE configs[2] (range-dependent Munk, sofar axis sloping 2e-4, 101 columns, 1e5 rays, 1000 km), "
                "fan kernel pgr_fan_kernel<false, 4, SAVE>, last dispatch of each rocprofv3 --pmc pass of `python3 scripts/kbench.py "
                "--slope 2e-4 --reps 3 --modes <mode>` (scripts/collect_profiles_cfg2.sh); FETCH_SIZE / WRITE_SIZE in KB "
                "(FETCH_SIZE counts half the bytes on gfx950)"}
for key, mode in (("end_state_only", "nosave"), ("trajectories", "sample")):
    c = {}
    for p in ("FETCH_SIZE", "WRITE_SIZE", "tcc", "tcp", "sq", "sq2"):
        c.update(fan_counters(f"{tag}_{mode}_{p}"))
    if not c:
        continue
    c["hbm_gb_per_launch"] = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024 / 1e9
    c["l2_hit_rate"] = c["TCC_HIT_sum"] / c["TCC_REQ_sum"]
    c["l1_hit_rate"] = 1 - c["TCP_TCC_READ_REQ_sum"] / c["TCP_TOTAL_CACHE_ACCESSES_sum"]
    c["wait_any_over_wave_cycles"] = c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]
    res[key] = c
json.dump(res, open(os.path.join(ROOT, "profiles", f"{rnd}_config2_counters.json"), "w"), indent=1)
st = glob.glob(os.path.join(G, f"{tag}_stats", "*", "*_kernel_stats.csv"))
if st:
    shutil.copy(st[0], os.path.join(ROOT, "profiles", f"{rnd}_config2_kernel_stats.csv"))
print(json.dumps({k: {a: v[a] for a in ("hbm_gb_per_launch", "l2_hit_rate", "l1_hit_rate", "wait_any_over_wave_cycles", "SQ_INSTS_VALU")}
                  for k, v in res.items() if isinstance(v, dict)}, indent=1))
