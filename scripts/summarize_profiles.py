"""Turn the rocprofv3 outputs of scripts/collect_profiles.sh (merged back under gpurun_out/) into
the committed summaries: profiles/<round>_kernel_stats*.csv and profiles/<round>_traffic.json -- which names the
BINARY the counters describe (sha256 of the gfx950 machine code, pgr_build_info(), git commit), so that bench.py
reports them only for that code.
usage: python scripts/summarize_profiles.py <tag> <round>   e.g.  prof4 r03"""
import csv, glob, json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, rnd = sys.argv[1], sys.argv[2]
G = os.path.join(ROOT, "gpurun_out")


def fan_counters(d):
    f = glob.glob(os.path.join(G, d, "*", "*_counter_collection.csv"))
    if not f:
        return {}
    per = {}
    for r in csv.DictReader(open(f[0])):
        if "pgr_fan_kernel" in r["Kernel_Name"]:
            per.setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
            per[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    # last dispatch of the fan kernel (a timed step, not the warm-up)
    return {k: v[sorted(v, key=int)[-1]] for k, v in per.items()}


out = {}
for variant, flag in (("", ""), ("flatearth", " --flat-earth"), ("rangedep", " --range-dependent")):
    P = variant + "_" if variant else ""
    K = variant + "-" if variant else ""
    for key, pre in ((K + "sample", ""), (K + "sample-nosave", "nosave_")):
        fe = fan_counters(f"{tag}_{P}{pre}FETCH_SIZE").get("FETCH_SIZE")
        wr = fan_counters(f"{tag}_{P}{pre}WRITE_SIZE").get("WRITE_SIZE")
        if fe is None or wr is None:
            continue
        # MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are in KB; gfx950 FETCH_SIZE counts half the bytes
        out[key] = {"rays": 100000, "FETCH_SIZE_KB": fe, "WRITE_SIZE_KB": wr,
                    "hbm_gb_per_launch": (2 * fe + wr) * 1024 / 1e9,
                    "note": f"profiles/{rnd}_traffic.json: rocprofv3 PMC passes of this command (FETCH_SIZE x 2 per the gfx950 "
                            "correction + WRITE_SIZE, fan kernel, last dispatch)",
                    "command": f"rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- "
                               f"python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-eigenray --no-legs{flag}{' --no-save' if pre else ''}"}
    sq = fan_counters(f"{tag}_{P}sq")
    if sq:
        out[K + "sq_counters_sample"] = sq
        if K + "sample" in out and "SQ_INSTS_VALU" in sq:
            out[K + "sample"]["valu_wave_instructions_per_launch"] = sq["SQ_INSTS_VALU"]
    st = glob.glob(os.path.join(G, f"{tag}_{P}stats", "*", "*_kernel_stats.csv"))
    if st:
        shutil.copy(st[0], os.path.join(ROOT, "profiles", f"{rnd}_kernel_stats{'_' + variant if variant else ''}.csv"))
        if not variant:
            print(open(st[0]).read())
# configs[2] with trajectories in the sample-blocked layout
fe = fan_counters(f"{tag}_rangedep_blocked_FETCH_SIZE").get("FETCH_SIZE")
wr = fan_counters(f"{tag}_rangedep_blocked_WRITE_SIZE").get("WRITE_SIZE")
if fe is not None and wr is not None:
    out["rangedep-blocked"] = {"rays": 100000, "FETCH_SIZE_KB": fe, "WRITE_SIZE_KB": wr, "hbm_gb_per_launch": (2 * fe + wr) * 1024 / 1e9,
                               "note": f"profiles/{rnd}_traffic.json: rocprofv3 PMC passes (FETCH_SIZE x 2 per the gfx950 correction + WRITE_SIZE), "
                                       "fan kernel pgr_fan_kernel<false, 4, 3>, last dispatch",
                               "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python3 bench.py --steps 3 "
                                          "--warmup 1 --no-cpu-baseline --no-eigenray --no-legs --range-dependent --blocked"}
    sq = fan_counters(f"{tag}_rangedep_blocked_sq")
    if sq:
        out["rangedep-blocked-sq_counters"] = sq
        out["rangedep-blocked"]["valu_wave_instructions_per_launch"] = sq.get("SQ_INSTS_VALU")
st = glob.glob(os.path.join(G, f"{tag}_rangedep_blocked_stats", "*", "*_kernel_stats.csv"))
if st:
    shutil.copy(st[0], os.path.join(ROOT, "profiles", f"{rnd}_kernel_stats_rangedep_blocked.csv"))
b = os.path.join(G, f"{tag}_binary.json")
if os.path.exists(b):
    out.update(json.load(open(b)))
try:
    out["git_commit"] = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "HEAD"], text=True).strip()
    out["git_dirty"] = bool(subprocess.check_output(["git", "-C", ROOT, "status", "--porcelain", "--", "pygenray_amd", "include"], text=True).strip())
except Exception:
    pass
out["_note"] = (f"{rnd}: fan kernel, last dispatch of each pass; device_code_sha256 = sha256 of the gfx950 .text the passes ran "
                "(pygenray_amd._lib.device_code_sha256): bench.py reports these counters only when the loaded library has the same")
json.dump(out, open(os.path.join(ROOT, "profiles", f"{rnd}_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
