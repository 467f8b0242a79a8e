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
# ---- round 5: the 1e6-ray leg (persistent waves), the lone steepest wave, pr.shoot_rays on configs[2]
def all_counters(d, needle, n_last=1):
    """the counter of directory d summed over the LAST n_last dispatches of the kernels whose name holds `needle`"""
    f = glob.glob(os.path.join(G, d, "*", "*_counter_collection.csv"))
    if not f:
        return None
    per = {}
    for r in csv.DictReader(open(f[0])):
        if needle in r["Kernel_Name"]:
            per.setdefault(r["Dispatch_Id"], 0.0)
            per[r["Dispatch_Id"]] += float(r["Counter_Value"])
    return sum(per[k] for k in sorted(per, key=int)[-n_last:]) if per else None


fe, wr = fan_counters(f"{tag}_1e6_FETCH_SIZE").get("FETCH_SIZE"), fan_counters(f"{tag}_1e6_WRITE_SIZE").get("WRITE_SIZE")
if fe is not None and wr is not None:
    out["sample-nosave@1000000"] = {"rays": 1000000, "FETCH_SIZE_KB": fe, "WRITE_SIZE_KB": wr, "hbm_gb_per_launch": (2 * fe + wr) * 1024 / 1e9,
                                    "note": f"profiles/{rnd}_traffic.json: rocprofv3 PMC passes (FETCH_SIZE x 2 per the gfx950 correction + WRITE_SIZE), "
                                            "fan kernel pgr_fan_kernel<true, 4, 0, true> (persistent waves), last dispatch",
                                    "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python3 bench.py --rays 1000000 "
                                               "--no-save --steps 2 --warmup 1 --no-cpu-baseline --no-eigenray --no-legs"}
    sq = fan_counters(f"{tag}_1e6_sq")
    if sq:
        out["sample-nosave@1000000-sq_counters"] = sq
        out["sample-nosave@1000000"]["valu_wave_instructions_per_launch"] = sq.get("SQ_INSTS_VALU")
# ---- round 6: the 1e6-ray fan WITH S = 1001 trajectories (pgr_fan_kernel<true, 4, 1, true>, 24 GB of samples)
fe, wr = fan_counters(f"{tag}_1e6traj_FETCH_SIZE").get("FETCH_SIZE"), fan_counters(f"{tag}_1e6traj_WRITE_SIZE").get("WRITE_SIZE")
if fe is not None and wr is not None:
    out["sample@1000000"] = {"rays": 1000000, "FETCH_SIZE_KB": fe, "WRITE_SIZE_KB": wr, "hbm_gb_per_launch": (2 * fe + wr) * 1024 / 1e9,
                             "sample_gb": 1000000 * 1001 * 24 / 1e9,
                             "note": f"profiles/{rnd}_traffic.json: rocprofv3 PMC passes (FETCH_SIZE x 2 per the gfx950 correction + WRITE_SIZE), "
                                     "fan kernel pgr_fan_kernel<true, 4, 1, true> (persistent waves, S = 1001 trajectories), last dispatch",
                             "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python3 bench.py --rays 1000000 "
                                        "--steps 2 --warmup 1 --no-cpu-baseline --no-eigenray --no-legs"}
    sq = fan_counters(f"{tag}_1e6traj_sq")
    if sq:
        out["sample@1000000-sq_counters"] = sq
        out["sample@1000000"]["valu_wave_instructions_per_launch"] = sq.get("SQ_INSTS_VALU")
for what, name in (("1e6", "1e6"), ("1e6traj", "1e6_traj"), ("nosave", "end_state"), ("lone", "lone_wave"), ("api2", "api_config2")):
    st = glob.glob(os.path.join(G, f"{tag}_{what}_stats", "*", "*_kernel_stats.csv"))
    if st:
        shutil.copy(st[0], os.path.join(ROOT, "profiles", f"{rnd}_kernel_stats_{name}.csv"))
api = {}
for mode in ("blocked", "rows"):
    row = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for kern, needle in (("fan_kernel", "pgr_fan_kernel"), ("unblock_or_gather", "cols")):
            v = all_counters(f"{tag}_api2_{mode}_{c}", needle, 1 if kern == "fan_kernel" else 3)   # (T, z, p: three passes)
            if v is not None:
                row[f"{kern}_{c}_KB"] = v
        lg = os.path.join(G, f"{tag}_api2_{mode}_{c}.log")
        if os.path.exists(lg):
            for ln in open(lg):
                if ln.startswith("{"):
                    row.update({k: v for k, v in json.loads(ln).items() if k in ("rays_kept", "sample_bytes")})
    if row:
        if "fan_kernel_WRITE_SIZE_KB" in row and "sample_bytes" in row:
            # (the fan kernel writes the samples of ALL 1e5 rays, dropped ones as NaN columns; the host gets the survivors')
            row["fan_kernel_write_over_sample_bytes"] = row["fan_kernel_WRITE_SIZE_KB"] * 1024 / (100000 * 1001 * 24)
        api[mode] = row
if api:
    api["command"] = "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python3 scripts/api_cfg2_run.py blocked|rows"
    api["note"] = ("pr.shoot_rays on configs[2] (1e5 rays, S = 1001, eager): the LAST dispatch of the fan kernel and of the pass that squeezes dropped "
                   "rays out (pgr_unblock_cols / pgr_gather_cols), each array's pass summed; blocked = PGR_OPT_API_BLOCKED 1 (default)")
    out["api-config2"] = api
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
