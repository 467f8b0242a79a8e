"""profiles/<round>_isa_budget.json from the PMC passes of scripts/collect_isa_budget.sh: dynamic instruction counts
of the fan kernels, per launch and per unit of work.

Each fan is counted end state only (SAVE = 0) and with trajectories (SAVE = 1; configs[2] also SAVE = 3, the
sample-blocked layout); `quiet` = the 60 000 rays of the fan that never touch a boundary (|theta| <= 12 deg: attempts
and samples only, one `service` per wave = its start-up), `full` = all 1e5 rays, `steep` = the 25 000 rays with
|theta| >= 15 deg (40 ... 64 bounces each).  Derived:
  per_wave_trip       = counters(quiet, SAVE 0) / wave-trips           -> one step ATTEMPT of a wave (64 lanes)
  per_sample_row      = (counters(quiet, SAVE s) - counters(quiet, SAVE 0)) / (saved samples / 64)
                                                                       -> evaluating + storing 64 samples
  per_service         = (counters(full, SAVE 0) - per_wave_trip x wave-trips(full)) / services(full)
                                                                       -> one bounce SERVICE phase of a wave
usage: python scripts/summarize_isa_budget.py <tag> <round>"""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, rnd = sys.argv[1], sys.argv[2]
G = os.path.join(ROOT, "gpurun_out")
NAMES = ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_WR", "SQ_INSTS_VMEM_RD", "SQ_INSTS_BRANCH",
         "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAVES")


def counters(d):
    out = {}
    for dd in (d, d + "_b"):
        f = glob.glob(os.path.join(G, dd, "*", "*_counter_collection.csv"))
        if not f:
            continue
        per = {}
        for r in csv.DictReader(open(f[0])):
            if "pgr_fan_kernel" in r["Kernel_Name"]:
                per.setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], [0.0, r["Kernel_Name"]])
                per[r["Counter_Name"]][r["Dispatch_Id"]][0] += float(r["Counter_Value"])
        for k, v in per.items():
            last = sorted(v, key=int)[-1]
            out[k] = v[last][0]
            out["kernel"] = v[last][1].split("(")[0]
    return out


def meta(k):
    p = os.path.join(G, f"{tag}_{k}.log")
    for l in open(p):
        if l.startswith("{"):
            return json.loads(l)
    return None


runs = {}
for p in sorted(glob.glob(os.path.join(G, f"{tag}_*_?.log"))):
    k = os.path.basename(p)[len(tag) + 1:-4]
    m = meta(k)
    c = counters(f"{tag}_{k}")
    if m and c:
        runs[k] = dict(m, **c)
out = {"_doc": __doc__.split("usage:")[0].strip(), "runs": runs, "derived": {}}
for wl in ("headline", "rangedep", "flatearth"):
    q0, f0 = runs.get(f"{wl}_quiet_0"), runs.get(f"{wl}_full_0")
    if not q0 or not f0:
        continue
    d = {}
    per_trip = {n: q0[n] / q0["wave_trips"] for n in NAMES if n in q0 and n != "SQ_WAVES"}
    d["per_wave_trip"] = dict(per_trip, lane_utilisation=(q0["accepted_steps"] + q0["rejected_attempts"]) / (64.0 * q0["wave_trips"]))
    for s in (1, 3):
        qs = runs.get(f"{wl}_quiet_{s}")
        if qs:
            rows = qs["saved_samples"] / 64.0
            d[f"per_sample_row_SAVE{s}"] = {n: (qs[n] - q0[n]) / rows for n in NAMES if n in qs and n in q0 and n != "SQ_WAVES"}
            # (the SAVE != 0 kernels are other instances: their attempt differs by a few instructions -- pinned coefficients)
    d["per_service"] = dict({n: (f0[n] - per_trip[n] * f0["wave_trips"]) / f0["services"] for n in per_trip},
                            services_per_launch=f0["services"], bounces_per_service=f0["bounces"] / max(f0["services"], 1))
    for s in (0, 1, 3):
        fs = runs.get(f"{wl}_full_{s}")
        if fs:
            d[f"launch_SAVE{s}"] = {"SQ_INSTS_VALU": fs["SQ_INSTS_VALU"], "SQ_INSTS_SALU": fs["SQ_INSTS_SALU"], "wave_trips": fs["wave_trips"],
                                    "services": fs["services"], "accepted_steps": fs["accepted_steps"],
                                    "valu_per_accepted_step_lane": fs["SQ_INSTS_VALU"] * 64.0 / fs["accepted_steps"],
                                    "share_attempts": per_trip["SQ_INSTS_VALU"] * fs["wave_trips"] / fs["SQ_INSTS_VALU"],
                                    "kernel": fs.get("kernel")}
            if s == 0:
                d["launch_SAVE0"]["share_services"] = 1.0 - d["launch_SAVE0"]["share_attempts"]
    st = runs.get("headline_steep_0") if wl == "headline" else None
    if st:
        d["steep_fan_SAVE0"] = {"lane_utilisation": (st["accepted_steps"] + st["rejected_attempts"]) / (64.0 * st["wave_trips"]),
                                "wave_trips": st["wave_trips"], "services": st["services"], "bounces": st["bounces"],
                                "SQ_INSTS_VALU": st["SQ_INSTS_VALU"]}
    out["derived"][wl] = d
b = os.path.join(G, f"{tag}_binary.json")
if os.path.exists(b):
    out.update(json.load(open(b)))
json.dump(out, open(os.path.join(ROOT, "profiles", f"{rnd}_isa_budget.json"), "w"), indent=1)
print(json.dumps(out["derived"], indent=1))
