"""When and where every wave of a fan ran (diagnostic build -DPGR_WAVE_TIMES, `scripts/build_variants.py wavetimes`):
each wave leaves its start / end stamps (s_memrealtime, 100 MHz), HW_ID, XCC_ID and workgroup in lanes 3..7 of n_rej
(PGR_DEBUG_TRIPS).  From them: how the launch's wave-slot time splits into running waves, slots idle inside a workgroup
that still holds its CU (a workgroup with the 96 KB LDS table leaves only when its LAST wave ends), slots idle between
workgroups, and the tail of the launch; per-SIMD residency (2 / 1 / 0 waves).

usage (GPU box): python scripts/wave_times.py [--lib scripts/ab/wavetimes.so] [--rays 1000000] [--save] [--out gpurun_out/x.json]
"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from helpers import munk_arrays
from pygenray_amd import _lib
from pygenray_amd.device_fan import DeviceFan, fan_y0

ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=os.path.join(ROOT, "scripts", "ab", "wavetimes.so"))
ap.add_argument("--rays", type=int, default=1_000_000)
ap.add_argument("--save", action="store_true")
ap.add_argument("--slope", type=float, default=0.0)
ap.add_argument("--out", default=None)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--persistent", type=int, default=1, help="PGR_OPT_PERSISTENT: 1 persistent waves + packet queue, 0 static deal of whole workgroups")
a = ap.parse_args()
_lib.LIB_PATH = os.path.abspath(a.lib)
arrs = munk_arrays(1000e3, nr=(101 if a.slope else 100), sofar_slope=a.slope)
env = _lib.EnvHandle(*arrs)
env.set_option("persistent", a.persistent)
n = a.rays
y0 = fan_y0(arrs, 1000.0, 0.0, -np.linspace(-20, 20, n))
fan = DeviceFan(env, y0, 0.0, 1000e3, 1001, save=a.save, sample_major=True)
fan.flags |= 16
res = []
for rep in range(a.reps):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); fan.run(); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    d = fan.n_rej.cpu().numpy().astype(np.int64)
    nw = (n + 63) // 64
    d = np.pad(d, (0, nw * 64 - n)).reshape(nw, 64)
    if n % 64:   # (the last wave's lanes 3..7 may be beyond N: drop it from the statistics)
        d = d[:-1]; nw -= 1
    u32 = lambda v: v & 0xffffffff
    trips, services = d[:, 0], d[:, 1]
    t0, t1 = u32(d[:, 3]), u32(d[:, 4])
    t1 = np.where(t1 < t0, t1 + (1 << 32), t1)
    hw, xcc, wg = u32(d[:, 5]), u32(d[:, 6]) & 0xf, u32(d[:, 7])
    base = t0.min()
    s, e = (t0 - base) * 1e-5, (t1 - base) * 1e-5    # ms (100 MHz ticks)
    span = e.max()
    dur = e - s
    simd = (xcc << 12) | (((hw >> 8) & 0xff) << 2) | ((hw >> 4) & 3)   # (xcc, se/sh/cu, simd)
    cu = simd >> 2
    n_simd, n_cu = len(np.unique(simd)), len(np.unique(cu))
    # workgroups: a workgroup's slots are held from its first wave's start to its last wave's end
    order = np.argsort(wg, kind="stable")
    wg_s, idx = np.unique(wg[order], return_index=True)
    wg_first = np.minimum.reduceat(s[order], idx)
    wg_last = np.maximum.reduceat(e[order], idx)
    wg_waves = np.diff(np.append(idx, len(order)))
    wpb = int(wg_waves.max())
    running = dur.sum()
    if a.persistent and wpb > 8:
        # persistent waves: a workgroup IS its CU for the whole launch and `wg_waves` counts the packets it claimed; the
        # slots are its 8 resident waves, and nothing is "held by a workgroup that waits for its last wave"
        packets_per_cu, wpb = wpb, 8
        held = running
    else:
        packets_per_cu = None
        held = ((wg_last - wg_first) * wpb).sum()                 # slot-ms held by workgroups (empty slots of short workgroups included)
    tail_in_wg = held - running
    slots = n_cu * wpb                                           # wave slots the launch can fill at once (one workgroup per CU)
    total = slots * span
    # per SIMD: time with 0 / 1 / 2+ waves resident (sweep)
    r0 = r1 = r2 = 0.0
    for sid in np.unique(simd):
        m = simd == sid
        ev = np.concatenate([np.stack([s[m], np.ones(m.sum())], 1), np.stack([e[m], -np.ones(m.sum())], 1)])
        ev = ev[np.lexsort((ev[:, 1], ev[:, 0]))]
        t_prev, lvl = 0.0, 0
        for t, dl in ev:
            dt = t - t_prev
            if lvl == 0: r0 += dt
            elif lvl == 1: r1 += dt
            else: r2 += dt
            t_prev, lvl = t, lvl + int(dl)
        r0 += span - t_prev
    # the launch's tail: from the moment the first SIMD runs dry for good (its last wave ended) to the end
    last_end = np.array([e[simd == sid].max() for sid in np.unique(simd)])
    out = {"rays": n, "save": bool(a.save), "persistent": a.persistent, "rep": rep, "kernel_ms_hip_events": ms, "span_ms_stamps": float(span),
           "waves": int(nw), "workgroups": int(len(wg_s)), "waves_per_workgroup": wpb, "max_packets_claimed_by_one_cu": packets_per_cu, "cus_seen": int(n_cu), "simds_seen": int(n_simd),
           "wave_ms": {"min": float(dur.min()), "median": float(np.median(dur)), "max": float(dur.max()), "sum_slot_ms": float(running)},
           "slot_time_split": {"total_slot_ms": float(total), "running_waves": float(running / total),
                               "idle_inside_a_resident_workgroup": float(tail_in_wg / total),
                               "idle_between_workgroups_and_launch_tail": float((total - held) / total)},
           "simd_residency": {"two_or_more_waves": float(r2 / (n_simd * span)), "one_wave": float(r1 / (n_simd * span)), "none": float(r0 / (n_simd * span))},
           "launch_tail": {"first_simd_dry_at_ms": float(last_end.min()), "median_simd_dry_at_ms": float(np.median(last_end)),
                           "mean_dry_before_end_ms": float((span - last_end).mean())},
           "workgroup_spread_ms": {"median_last_minus_first_wave_end": float(np.median(wg_last - np.minimum.reduceat(e[order], idx))),
                                   "p90": float(np.percentile(wg_last - np.minimum.reduceat(e[order], idx), 90)),
                                   "median_workgroup_life": float(np.median(wg_last - wg_first))},
           "trips": {"sum": int(trips.sum()), "services": int(services.sum())}}
    res.append(out)
    print(json.dumps(out), flush=True)
if a.out:
    json.dump(res, open(a.out, "w"), indent=1)
