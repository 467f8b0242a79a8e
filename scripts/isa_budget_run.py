"""One fan under rocprofv3 --pmc for the instruction budget (scripts/collect_isa_budget.sh): a workload, a kernel
instance, ONE counted launch; prints a JSON line with the fan's wave-trip / service / sample counts (PGR_DEBUG_TRIPS)
that the counters are divided by.
usage: isa_budget_run.py <workload: headline|rangedep|flatearth> <fan: full|quiet|steep> <save: 0|1|3>"""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from pygenray_amd import _lib
from pygenray_amd.device_fan import DeviceFan, fan_y0
from pygenray_amd.environment import _unpack_envi
wl, fanq, save = sys.argv[1], sys.argv[2], int(sys.argv[3])
env_obj, arrs = bench.munk_tables(1000e3, nr=101, sofar_slope=2e-4) if wl == "rangedep" else bench.munk_tables(1000e3)
if wl == "flatearth":
    env_obj.flat_earth_transform(lat=35)
    arrs = _unpack_envi(env_obj, flatearth=True)
theta = np.linspace(-20, 20, 100_000)
if fanq == "quiet":
    theta = theta[np.abs(theta) <= 12.0]          # no ray of these touches a boundary: attempts (+ samples) only
elif fanq == "steep":
    theta = theta[np.abs(theta) >= 15.0]          # every ray bounces, 40 ... 64 times
env = _lib.EnvHandle(*arrs)
y0 = fan_y0(arrs, 1000.0, 0.0, -theta)
fan = DeviceFan(env, y0, 0.0, 1000e3, 1001, save=(save != 0), sample_major=True, sample_blocked=(save == 3))
# the counted launch is the LAST fan-kernel dispatch of the process: first a PGR_DEBUG_TRIPS pass for the denominators
fan.flags |= 16
fan.run(); torch.cuda.synchronize()
d = fan.n_rej.cpu().numpy().astype(np.int64)
nw = (len(theta) + 63) // 64
d = np.pad(d, (0, nw * 64 - len(theta))).reshape(nw, 64)
trips, services, fallbacks = int(d[:, 0].sum()), int(d[:, 1].sum()), int(d[:, 2].sum())
fan.flags &= ~16
fan.run(); torch.cuda.synchronize()
ok = (fan.status == 0)
nst = fan.n_steps.to(torch.int64)
print(json.dumps({"workload": wl, "fan": fanq, "save": save, "rays": len(theta), "waves": nw, "wave_trips": trips, "services": services,
                  "exact_bisection_fallbacks": fallbacks, "accepted_steps": int(nst[ok].sum().item()), "rejected_attempts": int(fan.n_rej.to(torch.int64)[ok].sum().item()),
                  "bounces": int((fan.n_bott + fan.n_surf).to(torch.int64)[ok].sum().item()),
                  "saved_samples": int(ok.sum().item()) * 1001 if save else 0}))
