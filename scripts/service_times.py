"""Where one bounce SERVICE of a lone wave spends its cycles (library built with -DPGR_SVC_TIMING:
`python scripts/build_variants.py svctiming`; s_memtime stamps between the service's sections, accumulated per wave and
handed back in n_rej of lanes 0..23).  usage (GPU box): python scripts/service_times.py [scripts/ab/svctiming.so] [--out x.json]"""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from helpers import munk_arrays
from pygenray_amd import _lib
args = [a for a in sys.argv[1:] if not a.startswith("--")]
_lib.LIB_PATH = os.path.abspath(args[0] if args else os.path.join(ROOT, "scripts", "ab", "svctiming.so"))
from pygenray_amd.device_fan import DeviceFan, fan_y0

NAMES = ["0 entry: descriptor / argument loads", "1 Q = K.T P", "2 bracket values, bathymetry at the ends", "3 Newton",
         "4 band + true event at its edges", "5 doubles inside a narrow band", "6 replay phase 1", "7 replay phase 2",
         "8 (exact bisection)", "9 samples of the truncated step", "10 state at the root", "11 look-up + arcsine",
         "12 reflection law", "13 new slowness (sine)", "14 restart: rhs 1, d0, d1, h0", "15 restart: rhs 2, d2, power",
         "16 restart: events, nearest save point", "17 exit"]
arrs = munk_arrays(1000e3); env = _lib.EnvHandle(*arrs)
theta = np.linspace(-20, -19.9748, 64); y0 = fan_y0(arrs, 1000.0, 0.0, -theta)
out = {}
for save in (False, True):
    fan = DeviceFan(env, y0, 0.0, 1000e3, 1001 if save else 2, save=save, sample_major=True); fan.run(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); fan.run(); e1.record(); torch.cuda.synchronize()
    acc = fan.n_rej.cpu().numpy()[:24].astype(np.int64) & 0xffffffff
    fan.flags |= 16; fan.run(); torch.cuda.synchronize()
    d = fan.n_rej.cpu().numpy()
    trips, services = int(d[0]), int(d[1])
    svc = acc[:18].sum()
    print(f"save={save}: kernel {e0.elapsed_time(e1):.3f} ms, {trips} trips, {services} services; a service: {svc / services:.0f} cycles "
          f"(stamps included), outside the services: {acc[23] / trips:.0f} cycles per trip")
    for k, nm in enumerate(NAMES):
        print(f"   {nm:44s} {acc[k] / services:9.1f}")
    out["trajectories" if save else "end_state"] = {"kernel_ms": e0.elapsed_time(e1), "trips": trips, "services": services,
                                                     "cycles_per_service": float(svc / services),
                                                     "sections": {nm: float(acc[k] / services) for k, nm in enumerate(NAMES)}}
for a in sys.argv[1:]:
    if a.startswith("--out="):
        json.dump(out, open(a[6:], "w"), indent=1)
