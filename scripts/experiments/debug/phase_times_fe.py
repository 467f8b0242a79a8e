"""phase_times.py for the flat-earth (ZM = 5) and the uniform-grid (ZM = 4) trajectory kernels at several S: s_memtime ticks
per trip between the attempt's stamps (library built with -DPGR_TIMING: scripts/build_variants.py timing)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import pygenray_amd as pr
from pygenray_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "scripts/ab/timing.so")
from pygenray_amd.environment import _unpack_envi
from pygenray_amd.device_fan import DeviceFan, fan_y0
NAMES = ["0 gate/loop top", "1 min_step, t_new", "2 step_weights", "3 sum -> z2", "4 look-up+rhs 2", "5 sum -> z3",
         "6 look-up+rhs 3", "7 sum -> z4", "8 look-up+rhs 4", "9 sum -> z5", "10 look-up+rhs 5", "11 sum -> z6",
         "12 look-up+rhs 6", "13 sum -> y_new", "14 look-up+rhs 7", "15 error norm", "16 controller",
         "17 events", "18 commit/samples", "19 (rejected: skip)"]
rmax = 1000e3
z = np.arange(0, 6000, 1.0); r = np.linspace(0, rmax, 100)
ssp = pr.DataArray(np.array([pr.munk_ssp(z, 1300.0) for ri in r]), dims=["range", "depth"], coords={"range": r, "depth": z})
env = pr.OceanEnvironment2D(ssp, pr.DataArray(np.full(100, 5000.0), dims=["range"], coords={"range": r}), flat_earth_transform=True)
rows = {}
for fe in (False, True):
    arrs = _unpack_envi(env, flatearth=fe)
    h = _lib.EnvHandle(*arrs)
    y0 = fan_y0(arrs, 1000.0, 0.0, -np.linspace(-20, -19.9748, 64))
    for S in (2, 11, 101, 1001):
        fan = DeviceFan(h, y0, 0.0, rmax, S, save=True, sample_major=True); fan.run(); torch.cuda.synchronize()
        acc = fan.n_rej.cpu().numpy()[:24].astype(np.int64) & 0xffffffff
        fan.flags |= 16; fan.run(); torch.cuda.synchronize()
        trips = int(fan.n_rej.cpu().numpy()[0])
        rows[(fe, S)] = acc[:20] / trips
        print(f"flatearth={fe} S={S}: {trips} trips, {acc[:20].sum() / trips:.0f} ticks per trip; commit/samples {acc[18] / trips:.1f}, events {acc[17] / trips:.1f}, gate {acc[0] / trips:.1f}", flush=True)
for k, nm in enumerate(NAMES):
    print(f"   {nm:24s}" + "".join(f" {rows[(fe, S)][k]:8.1f}" for fe in (False, True) for S in (2, 1001)))
