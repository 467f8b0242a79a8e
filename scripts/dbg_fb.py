"""How often does the exact event bisection still run?  (PGR_DEBUG_TRIPS: lanes 2..63 report it)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from helpers import munk_arrays
from pygenray_amd import _lib
from pygenray_amd.device_fan import DeviceFan, fan_y0
arrs = munk_arrays(1000e3); env = _lib.EnvHandle(*arrs)
for lo, hi in ((19.7, 19.8), (14.0, 14.1)):
    theta = np.linspace(lo, hi, 64); y0 = fan_y0(arrs, 1000.0, 0.0, -theta)
    fan = DeviceFan(env, y0, 0.0, 1000e3, 2, save=False); fan.flags |= 16; fan.run(); torch.cuda.synchronize()
    d = fan.n_rej.cpu().numpy()
    print(f"[{lo},{hi}] deg: bounces/lane {float((fan.n_bott+fan.n_surf).float().mean()):.1f}, trips {d[0]}, services {d[1]}, exact-bisection fallbacks/lane {d[2:].mean():.2f}")
