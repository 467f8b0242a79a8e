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
# sloping sea floor on non-uniform ranges, range-dependent, flat earth: the bottom locator with a slope
import pygenray_amd as pr
from pygenray_amd.environment import _unpack_envi
rmax = 1000e3; z = np.arange(0, 6000, 1.0); r = np.linspace(0, rmax, 101); rng = np.random.default_rng(0)
br = np.sort(np.concatenate([[0, rmax], rng.uniform(0, rmax, 199)]))
ssp = pr.DataArray(np.array([pr.munk_ssp(z, 1300 + 2e-4 * ri) for ri in r]), dims=["range", "depth"], coords={"range": r, "depth": z})
e2 = pr.OceanEnvironment2D(ssp, pr.DataArray(4350 + 850 * np.sin(br / 150e3), dims=["range"], coords={"range": br}), flat_earth_transform=True)
arrs2 = _unpack_envi(e2, flatearth=True); env2 = _lib.EnvHandle(*arrs2)
theta = np.linspace(-20, 20, 6400); y0 = fan_y0(arrs2, 1000.0, 0.0, -theta)
fan = DeviceFan(env2, y0, 0.0, rmax, 2, save=False); fan.run(); torch.cuda.synchronize()
nb = fan.n_bott.cpu().numpy(); ns = fan.n_surf.cpu().numpy()
fan.flags |= 16; fan.run(); torch.cuda.synchronize()
d = fan.n_rej.cpu().numpy().reshape(-1, 64)[:, 2:]
print(f"sloping bottom: bounces/ray {float((nb + ns).mean()):.1f}; exact-bisection fallbacks per ray {d.mean():.3f} (max {d.max()})")
