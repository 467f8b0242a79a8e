import os
"""A 'realistic' environment: range-dependent SSP, flat-earth transform, sloping sea floor given on
(non-)uniform ranges.  Times the fan kernel for the combinations."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import pygenray_amd as pr
from pygenray_amd import _lib
if os.environ.get('PGR_LIB'):   # A/B against another build
    _lib.LIB_PATH = os.path.abspath(os.environ['PGR_LIB'])
from pygenray_amd.environment import _unpack_envi
from pygenray_amd.device_fan import DeviceFan, fan_y0
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
rmax = 1000e3
z = np.arange(0, 6000, 1.0); r = np.linspace(0, rmax, 101)
rng = np.random.default_rng(0)
for label, br in (("flat 5000 m, uniform ranges", None),
                  ("slope 3500-5200 m, uniform ranges", np.linspace(0, rmax, 201)),
                  ("slope 3500-5200 m, NON-uniform ranges", np.sort(np.concatenate([[0, rmax], rng.uniform(0, rmax, 199)])))):
    ssp = pr.DataArray(np.array([pr.munk_ssp(z, 1300 + 2e-4 * ri) for ri in r]), dims=["range", "depth"], coords={"range": r, "depth": z})
    if br is None:
        bathy = pr.DataArray(np.full(101, 5000.0), dims=["range"], coords={"range": r})
    else:
        bathy = pr.DataArray(4350 + 850 * np.sin(br / 150e3), dims=["range"], coords={"range": br})
    env = pr.OceanEnvironment2D(ssp, bathy, flat_earth_transform=True)
    arrs = _unpack_envi(env, flatearth=True)
    h = _lib.EnvHandle(*arrs)
    y0 = fan_y0(arrs, 1000.0, 0.0, -np.linspace(-20, 20, n))
    fan = DeviceFan(h, y0, 0.0, rmax, 1001, save=False, sample_major=True)
    fan.run(); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fan.run(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    print(f"{label:42s}: kernel {min(ts):7.2f} ms, {fan.ray_steps()/min(ts)/1e6:6.2f} G ray-steps/s, dropped {(fan.status != 0).sum().item()}, bounces/ray {float((fan.n_bott + fan.n_surf).float().mean()):.1f}", flush=True)
    del fan
