import os
"""Kernel time with the reference's DEFAULT environment handling: flat-earth transformed tables
(non-uniform zin -> the generic depth-cell search) against the untransformed uniform grid."""
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
z = np.arange(0, 6000, 1.0); r = np.linspace(0, rmax, 100)
slope = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0   # > 0: range-dependent sound speed (table stays in HBM)
ssp = pr.DataArray(np.array([pr.munk_ssp(z, 1300 + slope * ri) for ri in r]), dims=["range", "depth"], coords={"range": r, "depth": z})
bathy = pr.DataArray(np.full(100, 5000.0), dims=["range"], coords={"range": r})
env = pr.OceanEnvironment2D(ssp, bathy, flat_earth_transform=True)
for fe in (False, True):
    arrs = _unpack_envi(env, flatearth=fe)
    h = _lib.EnvHandle(*arrs)
    if os.environ.get('PGR_DEPTH_SEARCH'):   # 3 = quadratic estimate + three nodes (round 2's kernel), 2 = bucket table, 1 = binary search
        h.set_option("depth_search", int(os.environ['PGR_DEPTH_SEARCH']))
    y0 = fan_y0(arrs, 1000.0, 0.0, -np.linspace(-20, 20, n))
    for save in (False, True):
        fan = DeviceFan(h, y0, 0.0, rmax, 1001, save=save, sample_major=True)
        fan.run(); torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); fan.run(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        print(f"flatearth={fe} save={save}: kernel {min(ts):.2f} ms, {fan.ray_steps()/min(ts)/1e6:.2f} G ray-steps/s, dropped {(fan.status != 0).sum().item()}", flush=True)
        del fan
