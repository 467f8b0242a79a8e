import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import numpy as np, torch
import pygenray_amd as pr
from pygenray_amd import _lib
from pygenray_amd.environment import _unpack_envi
from pygenray_amd.device_fan import DeviceFan, fan_y0
n = 100000; rmax = 1000e3
z = np.arange(0, 6000, 1.0); r = np.linspace(0, rmax, 100)
ssp = pr.DataArray(np.array([pr.munk_ssp(z, 1300.0) for ri in r]), dims=["range", "depth"], coords={"range": r, "depth": z})
bathy = pr.DataArray(np.full(100, 5000.0), dims=["range"], coords={"range": r})
env = pr.OceanEnvironment2D(ssp, bathy, flat_earth_transform=True)
for fe in (False, True):
    arrs = _unpack_envi(env, flatearth=fe)
    h = _lib.EnvHandle(*arrs)
    y0 = fan_y0(arrs, 1000.0, 0.0, -np.linspace(-20, 20, n))
    for S, save in ((2, False), (2, True), (11, True), (101, True), (1001, True)):
        fan = DeviceFan(h, y0, 0.0, rmax, S, save=save, sample_major=True)
        for _ in range(3): fan.run()
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); fan.run(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        print(f"flatearth={fe} S={S} save={save}: kernel {min(ts):.3f} ms (query zm-cubic {h.query(5)})", flush=True)
        del fan
