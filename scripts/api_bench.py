"""Wall-clock of the drop-in API (host buffers, PCIe and NumPy included) on the headline fan."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import pygenray_amd as pr
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1001
rmax = 1000e3
z = np.arange(0, 6000, 1.0); r = np.linspace(0, rmax, 100)
env = pr.OceanEnvironment2D(pr.DataArray(np.tile(pr.munk_ssp(z), (100, 1)), dims=["range", "depth"], coords={"range": r, "depth": z}),
                            pr.DataArray(np.full(100, 5000.0), dims=["range"], coords={"range": r}), flat_earth_transform=False)
angles = np.linspace(-20, 20, n)
pr.shoot_rays(1000.0, 0.0, angles[:1000], rmax, S, env, debug=False, flatearth=False)  # warm up, table upload
for mode in (False, True):
    for k in range(3):
        t0 = time.perf_counter()
        fan = pr.shoot_rays(1000.0, 0.0, angles, rmax, S, env, debug=False, flatearth=False, device_resident=mode)
        dt = time.perf_counter() - t0
        t1 = time.perf_counter()
        z_end = fan.zs_end[0]
        zs = fan.zs
        dt2 = time.perf_counter() - t1
        print(f"shoot_rays({n} rays, S={S}, device_resident={mode}): {dt*1e3:.1f} ms wall, {len(fan)} rays kept; then fan.zs "
              f"{dt2*1e3:.1f} ms, zs[0,-1]={zs[0,-1]:.6f} (= zs_end {z_end:.6f})", flush=True)
        del fan, zs
