"""BASELINE configs[3]: eigenray search, fixed source/receiver, 1e6-angle fan + regula falsi."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import pygenray_amd as pr
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
rmax = 1000e3
z = np.arange(0, 6000, 1.0); r = np.linspace(0, rmax, 100)
env = pr.OceanEnvironment2D(pr.DataArray(np.tile(pr.munk_ssp(z), (100, 1)), dims=["range", "depth"], coords={"range": r, "depth": z}),
                            pr.DataArray(np.full(100, 5000.0), dims=["range"], coords={"range": r}), flat_earth_transform=False)
angles = np.linspace(-20, 20, n)
t0 = time.time()
fan = pr.shoot_rays(1000.0, 0.0, angles, rmax, 2, env, debug=False, flatearth=False)
t1 = time.time()
er = pr.find_eigenrays(fan, [1000.0], 1000.0, 0.0, rmax, 2, env, ztol=1, max_iter=20, debug=False, flatearth=False)
t2 = time.time()
print(f"fan of {n} angles (S=2): {t1-t0:.3f} s; eigenray search: {t2-t1:.3f} s; brackets {er.num_eigenrays[1000.0]}, found {er.num_eigenrays_found[0]}, failed {len(er.failed_eray_theta_brackets[0])}")
t0 = time.time()
fan = pr.shoot_rays(1000.0, 0.0, angles, rmax, 2, env, debug=False, flatearth=False)
er = pr.find_eigenrays(fan, [1000.0], 1000.0, 0.0, rmax, 2, env, ztol=1, max_iter=20, debug=False, flatearth=False)
print(f"second run total: {time.time()-t0:.3f} s")
print("arrival times (first 10):", np.sort(er.ts[0][:, -1])[:10])
