"""cProfile of pr.shoot_rays for a 1e6-ray end-state-sized fan (S = 2): where the non-kernel time goes."""
import sys, os, time, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import pygenray_amd as pr
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rmax = 1000e3
z = np.arange(0, 6000, 1.0); r = np.linspace(0, rmax, 100)
env = pr.OceanEnvironment2D(pr.DataArray(np.tile(pr.munk_ssp(z), (100, 1)), dims=["range", "depth"], coords={"range": r, "depth": z}),
                            pr.DataArray(np.full(100, 5000.0), dims=["range"], coords={"range": r}), flat_earth_transform=False)
angles = np.linspace(-20, 20, n)
for mode in (False, True):
    pr.shoot_rays(1000.0, 0.0, angles, rmax, S, env, debug=False, flatearth=False, device_resident=mode)
    p = cProfile.Profile(); p.enable(); t0 = time.perf_counter()
    fan = pr.shoot_rays(1000.0, 0.0, angles, rmax, S, env, debug=False, flatearth=False, device_resident=mode)
    dt = time.perf_counter() - t0; p.disable()
    print(f"shoot_rays({n}, S={S}, device_resident={mode}): {dt*1e3:.1f} ms")
    pstats.Stats(p).sort_stats("tottime").print_stats(12)
    del fan
