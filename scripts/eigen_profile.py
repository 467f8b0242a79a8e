import sys, os, time, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import pygenray_amd as pr
n = 1_000_000; rmax = 1000e3
z = np.arange(0, 6000, 1.0); r = np.linspace(0, rmax, 100)
env = pr.OceanEnvironment2D(pr.DataArray(np.tile(pr.munk_ssp(z), (100, 1)), dims=["range", "depth"], coords={"range": r, "depth": z}),
                            pr.DataArray(np.full(100, 5000.0), dims=["range"], coords={"range": r}), flat_earth_transform=False)
angles = np.linspace(-20, 20, n)
fan = pr.shoot_rays(1000.0, 0.0, angles[:1000], rmax, 2, env, debug=False, flatearth=False)  # warm up
def work():
    fan = pr.shoot_rays(1000.0, 0.0, angles, rmax, 2, env, debug=False, flatearth=False)
    import io, contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        er = pr.find_eigenrays(fan, [1000.0], 1000.0, 0.0, rmax, 2, env, ztol=1, max_iter=20, debug=False, flatearth=False)
    return er
pr_ = cProfile.Profile(); pr_.enable(); t0 = time.time(); er = work(); dt = time.time() - t0; pr_.disable()
print("total", dt)
pstats.Stats(pr_).sort_stats("cumulative").print_stats(18)
