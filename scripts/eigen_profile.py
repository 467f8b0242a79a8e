"""Where the host time of BASELINE configs[3] goes (1e6-angle fan + eigenray search through the Python API):
cProfile of the second run (everything warm).  usage: python scripts/eigen_profile.py [n_angles]"""
import sys, os, time, cProfile, pstats, io
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


def run():
    fan = pr.shoot_rays(1000.0, 0.0, angles, rmax, 2, env, debug=False, flatearth=False)
    return pr.find_eigenrays(fan, [1000.0], 1000.0, 0.0, rmax, 2, env, ztol=1, max_iter=20, debug=False, flatearth=False)


run()
t0 = time.time(); run(); print(f"warm run: {time.time() - t0:.3f} s")
pr_ = cProfile.Profile(); pr_.enable(); run(); pr_.disable()
st = io.StringIO(); pstats.Stats(pr_, stream=st).sort_stats("cumulative").print_stats(35); print(st.getvalue())
