"""configs[3]'s eigenray search taken apart: the 1e6-angle fan, the device-resident false-position loop, the re-shoot of the
eigenrays found.  usage: eig_ab.py [lib.so]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pygenray_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
import pygenray_amd as pr
from pygenray_amd import eigenrays as er_mod
z = np.arange(0, 6000, 1.0); r = np.linspace(0, 1000e3, 100)
env = pr.OceanEnvironment2D(pr.DataArray(np.tile(pr.munk_ssp(z), (100, 1)), dims=["range", "depth"], coords={"range": r, "depth": z}),
                            pr.DataArray(np.full(100, 5000.0), dims=["range"], coords={"range": r}), flat_earth_transform=False)
ang = np.linspace(-20, 20, 1_000_000)
orig = _lib.EnvHandle.eigen_refine
t_ref = [0.0]
def timed(self, *a, **k):
    t0 = time.perf_counter(); out = orig(self, *a, **k); t_ref[0] = time.perf_counter() - t0; return out
_lib.EnvHandle.eigen_refine = timed
for rep in range(4):
    t0 = time.perf_counter()
    fan = pr.shoot_rays(1000.0, 0.0, ang, 1000e3, 2, env, debug=False, flatearth=False)
    t1 = time.perf_counter()
    er_mod.LAST_SEARCH_STATS.clear()
    er = pr.find_eigenrays(fan, [1000.0], 1000.0, 0.0, 1000e3, 2, env, ztol=1, max_iter=20, debug=False, flatearth=False, quiet=True)
    t2 = time.perf_counter()
    print(f"{os.path.basename(_lib.LIB_PATH)}: fan {1e3 * (t1 - t0):.1f} ms, search {1e3 * (t2 - t1):.1f} ms of which the device loop {1e3 * t_ref[0]:.1f} ms "
          f"({er_mod.LAST_SEARCH_STATS.get('launches')} launches, {er_mod.LAST_SEARCH_STATS.get('trial_rays')} trial rays), found {er.num_eigenrays_found[0]}", flush=True)
import cProfile, pstats
p = cProfile.Profile(); p.enable()
er = pr.find_eigenrays(fan, [1000.0], 1000.0, 0.0, 1000e3, 2, env, ztol=1, max_iter=20, debug=False, flatearth=False, quiet=True)
p.disable()
pstats.Stats(p).sort_stats("tottime").print_stats(14)
