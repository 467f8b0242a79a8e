"""Latency of small drop-in API calls (table upload cached on the environment)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import numpy as np
import pygenray_amd as pr
env = pr.OceanEnvironment2D()   # the reference's default: Munk, flat-earth transform, sloping 4500-4900 m bottom
def t(label, f, reps=3):
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); r = f(); best = min(best, time.perf_counter() - t0)
    print(f"{label:60s} {best*1e3:8.2f} ms", flush=True); return r
t("first shoot_rays (upload + build check)", lambda: pr.shoot_rays(1000.0, 0.0, np.linspace(-15, 15, 100), 90e3, 200, env, debug=False), reps=1)
t("shoot_rays 64 angles, 90 km, S=200", lambda: pr.shoot_rays(1000.0, 0.0, np.linspace(-15, 15, 64), 90e3, 200, env, debug=False))
t("shoot_rays 10 000 angles, 90 km, S=200", lambda: pr.shoot_rays(1000.0, 0.0, np.linspace(-15, 15, 10000), 90e3, 200, env, debug=False))
t("shoot_ray single, 90 km, S=200", lambda: pr.shoot_ray(1000.0, 0.0, 5.0, 90e3, 200, env, debug=False))
t("shoot_rays backwards 1000 angles", lambda: pr.shoot_rays(1000.0, 90e3, np.linspace(-15, 15, 1000), 0.0, 200, env, debug=False))
fan = pr.shoot_rays(1000.0, 0.0, np.linspace(-15, 15, 10000), 90e3, 200, env, debug=False)
import io, contextlib
def eig():
    with contextlib.redirect_stdout(io.StringIO()):
        return pr.find_eigenrays(fan, [1000.0, 2000.0], 1000.0, 0.0, 90e3, 200, env, debug=False)
er = t("find_eigenrays 2 receiver depths (from the 10 000-ray fan)", eig)
print("eigenrays found:", er.num_eigenrays_found)
