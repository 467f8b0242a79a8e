"""Kernel timing harness: python scripts/kbench.py [--rays N] [--range KM] [--wpb W ...] [--amin A --amax B]"""
import sys, os, argparse, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pygenray_amd import _lib
from pygenray_amd.device_fan import DeviceFan, fan_y0
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from helpers import munk_arrays

ap = argparse.ArgumentParser()
ap.add_argument("--rays", type=int, default=100000)
ap.add_argument("--km", type=float, default=1000)
ap.add_argument("--wpb", type=int, nargs="*", default=[0])
ap.add_argument("--amin", type=float, default=-20)
ap.add_argument("--amax", type=float, default=20)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--modes", nargs="*", default=["nosave", "ray", "sample"])
ap.add_argument("--slope", type=float, default=0.0, help="sofar slope -> range dependent")
ap.add_argument("--S", type=int, default=1001)
ap.add_argument("--lib", default=None)
ap.add_argument("--place", type=int, default=-1)
ap.add_argument("--park", type=int, nargs="*", default=[64, 10], help="pairs: lanes trips lanes trips ...")
ap.add_argument("--persistent", type=int, default=-1)
ap.add_argument("--exact", action="store_true")
ap.add_argument("--exact-samples", action="store_true")
a = ap.parse_args()
if a.lib:
    _lib.LIB_PATH = os.path.abspath(a.lib)
arrs = munk_arrays(a.km * 1e3, nr=(101 if a.slope else 100), sofar_slope=a.slope)
env = _lib.EnvHandle(*arrs)
if a.place >= 0:
    env.set_option("placement", a.place)
if a.persistent >= 0:
    env.set_option("persistent", a.persistent)
theta = np.linspace(a.amin, a.amax, a.rays)
y0 = fan_y0(arrs, 1000.0, 0.0, -theta)
for mode in a.modes:
    fan = DeviceFan(env, y0, 0.0, a.km * 1e3, a.S, save=(mode != "nosave"), sample_major=(mode == "sample"), exact_bisection=a.exact, exact_samples=a.exact_samples)
    for w, (pl, pt) in [(w, pp) for w in a.wpb for pp in zip(a.park[0::2], a.park[1::2])]:
        env.set_option("waves_per_block", w)
        env.set_option("park", pl, pt)
        fan.run(); torch.cuda.synchronize()
        ts = []
        for _ in range(a.reps):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); fan.run(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        steps = fan.ray_steps(); rej = int(fan.n_rej.sum().item())
        ms = min(ts)
        print(f"mode={mode:7s} wpb={w} park=({pl},{pt}) rays={a.rays} km={a.km:.0f} angles=[{a.amin},{a.amax}] kernel {ms:8.3f} ms  "
              f"steps {steps:.4e} rej {rej:.3e}  {steps/ms/1e6:8.2f} G ray-steps/s  dropped {(fan.status!=0).sum().item()}", flush=True)
    del fan
