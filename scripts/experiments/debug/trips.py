import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from helpers import munk_arrays
from pygenray_amd import _lib
from pygenray_amd.device_fan import DeviceFan, fan_y0
arrs = munk_arrays(1000e3); env = _lib.EnvHandle(*arrs)
theta = np.linspace(-20, 20, 100000); y0 = fan_y0(arrs, 1000.0, 0.0, -theta)
for park in ((64, 64), (64, 16), (56, 64), (60, 64)):
    env.set_option("park", *park)
    fan = DeviceFan(env, y0, 0.0, 1000e3, 2, save=False); fan.run(); torch.cuda.synchronize()
    att = (fan.n_steps + fan.n_rej).cpu().numpy().astype(np.int64)
    fan.flags |= 16; fan.run(); torch.cuda.synchronize()
    dbg = fan.n_rej.cpu().numpy()
    nw = len(att) // 64
    a = att[:nw * 64].reshape(nw, 64); trips = dbg[:nw * 64].reshape(nw, 64)[:, 0]; serv = dbg[:nw * 64].reshape(nw, 64)[:, 1]; fb = dbg[:nw * 64].reshape(nw, 64)[:, 2:]
    mx = a.max(1); mean = a.mean(1)
    for lo, hi, name in ((0, 40, "steepest 40 waves"), (700, 800, "middle waves")):
        print(park, name, "trips", trips[lo:hi].mean().round(), "max-lane attempts", mx[lo:hi].mean().round(), "mean-lane attempts", mean[lo:hi].mean().round(), "services", serv[lo:hi].mean().round(), "exact-bisection fallbacks per wave (62 lanes)", fb[lo:hi].sum(1).mean().round(1), "overhead trips/max", (trips[lo:hi] / mx[lo:hi]).mean().round(3))
print("---- whole-fan maxima")
for park in ((64, 64), (64, 16), (64, 8), (64, 4)):
    env.set_option("park", *park)
    fan = DeviceFan(env, y0, 0.0, 1000e3, 2, save=False); fan.run(); torch.cuda.synchronize()
    att = (fan.n_steps + fan.n_rej).cpu().numpy().astype(np.int64)
    fan.flags |= 16; fan.run(); torch.cuda.synchronize()
    dbg = fan.n_rej.cpu().numpy()
    nw = len(att) // 64
    a = att[:nw * 64].reshape(nw, 64); trips = dbg[:nw * 64].reshape(nw, 64)[:, 0]; serv = dbg[:nw * 64].reshape(nw, 64)[:, 1]; fb = dbg[:nw * 64].reshape(nw, 64)[:, 2:]
    k = int(np.argmax(trips))
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    fan.flags &= ~16; e0.record(); fan.run(); e1.record(); torch.cuda.synchronize()
    print(park, "max trips", trips.max(), "at wave", k, "theta", theta[k * 64], "max-lane attempts there", a[k].max(), "services", serv[k], "| sum trips", trips.sum(), "| kernel ms", round(e0.elapsed_time(e1), 3))
