import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from pygenray_amd import _lib
from pygenray_amd.device_fan import DeviceFan, fan_y0
def run(z, c, label, ang=(5.0, 5.1), n=64):
    nr = 100; r = np.linspace(0, 1000e3, nr)
    cin = np.tile(c, (nr, 1)); cpin = np.gradient(cin, z, axis=1, edge_order=1)
    arrs = [cin, cpin, r, z, np.full(nr, 1e9), r.copy(), np.zeros(nr)]
    env = _lib.EnvHandle(*arrs)
    y0 = fan_y0(arrs, 1000.0, 0.0, np.linspace(*ang, n))
    fan = DeviceFan(env, y0, 0.0, 1000e3, 2, save=False)
    fan.run(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); fan.run(); e1.record(); torch.cuda.synchronize()
    att = (fan.n_steps + fan.n_rej).cpu().numpy()
    ms = e0.elapsed_time(e1)
    print(f"{label:40s} ZS={env.query(1)} lds={env.lds_path} trips(max lane)={att.max()} ms={ms:.3f} us/trip={1e3*ms/att.max():.3f} status0={int((fan.status==0).sum())}")
# linear profile, 2 nodes: every lane reads the same LDS entry (broadcast)
run(np.array([0.0, 8192.0]), np.array([1500.0, 1500.0 + 0.016 * 8192]), "linear c(z), nz=2 (broadcast reads)")
# same linear profile on a 1 m grid: distinct addresses per lane
z = np.arange(0, 8192, 1.0)
run(z, 1500.0 + 0.016 * z, "linear c(z), nz=8192, dz=1 (spread reads)")
