"""Per-wave trip and service counts of the 1e5-ray headline fan (PGR_DEBUG_TRIPS): how steeply a wave's run time falls
with its rank -- i.e. how many waves would have to be made faster to shorten the fan by x per cent."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from helpers import munk_arrays
from pygenray_amd import _lib
from pygenray_amd.device_fan import DeviceFan, fan_y0
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
arrs = munk_arrays(1000e3); env = _lib.EnvHandle(*arrs)
y0 = fan_y0(arrs, 1000.0, 0.0, -np.linspace(-20, 20, n))
fan = DeviceFan(env, y0, 0.0, 1000e3, 1001, save=True, sample_major=True); fan.flags |= 16; fan.run(); torch.cuda.synchronize()
d = fan.n_rej.cpu().numpy().astype(np.int64)
nw = (n + 63) // 64
d = np.pad(d, (0, nw * 64 - n)).reshape(nw, 64)
trips, services = d[:, 0], d[:, 1]
est = (trips * 4100.0 + services * 24000.0) / 2.4e6     # ms at 2.4 GHz, trajectory kernel
order = np.argsort(-est)
top = est[order[0]]
print(f"{nw} waves; most expensive: wave {order[0]} trips {trips[order[0]]} services {services[order[0]]} est {top:.2f} ms; sum of estimates {est.sum():.0f} SIMD-ms")
for frac in (0.99, 0.98, 0.97, 0.955, 0.94, 0.924, 0.9, 0.8, 0.7, 0.5):
    k = int((est >= frac * top).sum())
    print(f"  waves within {100 * (1 - frac):4.1f} % of the top: {k:5d}   (SIMD-ms they hold: {est[est >= frac * top].sum():7.0f})")
print("rank: est ms", [(int(r), round(float(est[order[r]]), 2)) for r in (0, 10, 20, 50, 100, 200, 300, 485, 600, 800, 1000, 1200, 1400, nw - 1)])
