"""PGR_SAMPLE_BLOCKED against the plain sample-major layout: same bits, and the timing of both (configs[2], S = 1001)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import bench
from pygenray_amd import _lib
from pygenray_amd.device_fan import DeviceFan, fan_y0
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
if len(sys.argv) > 2:
    _lib.LIB_PATH = os.path.abspath(sys.argv[2])
modes = sys.argv[3].split(",") if len(sys.argv) > 3 else ["plain", "blocked", "plain2", "blocked2", "end_state"]
_, arrs = bench.munk_tables(1000e3, nr=101, sofar_slope=2e-4)
y0 = fan_y0(arrs, 1000.0, 0.0, -np.linspace(-20, 20, n))
env = _lib.EnvHandle(*arrs)
res = {}
for name, kw in (("plain", {}), ("blocked", dict(sample_blocked=True)), ("plain2", {}), ("blocked2", dict(sample_blocked=True)), ("end_state", dict(save=False))):
    if name not in modes:
        continue
    kw = dict(dict(save=True), **kw)
    fan = DeviceFan(env, y0, 0.0, 1000e3, 1001, sample_major=True, **kw)
    for _ in range(3): fan.run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(8):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fan.run(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    print(f"{os.path.basename(_lib.LIB_PATH)} {name:10s} {np.mean(ts):.3f} ms (min {np.min(ts):.3f})", flush=True)
    if kw["save"] and name in ("plain", "blocked"):
        res[name] = [fan.rows(t).cpu().numpy() for t in (fan.T, fan.Z, fan.P)] + [fan.end.cpu().numpy(), fan.status.cpu().numpy()]
    del fan
if "plain" not in res or "blocked" not in res:
    sys.exit(0)
a, b = res["plain"], res["blocked"]
print("blocked == plain:", all(np.array_equal(x, y, equal_nan=True) for x, y in zip(a, b)), " dropped", int((a[4] != 0).sum()),
      " NaN columns equal", np.array_equal(np.isnan(a[0]), np.isnan(b[0])))
