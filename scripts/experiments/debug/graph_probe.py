"""Is a fan launch capturable into a HIP graph (torch.cuda.CUDAGraph) and replayable?"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from helpers import munk_arrays
from pygenray_amd import _lib
from pygenray_amd.device_fan import DeviceFan, fan_y0
arrs = munk_arrays(1000e3); env = _lib.EnvHandle(*arrs)
y0 = fan_y0(arrs, 1000.0, 0.0, -np.linspace(-20, 20, 100000))
fan = DeviceFan(env, y0, 0.0, 1000e3, 101, save=True, sample_major=True)
fan.run(); torch.cuda.synchronize()
ref = fan.end.clone(); refT = fan.T.clone()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    fan.run(); torch.cuda.synchronize()   # warm up on the side stream (allocations done)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        fan.run()
fan.end.zero_(); fan.T.zero_()
g.replay(); torch.cuda.synchronize()
print("graph replay equals eager:", torch.equal(fan.end.nan_to_num(), ref.nan_to_num()), torch.equal(fan.T.nan_to_num(), refT.nan_to_num()))
t0 = time.perf_counter()
for _ in range(20): g.replay()
torch.cuda.synchronize(); print("graph replay ms/step", (time.perf_counter() - t0) / 20 * 1e3)
t0 = time.perf_counter()
for _ in range(20): fan.run()
torch.cuda.synchronize(); print("eager ms/step", (time.perf_counter() - t0) / 20 * 1e3)
