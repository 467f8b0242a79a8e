"""In how many trips of the fan's steepest wave does ANY lane evaluate a sample, and how many lanes do (library built
with -DPGR_DBG_SAMPLE_TRIPS: scripts/build_variants.py smptrips)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from helpers import munk_arrays
from pygenray_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "scripts/ab/smptrips.so")
from pygenray_amd.device_fan import DeviceFan, fan_y0
arrs = munk_arrays(1000e3); env = _lib.EnvHandle(*arrs)
for a0, a1 in ((-20, -19.9748), (-15, -14.9748), (-8, -7.9748)):
    y0 = fan_y0(arrs, 1000.0, 0.0, -np.linspace(a0, a1, 64))
    fan = DeviceFan(env, y0, 0.0, 1000e3, 1001, save=True, sample_major=True); fan.flags |= 16; fan.run(); torch.cuda.synchronize()
    d = fan.n_rej.cpu().numpy().astype(np.int64)
    print(f"angles {a0}..{a1}: trips {d[0]}, services {d[1]}; trips with a sample in some lane {d[2]} ({d[2] / d[0]:.2f}); lanes evaluating per such trip "
          f"{d[3] / max(d[2], 1):.1f}; stepping lanes per trip {d[4] / d[0]:.1f}")
