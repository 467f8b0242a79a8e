"""Where one step attempt of a LONE wave spends its time (experiments only).

Needs a library built with -DPGR_TIMING (s_memtime stamps along the attempt, see PGR_STAMP in
pgr_hip.hip):  hipcc <flags of _lib.HIPCC_FLAGS> -DPGR_TIMING -o pygenray_amd/csrc/timing.so ...
python scripts/phase_times.py pygenray_amd/csrc/timing.so
Prints s_memtime ticks per trip between consecutive stamps (the stamps themselves cost a few
ticks each and fence the scheduler, so the sum is larger than an uninstrumented trip)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from helpers import munk_arrays
from pygenray_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
from pygenray_amd.device_fan import DeviceFan, fan_y0

NAMES = ["0 gate/loop top", "1 min_step, t_new", "2 step_weights", "3 sum -> z2", "4 look-up+rhs 2", "5 sum -> z3",
         "6 look-up+rhs 3", "7 sum -> z4", "8 look-up+rhs 4", "9 sum -> z5", "10 look-up+rhs 5", "11 sum -> z6",
         "12 look-up+rhs 6", "13 sum -> y_new", "14 look-up+rhs 7", "15 error norm", "16 controller",
         "17 events", "18 commit/samples", "19 (rejected: skip)"]
arrs = munk_arrays(1000e3); env = _lib.EnvHandle(*arrs)
theta = np.linspace(-20, -19.9748, 64); y0 = fan_y0(arrs, 1000.0, 0.0, -theta)
for save in (False, True):
    fan = DeviceFan(env, y0, 0.0, 1000e3, 1001 if save else 2, save=save); fan.run(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); fan.run(); e1.record(); torch.cuda.synchronize()
    acc = fan.n_rej.cpu().numpy()[:24].astype(np.int64) & 0xffffffff
    fan.flags |= 16; fan.run(); torch.cuda.synchronize()
    trips = int(fan.n_rej.cpu().numpy()[0])
    print(f"save={save}: kernel {e0.elapsed_time(e1):.3f} ms, {trips} trips, {acc.sum() / trips:.0f} ticks per trip")
    for k, nm in enumerate(NAMES):
        print(f"   {nm:24s} {acc[k] / trips:8.1f}")
