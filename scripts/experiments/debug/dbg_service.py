"""Where a bounce service spends its cycles (library built with -DPGR_DBG_REPLAY: scripts/build_variants.py dbgreplay):
s_memtime stamps accumulated over the services of one wave of steep rays."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from helpers import munk_arrays
from pygenray_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "scripts/ab/dbgreplay.so"))
from pygenray_amd.device_fan import DeviceFan, fan_y0
slope = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
arrs = munk_arrays(1000e3, nr=(101 if slope else 100), sofar_slope=slope); env = _lib.EnvHandle(*arrs)
for save in (False, True):
    theta = np.linspace(19.7, 19.8, 64); y0 = fan_y0(arrs, 1000.0, 0.0, -theta)
    fan = DeviceFan(env, y0, 0.0, 1000e3, 1001, save=save, sample_major=True); fan.flags |= 16; fan.run(); torch.cuda.synchronize()
    d = fan.n_rej.cpu().numpy().astype(np.int64)
    n = d[1]
    names = {2: "replay phase 1", 3: "replay phase 2 (incl. evaluations)", 4: "  of which true-event evaluations", 5: "phase-2 iterations (count)",
             6: "WHOLE SERVICE", 7: "stage replay + Q", 8: "Newton + band edges", 9: "(after replay, before samples)", 10: "samples of the truncated step + root + reflection (CR asin/sin)", 11: "restart (2 RHS, CR pow, events, nearest sample)",
             12: "  Newton (to the converged root)", 13: "  noise band E, nu, edges xa/xb", 14: "  true event at the two edges",
             15: "  samples + state at the root", 16: "  table look-up + CR arcsin", 17: "  reflection (beta, CR sin, divide)"}
    print(f"save={save}: trips {d[0]}, services {n}; cycles per service:")
    for k in range(2, 18):
        print(f"   {names[k]:70s} {d[k] / n:10.1f}")
