"""PGR_SAMPLE_BLOCKED against the row layout over the random environments of tests/helpers.random_case that take the HBM-table
path (range-dependent sound speed): every ray, every sample, NaN columns, padding rows pre-filled.  usage: python scripts/fuzz_blocked.py"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, oracle
from helpers import random_case, y0_for
from pygenray_amd import _lib
from pygenray_amd.device_fan import DeviceFan
n_env = n_hbm = 0
for seed in range(20000, 20600):
    arrs, (src, x0, th), kw, desc = random_case(seed)
    env = _lib.EnvHandle(*arrs)
    n_env += 1
    if env.lds_path:
        env.close(); continue
    n_hbm += 1
    y0 = y0_for(oracle, arrs, src, x0, th)
    outs = []
    for blocked in (False, True):
        fan = DeviceFan(env, y0, kw["x0"], kw["x1"], kw["S"], rtol=kw["rtol"], terminate_backwards=kw["terminate_backwards"],
                        save=True, sample_major=True, sample_blocked=blocked)
        if blocked:
            for t in (fan.T, fan.Z, fan.P): t.fill_(123.0)
        fan.run(); torch.cuda.synchronize()
        outs.append([fan.rows(t).cpu().numpy() for t in (fan.T, fan.Z, fan.P)] + [fan.end.cpu().numpy(), fan.status.cpu().numpy()])
    ok = all(np.array_equal(a, b, equal_nan=True) for a, b in zip(*outs))
    if not ok:
        print("MISMATCH seed", seed, desc); sys.exit(1)
    env.close()
print(f"{n_env} random environments, {n_hbm} on the HBM-table path: blocked layout == row layout on every ray and sample")
