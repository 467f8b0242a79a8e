"""PGR_SAMPLE_BLOCKED against the row layout over the random environments of tests/helpers.random_case that take the HBM-table
path (range-dependent sound speed): every ray, every sample, NaN columns, padding rows pre-filled.  usage: python scripts/fuzz_blocked.py [first:last]   (default 20000:20600)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, oracle
from helpers import random_case, y0_for
from pygenray_amd import _lib
from pygenray_amd.device_fan import DeviceFan
n_env = n_hbm = 0
_A, _B = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "20000:20600").split(":"))
for seed in range(_A, _B):
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
    # the host-pointer entry (what API callers get): sample-blocked kernel + un-blocking pass against plain rows, with and
    # without the compaction of dropped rays, and a fan handle's fetches
    for compact in (False, True):
        res = []
        for api_blocked in (0, 1):
            env.set_option("api_blocked", api_blocked)
            res.append(env.shoot_fan(y0, kw["x0"], kw["x1"], kw["S"], rtol=kw["rtol"], terminate_backwards=kw["terminate_backwards"],
                                     sample_major=True, stored_sign=True, compact=compact))
        for k in ("T", "z", "p", "end", "status", "n_bott", "n_surf", "n_steps", "n_rej"):
            if not (res[0][k].shape == res[1][k].shape and np.array_equal(res[0][k], res[1][k], equal_nan=True)):
                print("API MISMATCH seed", seed, desc, "compact", compact, k); sys.exit(1)
        if not compact and not all(np.array_equal(res[1][k], -o if k in "zp" else o, equal_nan=True) for k, o in zip("Tzp", outs[0][:3])):
            print("API vs device MISMATCH seed", seed, desc); sys.exit(1)
    h = _lib.FanHandle(env, kw["x0"], kw["x1"], kw["S"], y0=y0, rtol=kw["rtol"], terminate_backwards=kw["terminate_backwards"], stored_sign=True)
    full = h.fetch_samples(compact=False)
    ref_full = env.shoot_fan(y0, kw["x0"], kw["x1"], kw["S"], rtol=kw["rtol"], terminate_backwards=kw["terminate_backwards"],
                             sample_major=True, stored_sign=True)
    if not all(np.array_equal(full[k], ref_full[k], equal_nan=True) for k in "Tzp"):
        print("FAN HANDLE MISMATCH seed", seed, desc); sys.exit(1)
    h.close()
    env.close()
print(f"{n_env} random environments, {n_hbm} on the HBM-table path: blocked layout == row layout on every ray and sample "
      "(device entry, host entry with and without compaction, fan handle)")
