"""Persistent waves (PGR_OPT_PERSISTENT 1, the default for fans of several rounds; 2 / 3: every packet from the list's head /
the SIMD partners' first packets from its cheap end, which 1 chooses between by the number of rounds) against the static
deal of whole cost-sorted workgroups (0): every output array of every ray must hold the same bits -- which wave integrates a
packet, and when, never changes what it computes -- and the kernel times of both.

usage (GPU box): python scripts/persist_check.py [--lib x.so] [--quick]
"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from helpers import munk_arrays
from pygenray_amd import _lib
from pygenray_amd.device_fan import DeviceFan, fan_y0

ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=None)
ap.add_argument("--quick", action="store_true")
ap.add_argument("--out", default=None)
a = ap.parse_args()
if a.lib:
    _lib.LIB_PATH = os.path.abspath(a.lib)


def flat_earth(arrs):
    import pygenray_amd as pr
    from pygenray_amd.environment import _unpack_envi
    cin, _, r, z, depths, br, _ = arrs
    env = pr.OceanEnvironment2D(pr.DataArray(cin, dims=["range", "depth"], coords={"range": r, "depth": z}),
                                pr.DataArray(depths, dims=["range"], coords={"range": br}), flat_earth_transform=True)
    return _unpack_envi(env, flatearth=True)


cases = [("configs[1] tables, 1e6 rays, end state only", munk_arrays(1000e3), 1_000_000, dict(save=False)),
         ("configs[1] tables, 140k rays, end state only", munk_arrays(1000e3), 140_000, dict(save=False)),
         ("configs[1] tables, 200k rays, S = 101 rows", munk_arrays(1000e3), 200_000, dict(save=True, S=101)),
         ("configs[1] tables, 300k rays, S = 101 rows", munk_arrays(1000e3), 300_000, dict(save=True, S=101)),
         ("configs[2] tables, 300k rays, end state only", munk_arrays(1000e3, nr=101, sofar_slope=2e-4), 300_000, dict(save=False)),
         ("configs[2] tables, 300k rays, S = 103 sample-blocked", munk_arrays(1000e3, nr=101, sofar_slope=2e-4), 300_000, dict(save=True, S=103, blocked=True)),
         ("configs[2] tables, 300k rays, S = 101 rows", munk_arrays(1000e3, nr=101, sofar_slope=2e-4), 300_000, dict(save=True, S=101)),
         ("flat-earth tables (cubic index), 300k rays, S = 51 rows", flat_earth(munk_arrays(1000e3)), 300_000, dict(save=True, S=51))]
if a.quick:
    cases = cases[:1]
res = []
bad = 0
for name, arrs, n, kw in cases:
    env = _lib.EnvHandle(*arrs)
    y0 = fan_y0(arrs, 1000.0, 0.0, -np.linspace(-20, 20, n))
    outs, times = {}, {}
    for mode in (0, 1, 2, 3):
        env.set_option("persistent", mode)
        fan = DeviceFan(env, y0, 0.0, 1000e3, kw.get("S", 1), save=kw["save"], sample_major=True, sample_blocked=kw.get("blocked", False))
        fan.run(); torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); fan.run(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        times[mode] = min(ts)
        o = {k: getattr(fan, k).cpu().numpy() for k in ("end", "n_bott", "n_surf", "status", "n_steps", "n_rej")}
        if kw["save"]:
            for k in ("T", "Z", "P"):
                o[k] = fan.rows(getattr(fan, k)).cpu().numpy()
        outs[mode] = o
        steps = fan.ray_steps()
        del fan
    diff = [(m, k) for m in (1, 2, 3) for k in outs[0] if not np.array_equal(outs[0][k], outs[m][k], equal_nan=True)]
    bad += len(diff)
    r = {"case": name, "rays": n, "static_ms": times[0], "persistent_ms": times[1], "persistent_all_from_the_head_ms": times[2], "persistent_partners_start_at_the_cheap_end_ms": times[3],
         "ray_steps": steps, "arrays_that_differ": diff}
    res.append(r)
    print(json.dumps(r), flush=True)
    env.close()
if a.out:
    json.dump(res, open(a.out, "w"), indent=1)
sys.exit(1 if bad else 0)
