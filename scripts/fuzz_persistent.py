"""Persistent waves over the random environments of tests/helpers.random_case (every depth-search form, LDS- and HBM-table
kernels, rows and the sample-blocked layout, loose and tight tolerances, mirrored frames): a fan of 135 000 ... 300 000 rays
of each environment under PGR_OPT_PERSISTENT 0 / 1 / 2 / 3 -- every output array of every ray the same bits -- and every
400th ray of the default mode against the oracle (oracle.MATH_CR) bit for bit.
usage (GPU box): python scripts/fuzz_persistent.py [first:last] [flatearth]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, oracle
from helpers import random_case, y0_for
from pygenray_amd import _lib
from pygenray_amd.device_fan import DeviceFan
a, b = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "30000:30040").split(":"))
FLAT = len(sys.argv) > 2 and sys.argv[2] == "flatearth"
kinds = {}
n_rays = n_checked = 0
t0 = time.time()
for seed in range(a, b):
    arrs, (src, x0, th), kw, desc = random_case(seed)
    if FLAT:
        from pygenray_amd.environment import eflat
        cin, cpin, r, z, depths, br, ba = arrs
        zf, _ = eflat(z, 35.0, cin[0])
        cf = np.array([eflat(z, 35.0, row)[1] for row in cin])
        df, _ = eflat(depths, 35.0, np.zeros_like(depths) + 1500.0)
        arrs = [cf, np.gradient(cf, zf, axis=1, edge_order=1), r, zf, df, br, ba]
    rng = np.random.default_rng(seed)
    n = int(rng.integers(135_000, 300_000))
    S = int(rng.integers(2, 9))
    save = bool(rng.random() < 0.6)
    env = _lib.EnvHandle(*arrs)
    blocked = save and (not env.lds_path) and bool(rng.random() < 0.5)
    kind = ("LDS" if env.lds_path else "HBM") + (" cubic" if env.query(5) else " quad" if env.query(6) else " bins" if env.query(7) else
                                                  " uniform" if env.query(1) else " search") + (" blocked" if blocked else " rows" if save else " end")
    kinds[kind] = kinds.get(kind, 0) + 1
    y0 = y0_for(oracle, arrs, src, x0, np.linspace(th[0], th[-1], n))
    outs = {}
    for mode in (0, 1, 2, 3):
        env.set_option("persistent", mode)
        fan = DeviceFan(env, y0, kw["x0"], kw["x1"], S, rtol=kw["rtol"], terminate_backwards=kw["terminate_backwards"], save=save,
                        sample_major=True, sample_blocked=blocked)
        fan.run(); torch.cuda.synchronize()
        o = {k: getattr(fan, k).cpu().numpy() for k in ("end", "n_bott", "n_surf", "status", "n_steps", "n_rej")}
        if save:
            o.update({k: fan.rows(getattr(fan, k)).cpu().numpy() for k in ("T", "Z", "P")})
        outs[mode] = o
        del fan
    for mode in (1, 2, 3):
        for k in outs[0]:
            if not np.array_equal(outs[0][k], outs[mode][k], equal_nan=True):
                print("MISMATCH seed", seed, desc, "mode", mode, k); sys.exit(1)
    sub = np.arange(0, n, 400)
    ref = oracle.shoot_fan(*arrs, y0[sub], kw["x0"], kw["x1"], S, rtol=kw["rtol"], terminate_backwards=kw["terminate_backwards"], math=oracle.MATH_CR)
    g = outs[1]
    ok = ref["status"] == 0
    end_o = np.stack([ref["T"][:, -1], ref["z"][:, -1], ref["p"][:, -1]], 1)
    if not (np.array_equal(ref["status"], g["status"][sub]) and np.array_equal(end_o[ok], g["end"][sub][ok]) and
            np.array_equal(ref["n_steps"][ok], g["n_steps"][sub][ok]) and np.array_equal(ref["n_rej"][ok], g["n_rej"][sub][ok])):
        print("ORACLE MISMATCH seed", seed, desc); sys.exit(1)
    n_rays += n; n_checked += len(sub)
    env.close()
print(f"{b - a} random environments{' (flat-earth mapped)' if FLAT else ''}, {n_rays} rays in fans of 135 000 ... 300 000: persistent modes 1 / 2 / 3 == static deal on every ray and sample; "
      f"{n_checked} rays (every 400th) bit-identical to the oracle; kernel shapes: {kinds}; {time.time() - t0:.0f} s")
