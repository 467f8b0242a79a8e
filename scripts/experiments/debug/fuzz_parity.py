"""Randomised GPU-vs-oracle sweep: smooth random environments (uniform / stretched depth grids,
uniform / random range grids, flat / sloping sea floors, range (in)dependent sound speed), random
sources and tolerances.  Flags rays whose status or bounce counts differ from the C oracle or whose
end state differs by more than the oracle's own 1-ulp self-noise allows."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle
from helpers import munk, y0_for, oracle_selfnoise
from pygenray_amd import _lib

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
only = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else None   # seeds to run
if len(sys.argv) > 3:
    _lib.LIB_PATH = os.path.abspath(sys.argv[3])                                  # e.g. the -DPGR_STRICT build
bad_total = 0
t_start = time.time()
for seed in (only if only else range(n_seeds)):
    rng = np.random.default_rng(1000 + seed)
    zmax = rng.uniform(1500, 6000)
    nz = int(rng.integers(150, 2500))
    kind_z = rng.integers(0, 3)
    if kind_z == 0:
        z = np.linspace(0, zmax, nz)                       # uniform (not a power of two in general)
    elif kind_z == 1:
        z = np.arange(0, zmax, 2.0 ** rng.integers(-1, 3))  # j * dz, dz a power of two
    else:
        z = zmax * np.linspace(0, 1, nz) ** rng.uniform(1.0, 1.6)  # stretched
    rmax = rng.uniform(30e3, 300e3)
    nr = int(rng.integers(3, 120))
    r = np.linspace(0, rmax, nr) if rng.random() < 0.6 else np.sort(np.concatenate([[0, rmax], rng.uniform(0, rmax, nr - 2)]))
    slope = 0.0 if rng.random() < 0.4 else rng.uniform(-2e-3, 2e-3)
    axis = rng.uniform(0.15, 0.5) * zmax
    cin = np.array([munk(z, axis + slope * ri) for ri in r]) if slope else np.tile(munk(z, axis), (len(r), 1))
    cpin = np.gradient(cin, z, axis=1, edge_order=1)
    nb = int(rng.integers(4, 60))
    br = np.linspace(0, rmax, nb) if rng.random() < 0.5 else np.sort(np.concatenate([[0, rmax], rng.uniform(0, rmax, nb - 2)]))
    floor = rng.uniform(0.7, 0.95) * zmax
    depths = np.full(nb, floor) if rng.random() < 0.4 else floor + rng.uniform(0, 0.04) * zmax * np.sin(br / rng.uniform(20e3, 90e3))
    ba = np.degrees(np.arctan(np.gradient(depths, br)))
    arrs = [cin, cpin, r, z, depths, br, ba]
    src = rng.uniform(0.05, 0.6) * zmax
    rtol = [1e-9, 1e-7, 1e-5][int(rng.integers(0, 3))]
    x1 = rng.uniform(0.3, 1.0) * rmax
    th = np.linspace(-rng.uniform(5, 25), rng.uniform(5, 25), 96)
    y0 = y0_for(oracle, arrs, src, 0.0, th)
    S = int(rng.integers(2, 60))
    env = _lib.EnvHandle(*arrs)
    g = env.shoot_fan(y0, 0.0, x1, S, rtol=rtol)
    env.close()
    o = oracle.shoot_fan(*arrs, y0, 0.0, x1, S, rtol=rtol)
    noise = oracle_selfnoise(oracle, arrs, y0, 0.0, x1, S, rtol=rtol)
    st_bad = (g["status"] != 0) != (o["status"] != 0)
    ok = (g["status"] == 0) & (o["status"] == 0)
    cnt_bad = ok & ((g["n_bott"] != o["n_bott"]) | (g["n_surf"] != o["n_surf"]))
    dz = np.abs(g["end"][:, 1] - o["z"][:, -1]) / zmax
    spread = np.zeros(len(th))
    for n_ in noise:
        s = np.abs(n_["z"][:, -1] - o["z"][:, -1]) / zmax
        s[n_["status"] != 0] = np.inf
        spread = np.maximum(spread, np.nan_to_num(s, nan=np.inf))
    cls_spread = np.where(np.isfinite(spread), spread, 0).max()
    tol = np.maximum(np.maximum(1e-8, 1e-3 * rtol), np.maximum(20 * np.where(np.isfinite(spread), spread, 0), 3 * cls_spread))
    val_bad = ok & ~cnt_bad & (dz > tol)
    # a different status / bounce count only counts when the oracle's own 1-ulp neighbours all agree with it
    stable = np.ones(len(th), bool)
    for n_ in noise:
        stable &= (n_["status"] == o["status"]) & (n_["n_bott"] == o["n_bott"]) & (n_["n_surf"] == o["n_surf"])
    nbad = int((st_bad & stable).sum() + (cnt_bad & stable).sum() + val_bad.sum())
    bad_total += nbad
    print(f"seed {seed:3d}: nz {len(z):5d} ({['uniform','pow2','stretched'][kind_z]}) nr {len(r):3d} nb {nb:2d} rtol {rtol:g} S {S:2d} "
          f"range-dep {bool(slope)!s:5s} dropped {int((o['status'] != 0).sum()):2d} "
          f"max rel dz {np.nanmax(np.where(ok, dz, 0)):.1e}  flagged {nbad}"
          + (f"  <-- status {np.where(st_bad & stable)[0][:4]} counts {np.where(cnt_bad & stable)[0][:4]} values {np.where(val_bad)[0][:4]}" if nbad else ""), flush=True)
print(f"{n_seeds} environments, {bad_total} flagged rays, {time.time() - t_start:.0f} s")
sys.exit(1 if bad_total else 0)
