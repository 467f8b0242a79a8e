"""Randomised BIT-PARITY sweep, GPU vs the oracle in its correctly-rounded-libm mode, over the random
environments and shots of tests/helpers.random_case.  Every ray must agree bit for bit -- status,
bounce counts, accepted and rejected steps, end state, every sample (SciPy order).
usage: fuzz_bitparity.py [n_envs | seed,seed,... | first:last] [lib.so | -] [flatearth]
`flatearth`: every environment's depth grid, sound speeds and sea floor go through the reference's flat-earth map first
(REF/environment.py:121-154, 371-401): smoothly non-uniform zin -> the cubic-index depth look-up (kernel ZM = 5) where the
grid qualifies, the three-node / bin-table forms where it does not."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle
from helpers import y0_for, random_case
from pygenray_amd import _lib

arg = sys.argv[1] if len(sys.argv) > 1 else "60"
seeds = ([int(v) for v in arg.split(",") if v] if "," in arg else
         list(range(*[int(v) for v in arg.split(":")])) if ":" in arg else list(range(int(arg))))
n_seeds = len(seeds)
if len(sys.argv) > 2 and sys.argv[2] != "-":
    _lib.LIB_PATH = os.path.abspath(sys.argv[2])
FLAT = len(sys.argv) > 3 and sys.argv[3] == "flatearth"
n_cubic = 0
tot = odd = 0
worst = []
t_start = time.time()
for seed in seeds:
    arrs, (src, x0, th), kw, desc = random_case(seed)
    if FLAT:
        from pygenray_amd.environment import eflat
        cin, cpin, rin, zin, depths, dr, ba = arrs
        zf = eflat(zin, 35.0)[0]
        cf = np.array([eflat(zin, 35.0, row)[1] for row in cin])
        arrs = [cf, np.gradient(cf, zf, axis=1, edge_order=1), rin, zf, eflat(depths, 35.0)[0], dr, ba]
    y0 = y0_for(oracle, arrs, src, x0, th)
    env = _lib.EnvHandle(*arrs)
    n_cubic += env.query(5)
    g = env.shoot_fan(y0, kw["x0"], kw["x1"], kw["S"], rtol=kw["rtol"], exact_samples=True, terminate_backwards=kw["terminate_backwards"])
    env.close()
    o = oracle.shoot_fan(*arrs, y0, kw["x0"], kw["x1"], kw["S"], rtol=kw["rtol"], math=oracle.MATH_CR,
                         terminate_backwards=kw["terminate_backwards"])
    ok = o["status"] == 0
    same = (g["status"] == o["status"])
    same &= ~ok | ((g["n_bott"] == o["n_bott"]) & (g["n_surf"] == o["n_surf"]) & (g["n_steps"] == o["n_steps"]) & (g["n_rej"] == o["n_rej"]))
    for nm in "Tzp":
        same &= np.all((g[nm] == o[nm]) | (np.isnan(g[nm]) & np.isnan(o[nm])), axis=1)
    end_ref = np.stack([o["T"][:, -1], o["z"][:, -1], o["p"][:, -1]], 1)
    same &= ~ok | np.all(g["end"] == end_ref, axis=1)
    n_odd = int((~same).sum())
    tot += len(th); odd += n_odd
    bounces = int((o["n_bott"] + o["n_surf"])[ok].sum())
    print(f"seed {seed:3d}: {desc} dropped {int((~ok).sum()):3d} bounces {bounces:5d}  not bit-identical: {n_odd}"
          + (f"  <-- rays {np.where(~same)[0][:6]} status g/o {g['status'][~same][:4]}/{o['status'][~same][:4]}" if n_odd else ""), flush=True)
    if n_odd:
        worst.append((seed, n_odd))
print(f"{n_cubic} of the environments took the cubic-index depth look-up (ZM = 5)")
print(f"{n_seeds} environments, {tot} rays, {odd} not bit-identical ({odd / tot:.2e}); {time.time() - t_start:.0f} s; seeds with odd rays: {worst}")
