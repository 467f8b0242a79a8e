"""Distribution of HIP-vs-oracle end-state deviations on the headline fan (every k-th ray),
next to the oracle's own spread under 1-ulp perturbations.
usage: parity_stats.py [lib.so|-] [stride] [exact]   (exact: PGR_EXACT_BISECTION, SciPy's own event bisection)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, oracle
from helpers import munk_arrays, y0_for, oracle_selfnoise
from pygenray_amd import _lib
if len(sys.argv) > 1 and sys.argv[1] != "-":
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
stride = int(sys.argv[2]) if len(sys.argv) > 2 else 100
arrs = munk_arrays(1000e3)
theta = np.linspace(-20, 20, 100_000)[::stride]
y0 = y0_for(oracle, arrs, 1000.0, 0.0, -theta)
o = oracle.shoot_fan(*arrs, y0, 0.0, 1000e3, 2)
noise = oracle_selfnoise(oracle, arrs, y0, 0.0, 1000e3, 2)
exact = len(sys.argv) > 3 and sys.argv[3] == "exact"
g = _lib.EnvHandle(*arrs).shoot_fan(y0, 0.0, 1000e3, 2, exact_bisection=exact)
ok = (o["status"] == 0) & (g["status"] == 0)
quiet = ((o["n_bott"] + o["n_surf"]) == 0) & ok
bnc = ok & ~quiet
print("lib:", _lib.LIB_PATH, "exact bisection:", exact, " rays:", len(theta), "status equal:", np.array_equal(o["status"], g["status"]),
      "bounce counts equal:", np.array_equal(o["n_bott"][ok], g["n_bott"][ok]) and np.array_equal(o["n_surf"][ok], g["n_surf"][ok]),
      "n_steps equal frac:", np.mean(o["n_steps"][ok] == g["n_steps"][ok]))
def q(x): return " ".join(f"{v:.1e}" for v in np.quantile(x, [0.5, 0.9, 0.99, 1.0]))
for name, m in (("non-bouncing", quiet), ("bouncing", bnc)):
    dz = np.abs(g["end"][m, 1] - o["z"][m, -1]) / 5000.0
    dt = np.abs(g["end"][m, 0] - o["T"][m, -1]) / o["T"][m, -1]
    nz = np.max([np.abs(n["z"][m, -1] - o["z"][m, -1]) for n in noise], axis=0) / 5000.0
    nt = np.max([np.abs(n["T"][m, -1] - o["T"][m, -1]) for n in noise], axis=0) / o["T"][m, -1]
    print(f"{name:13s} n={m.sum():5d}  rel dz (q50 q90 q99 max): HIP-vs-oracle {q(dz)} | oracle 1-ulp spread {q(nz)} | frac(HIP<=1e-8) {np.mean(dz<=1e-8):.3f} frac(oracle<=1e-8) {np.mean(nz<=1e-8):.3f}")
    print(f"{'':13s}          rel dT (q50 q90 q99 max): HIP-vs-oracle {q(dt)} | oracle 1-ulp spread {q(nt)}")
