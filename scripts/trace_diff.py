"""Where does a ray that is not bit-identical to the oracle (MATH_CR) leave it?  For the rays of the
headline fan whose end state differs, replays every step attempt of the oracle's trace on the device
(pgr_debug_step) and reports the first attempt whose y_new / f_new / error_norm / power differs.
usage: trace_diff.py [stride] [config] [max_rays]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, oracle
from helpers import munk_arrays, y0_for
from pygenray_amd import _lib
stride = int(sys.argv[1]) if len(sys.argv) > 1 else 20
config = int(sys.argv[2]) if len(sys.argv) > 2 else 1
max_rays = int(sys.argv[3]) if len(sys.argv) > 3 else 12
arrs = munk_arrays(1000e3) if config == 1 else munk_arrays(1000e3, nr=101, sofar_slope=2e-4)
theta = np.linspace(-20, 20, 100_000)[::stride]
y0 = y0_for(oracle, arrs, 1000.0, 0.0, -theta)
o = oracle.shoot_fan(*arrs, y0, 0.0, 1000e3, 2, math=oracle.MATH_CR)
env = _lib.EnvHandle(*arrs)
g = env.shoot_fan(y0, 0.0, 1000e3, 2)
ok = (o["status"] == 0) & (g["status"] == 0)
diff = ok & np.any(g["end"] != np.stack([o["T"][:, -1], o["z"][:, -1], o["p"][:, -1]], 1), axis=1)
print(f"{int(diff.sum())} of {int(ok.sum())} rays differ")
names = ["T_new", "z_new", "p_new", "fT_new", "fz_new", "fp_new", "error_norm", "0.9*err**-0.2", "fT", "fz", "fp"]
for k in np.where(diff)[0][:max_rays]:
    tr = oracle.trace_ray(*arrs, y0[k], 0.0, 1000e3, math=oracle.MATH_CR)
    d = env.debug_step(tr[:, 0], tr[:, 2:5], tr[:, 1])
    # oracle's own values for the same attempt: f (given), error_norm (given); y_new / f_new = the next row's y / f after an accepted attempt
    acc = tr[:, 9] == 1
    nxt = np.r_[np.arange(1, len(tr)), len(tr) - 1]
    same_seg = tr[nxt, 11] == tr[:, 11]
    chk = acc & same_seg & (np.arange(len(tr)) < len(tr) - 1)
    ref = np.full((len(tr), 11), np.nan)
    ref[:, 8:11] = tr[:, 5:8]
    ref[:, 6] = tr[:, 8]
    ref[chk, 0:3] = tr[nxt[chk], 2:5]
    ref[chk, 3:6] = tr[nxt[chk], 5:8]
    ref[:, 7] = 0.9 * oracle.math_fn("pow_m02", tr[:, 8])
    with np.errstate(invalid="ignore"):
        # (after an event the next row's y is the event point, not y_new: only compare where the next attempt starts at t + h)
        cont = np.isclose(tr[nxt, 0], tr[:, 0] + tr[:, 1], rtol=0, atol=0)
        ref[~cont, 0:6] = np.nan
        bad = (d != ref) & ~np.isnan(ref)
    rows = np.where(bad.any(1))[0]
    print(f"ray {k} theta {theta[k]:.6f} bounces {int(o['n_bott'][k] + o['n_surf'][k])} attempts {len(tr)}: "
          f"{len(rows)} attempts differ; first {rows[:3]}")
    for r in rows[:2]:
        cols = np.where(bad[r])[0]
        print(f"   attempt {r}: t {tr[r, 0]:.6f} h {tr[r, 1]:.6f} err {tr[r, 8]:.17g} accepted {int(tr[r, 9])}")
        for c in cols:
            print(f"      {names[c]:14s} device {d[r, c]!r:>26} oracle {ref[r, c]!r:>26}  diff {d[r, c] - ref[r, c]:.3e} ({(d[r, c] - ref[r, c]) / np.spacing(abs(ref[r, c])):+.1f} ulp)")
