"""Bit-level agreement of the HIP path with the CPU oracle in its correctly-rounded-libm mode
(oracle.MATH_CR) on every k-th ray of the headline fan (configs[1]) or of the range-dependent
fan (configs[2]): end states, accepted / rejected step counts and bounce counts.  Configs 3-5 are the reference's
DEFAULT environment handling (flat-earth transformed tables, non-uniform depth grid -> the cubic-index look-up):
3 = configs[1]'s tables transformed, 4 = configs[2]'s tables transformed (both 1000 km), 5 = OceanEnvironment2D()
itself (4500 -> 4900 m slope, 100 km).  11 / 12 / 13 = the tables bench.py itself builds for its headline line, its
range_dependent leg and its flatearth_default leg.
usage: bitparity.py [lib.so|-] [stride] [config 1..5] [exact|-] [angles in the fan, default 100000] [S, default 101] [default-form|-]
(default-form: also the DEFAULT sample form -- stage-major FMAs inside a step, the SAVE = 1 kernel bench.py times -- against the oracle:
bit-equal outside a step and in the last column, deviation inside a step reported)"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, oracle
from helpers import munk_arrays, y0_for
from pygenray_amd import _lib
if len(sys.argv) > 1 and sys.argv[1] != "-":
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
stride = int(sys.argv[2]) if len(sys.argv) > 2 else 100
config = int(sys.argv[3]) if len(sys.argv) > 3 else 1
exact = len(sys.argv) > 4 and sys.argv[4] == "exact"
rmax = 1000e3
if config in (1, 2):
    arrs = munk_arrays(1000e3) if config == 1 else munk_arrays(1000e3, nr=101, sofar_slope=2e-4)
elif config in (11, 12, 13):
    # bench.py's OWN tables (bench.munk_tables: the drop-in environment, whose bottom angle is arctan(np.gradient(5000 m)) =
    # 1e-15 degrees, not 0): 11 = the headline line, 12 = its range_dependent leg, 13 = its flatearth_default leg
    import bench
    from pygenray_amd.environment import _unpack_envi
    env_obj, arrs = bench.munk_tables(1000e3, nr=101, sofar_slope=2e-4) if config == 12 else bench.munk_tables(1000e3)
    if config == 13:
        env_obj.flat_earth_transform(lat=35)
        arrs = _unpack_envi(env_obj, flatearth=True)
else:
    import pygenray_amd as pr
    from pygenray_amd.environment import _unpack_envi
    if config == 5:
        env_obj, rmax = pr.OceanEnvironment2D(), 100e3
    else:
        z = np.arange(0, 6000, 1.0); r = np.linspace(0, rmax, 100 if config == 3 else 101)
        ssp = pr.DataArray(np.array([pr.munk_ssp(z, 1300 + (2e-4 if config == 4 else 0.0) * ri) for ri in r]), dims=["range", "depth"],
                           coords={"range": r, "depth": z})
        env_obj = pr.OceanEnvironment2D(ssp, pr.DataArray(np.full(len(r), 5000.0), dims=["range"], coords={"range": r}), flat_earth_transform=True)
    arrs = _unpack_envi(env_obj, flatearth=True)
n_fan = int(sys.argv[5]) if len(sys.argv) > 5 else 100_000
theta = np.linspace(-20, 20, n_fan)[::stride]
y0 = y0_for(oracle, arrs, 1000.0, 0.0, -theta)
S = int(sys.argv[6]) if len(sys.argv) > 6 else 101
t0 = time.time()
o = oracle.shoot_fan(*arrs, y0, 0.0, rmax, S, math=oracle.MATH_CR)
t1 = time.time()
env = _lib.EnvHandle(*arrs)
g = env.shoot_fan(y0, 0.0, rmax, S, exact_bisection=exact, exact_samples=True)
ok = (o["status"] == 0) & (g["status"] == 0)
quiet = ((o["n_bott"] + o["n_surf"]) == 0) & ok
print(f"lib {_lib.LIB_PATH} config {config} rays {len(theta)} (oracle CR {t1 - t0:.1f} s) exact_bisection={exact} cubic-index look-up={env.query(5)} table in LDS={env.lds_path}")
print("status equal:", np.array_equal(o["status"], g["status"]), " dropped:", int((o["status"] != 0).sum()))
for name, m in (("non-bouncing", quiet), ("bouncing", ok & ~quiet), ("all ok", ok)):
    if not m.any():
        continue
    end_eq = np.all(g["end"][m] == np.stack([o["T"][m, -1], o["z"][m, -1], o["p"][m, -1]], 1), axis=1)
    st_eq = o["n_steps"][m] == g["n_steps"][m]
    rj_eq = o["n_rej"][m] == g["n_rej"][m]
    bc_eq = (o["n_bott"][m] == g["n_bott"][m]) & (o["n_surf"][m] == g["n_surf"][m])
    smp_eq = np.all((g["T"][m] == o["T"][m]) & (g["z"][m] == o["z"][m]) & (g["p"][m] == o["p"][m]), axis=1)
    dz = np.abs(g["end"][m, 1] - o["z"][m, -1]) / 5000.0
    print(f"{name:13s} n={int(m.sum()):6d}  bit-equal: end state {end_eq.mean():.5f}  n_steps {st_eq.mean():.5f}  n_rej {rj_eq.mean():.5f}  "
          f"bounces {bc_eq.mean():.5f}  all {S} samples (SciPy order) {smp_eq.mean():.5f} | rel dz q50 {np.median(dz):.1e} q99 {np.quantile(dz, 0.99):.1e} max {dz.max():.1e}")
    bad = np.where(m)[0][~end_eq]
    if bad.size:
        print("   first differing rays (index, theta, bounces, n_steps oracle/hip):",
              [(int(k), round(float(theta[k]), 4), int(o["n_bott"][k] + o["n_surf"][k]), int(o["n_steps"][k]), int(g["n_steps"][k])) for k in bad[:6]])

if len(sys.argv) > 7 and sys.argv[7] == "default-form":
    del g
    d = env.shoot_fan(y0, 0.0, rmax, S, exact_bisection=exact, sample_major=True)
    for k in "Tzp":
        d[k] = d[k].T      # (sample-major [S][N] on the device, as bench.py runs it; compared ray-major)
    inside = (o["xi"] >= 0) & (o["xi"] <= 1)
    inside[:, -1] = False
    okc = ok[:, None]
    out_eq = all(np.array_equal(d[k][okc & ~inside], o[k][okc & ~inside]) for k in "Tzp")
    end_eq = np.array_equal(d["end"][ok], np.stack([o["T"][ok, -1], o["z"][ok, -1], o["p"][ok, -1]], 1))
    cnt_eq = all(np.array_equal(d[k][ok], o[k][ok]) for k in ("n_steps", "n_rej", "n_bott", "n_surf"))
    m = okc & inside
    print(f"default sample form (SAVE = 1 kernel, sample-major): end states bit-equal {end_eq}; step / bounce counts equal {cnt_eq}; "
          f"{int((okc & ~inside).sum())} samples outside a step (Q5, last column) bit-equal {out_eq}; {int(m.sum())} samples inside a step: "
          f"max rel dev T {np.abs(d['T'] - o['T'])[m].max() / np.nanmax(o['T'][ok]):.1e}  z {np.abs(d['z'] - o['z'])[m].max() / 5000.0:.1e}  "
          f"p {np.abs(d['p'] - o['p'])[m].max() * 1500.0:.1e}; bit-equal {np.mean((d['z'] == o['z'])[m]):.4f} of them")
