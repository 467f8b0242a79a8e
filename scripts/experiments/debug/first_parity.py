"""First GPU bring-up: HIP fan vs C oracle on golden-shaped inputs + a quick timing."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from pygenray_amd import _lib

def munk(z, zs=1300.0, eps=0.00737):
    zh = 2 * (z - zs) / zs
    return 1500 * (1 + eps * (zh - 1 + np.exp(-zh)))

def env_arrays(rmax, nr=100):
    z = np.arange(0, 6000, 1.0); r = np.linspace(0, rmax, nr)
    cin = np.tile(munk(z), (nr, 1)); cpin = np.gradient(cin, z, axis=1, edge_order=1)
    return [cin, cpin, r, z, np.full(nr, 5000.0), r.copy(), np.zeros(nr)]

def y0_of(arrs, zs, theta):
    c0 = oracle.bilinear(0.0, zs, arrs[2], arrs[3], arrs[0])
    return np.stack([np.zeros_like(theta), np.full_like(theta, zs), np.sin(np.radians(theta)) / c0], 1)

def compare(tag, g, o):
    ok = (o['status'] == 0)
    print(tag, "status equal:", np.array_equal(g['status'], o['status']), "nb/ns equal:",
          np.array_equal(g['n_bott'], o['n_bott']), np.array_equal(g['n_surf'], o['n_surf']),
          "nsteps equal:", np.array_equal(g['n_steps'], o['n_steps']), "max step diff", np.abs(g['n_steps'] - o['n_steps']).max())
    good = np.abs(o['xi']) <= 8
    for nm in 'Tzp':
        d = np.abs(g[nm] - o[nm]); d[~good] = 0
        print("   ", nm, "max abs err (well-conditioned samples):", np.nanmax(d[ok]), " end-state err:", np.nanmax(np.abs(g[nm][ok][:, -1] - o[nm][ok][:, -1])))
    print("    NaN pattern equal:", np.array_equal(np.isnan(g['z']), np.isnan(o['z'])))

for rmax, S, n in ((100e3, 101, 64), (1000e3, 101, 64)):
    arrs = env_arrays(rmax)
    theta = np.linspace(-20, 20, n)
    y0 = y0_of(arrs, 1000.0, theta)
    env = _lib.EnvHandle(*arrs)
    print("lds path:", env.lds_path, "range indep:", env.range_independent, "z uniform", env.query(1), "r uniform", env.query(2))
    t = time.time(); g = env.shoot_fan(y0, 0.0, rmax, S); tg = time.time() - t
    t = time.time(); o = oracle.shoot_fan(*arrs, y0, 0.0, rmax, S); to = time.time() - t
    print(f"rmax={rmax} gpu {tg:.3f}s oracle {to:.3f}s steps {g['n_steps'].sum()}")
    compare(f"munk {rmax/1e3:.0f}km", g, o)

# timing on the headline config (end-state only and with trajectories)
arrs = env_arrays(1000e3)
env = _lib.EnvHandle(*arrs)
for n in (10_000, 100_000):
    theta = np.linspace(-20, 20, n)
    y0 = y0_of(arrs, 1000.0, theta)
    for save in (False, True):
        t = time.time(); g = env.shoot_fan(y0, 0.0, 1000e3, 1001, save=save); dt = time.time() - t
        print(f"N={n} save={save}: {dt:.3f}s wall (incl. copies), steps={g['n_steps'].sum():.3e}, rej={g['n_rej'].sum():.3e}, "
              f"{g['n_steps'].sum()/dt:.3e} ray-steps/s, dropped={np.sum(g['status']!=0)}")
