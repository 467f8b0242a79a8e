import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, oracle
from helpers import munk, y0_for
from pygenray_amd import _lib
rng = np.random.default_rng(11)
z = np.arange(0, 5500, 2.0); r = np.linspace(0, 150e3, 61)
cin = np.array([munk(z, 1300 + 2e-3 * ri) for ri in r]) + rng.normal(0, 0.05, (61, len(z))).cumsum(1) * 0.01
cpin = np.gradient(cin, z, axis=1, edge_order=1)
br = np.linspace(0, 150e3, 31); depths = 4800 + 300 * np.sin(br / 20e3)
ba = np.degrees(np.arctan(np.gradient(depths, br)))
arrs = [cin, cpin, r, z, depths, br, ba]
y0 = y0_for(oracle, arrs, 700.0, 5e3, np.linspace(-18, 18, 130))
o = oracle.shoot_fan(*arrs, y0, 5e3, 140e3, 136)
env = _lib.EnvHandle(*arrs)
for label, kw, park in (("default", {}, (64, 64)), ("exact", dict(exact_bisection=True), (64, 64)), ("nopark", {}, (1, 0))):
    env.set_option("park", *park)
    g = env.shoot_fan(y0, 5e3, 140e3, 136, **kw)
    d = np.abs(g["T"] - o["T"]); d[np.abs(o["xi"]) > 8] = 0
    bad = np.argwhere(d > 1e-5)
    print(label, "status eq", np.array_equal(g["status"], o["status"]), "nb eq", np.array_equal(g["n_bott"], o["n_bott"]), "ns eq", np.array_equal(g["n_surf"], o["n_surf"]), "n bad samples", len(bad), "rays", np.unique(bad[:, 0])[:20])
    for (i, j) in bad[:6]:
        print("   ray", i, "sample", j, "r", o["r"][j], "T gpu/orc", g["T"][i, j], o["T"][i, j], "z", g["z"][i, j], o["z"][i, j], "xi", o["xi"][i, j], "nb/ns", o["n_bott"][i], o["n_surf"][i], "steps", g["n_steps"][i], o["n_steps"][i])
env.set_option("park", 64, 64)
g = env.shoot_fan(y0, 5e3, 140e3, 136)
d = np.abs(g["T"] - o["T"]); d[np.abs(o["xi"]) > 8] = 0
i, j = np.unravel_index(np.nanargmax(d), d.shape)
print("worst", i, j, d[i, j], "xi", o["xi"][i, j-2:j+3], "T gpu", g["T"][i, j-2:j+3], "T orc", o["T"][i, j-2:j+3], "z gpu", g["z"][i, j-2:j+3], "z orc", o["z"][i, j-2:j+3])
print("nb/ns", o["n_bott"][i], o["n_surf"][i], g["n_bott"][i], g["n_surf"][i], "steps", g["n_steps"][i], o["n_steps"][i], "theta", np.linspace(-18, 18, 130)[i])
n1 = oracle.shoot_fan(*arrs, np.column_stack([y0[:, 0], y0[:, 1], np.nextafter(y0[:, 2], np.inf)]), 5e3, 140e3, 136)
print("oracle +1ulp: T", n1["T"][i, j-2:j+3], "xi", n1["xi"][i, j-2:j+3])
