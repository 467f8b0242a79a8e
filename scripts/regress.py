"""Bit-exact regression of the HIP fan kernel across refactorings.

python scripts/regress.py --save scripts/regress_ref.json     (on the GPU box, BEFORE a change)
python scripts/regress.py --check scripts/regress_ref.json    (after): every output array must hash the same

Restructuring the kernel (branches -> selects, scheduling, register use) must not change ONE bit of
any ray; the parity tests only bound the distance to the oracle."""
import sys, os, json, hashlib, argparse
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from pygenray_amd import _lib
from pygenray_amd.device_fan import fan_y0
from helpers import munk_arrays


def h(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


def cases():
    out = {}
    # (a) headline tables, every 10th ray of the fan, trajectories in both evaluation orders
    arrs = munk_arrays(1000e3)
    y0 = fan_y0(arrs, 1000.0, 0.0, -np.linspace(-20, 20, 100000)[::10])
    out["munk1000"] = (arrs, y0, 0.0, 1000e3, 101, {})
    out["munk1000_exact_samples"] = (arrs, y0[::4], 0.0, 1000e3, 101, dict(exact_samples=True))
    out["munk1000_exact_bisect"] = (arrs, y0[::8], 0.0, 1000e3, 11, dict(exact_bisection=True))
    # (b) range dependent (HBM table variant)
    arrs2 = munk_arrays(1000e3, nr=101, sofar_slope=2e-4)
    out["rangedep"] = (arrs2, fan_y0(arrs2, 1000.0, 0.0, -np.linspace(-20, 20, 3000)), 0.0, 1000e3, 51, {})
    # (c) sloping bottom + non-uniform depth grid + fine range grid (steps wider than a range cell)
    z = np.concatenate([np.linspace(0, 1000, 300), np.linspace(1000, 5500, 500)[1:]])
    r = np.linspace(0, 200e3, 801)  # 250 m cells
    c = 1500 * (1 + 0.00737 * ((2 * (z - 1300) / 1300) - 1 + np.exp(-(2 * (z - 1300) / 1300))))
    cin = np.tile(c, (len(r), 1)) * (1 + 2e-4 * np.sin(r / 30e3))[:, None]
    cpin = np.gradient(cin, z, axis=1)
    br = np.linspace(0, 200e3, 41)
    depths = 4800 + 400 * np.sin(br / 40e3)
    ba = np.degrees(np.arctan(np.gradient(depths, br)))
    arrs3 = (cin, cpin, r, z, depths, br, ba)
    out["slope_nonuniform_fine"] = (arrs3, fan_y0(arrs3, 800.0, 0.0, -np.linspace(-18, 18, 1500)), 0.0, 200e3, 64, {})
    # (d) the same with loose tolerance: long steps over many range cells
    out["slope_loose"] = (arrs3, fan_y0(arrs3, 800.0, 0.0, -np.linspace(-18, 18, 700)), 0.0, 200e3, 64, dict(rtol=1e-5))
    # (e) LDS table variant with a fine range grid (range independent, 100 m cells)
    arrs4 = munk_arrays(100e3, nr=1001)
    out["munk_fine_r"] = (arrs4, fan_y0(arrs4, 1000.0, 0.0, -np.linspace(-20, 20, 1000)), 0.0, 100e3, 40, dict(rtol=1e-6))
    # (f) the reference's default: flat-earth transformed tables (smoothly non-uniform zin), range
    # independent (LDS table) and range dependent (HBM table)
    import pygenray_amd as pr
    from pygenray_amd.environment import _unpack_envi
    zz = np.arange(0, 6000, 1.0); rr = np.linspace(0, 500e3, 51)
    for name, slope in (("flatearth_indep", 0.0), ("flatearth_rangedep", 2e-4)):
        c2 = np.array([pr.munk_ssp(zz, 1300 + slope * ri) for ri in rr])
        env = pr.OceanEnvironment2D(pr.DataArray(c2, dims=["range", "depth"], coords={"range": rr, "depth": zz}),
                                    pr.DataArray(np.full(51, 5000.0), dims=["range"], coords={"range": rr}),
                                    flat_earth_transform=True)
        arrs5 = _unpack_envi(env, flatearth=True)
        out[name] = (arrs5, fan_y0(arrs5, 1000.0, 0.0, -np.linspace(-20, 20, 2000)), 0.0, 500e3, 51, {})
    # (g) strongly non-uniform depth grid (fine near the surface, coarse at depth), range independent
    zg = np.concatenate([np.linspace(0, 200, 401), np.linspace(200, 5500, 531)[1:]])
    cg = np.tile(1500 * (1 + 0.00737 * ((2 * (zg - 1300) / 1300) - 1 + np.exp(-(2 * (zg - 1300) / 1300)))), (30, 1))
    rg = np.linspace(0, 300e3, 30)
    arrs6 = (cg, np.gradient(cg, zg, axis=1), rg, zg, np.full(30, 5200.0), rg.copy(), np.zeros(30))
    out["nonuniform_indep"] = (arrs6, fan_y0(arrs6, 300.0, 0.0, -np.linspace(-15, 15, 1000)), 0.0, 300e3, 40, {})
    return out


def run():
    res = {}
    for name, (arrs, y0, x0, x1, S, kw) in cases().items():
        env = _lib.EnvHandle(*arrs)
        o = env.shoot_fan(y0, x0, x1, S, sample_major=True, **kw)
        res[name] = {k: h(o[k]) for k in ("T", "z", "p", "end", "n_bott", "n_surf", "status", "n_steps", "n_rej")}
        res[name]["steps"] = int(o["n_steps"].sum()); res[name]["dropped"] = int((o["status"] != 0).sum())
        env.close()
    return res


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--save"); ap.add_argument("--check")
    ap.add_argument("--lib", default=None, help="a variant library (scripts/build_variants.py) instead of the product")
    a = ap.parse_args()
    if a.lib:
        _lib.LIB_PATH = os.path.abspath(a.lib)
    res = run()
    if a.save:
        json.dump(res, open(a.save, "w"), indent=1); print("saved", a.save)
    if a.check:
        ref = json.load(open(a.check)); bad = 0
        for name in ref:
            for k, v in ref[name].items():
                if res.get(name, {}).get(k) != v:
                    print("MISMATCH", name, k, v, res.get(name, {}).get(k)); bad += 1
        print("regression:", "OK (bit-identical)" if not bad else f"{bad} mismatches")
        sys.exit(1 if bad else 0)
    if not a.save and not a.check:
        print(json.dumps(res, indent=1))
