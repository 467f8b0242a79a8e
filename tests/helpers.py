"""Shared helpers for the parity tests (test infrastructure)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ENV_KEYS = ["cin", "cpin", "rin", "zin", "depths", "depth_ranges", "bottom_angles"]

# Parity policy (DESIGN.md "Parity").  north_star: (r, z, tau) within REL_TOL relative of the
# CPU reference, relative to the field scale (water-column depth for z, arrival time for T,
# 1/c for p; r is exact).  The reference integrator is, however, chaotic at the last bit: the
# embedded error estimate is a near-cancelling sum whose rounding noise (1e-8 relative) is fed
# back into every later step size, so a 1-ulp change of ANY input (p0, rtol, the result of
# pow) moves a random ~10 % of the rays -- bouncing or not -- by up to millimetres at 1000 km
# while the rest agree to 1e-10 (measured with the reference itself, golden g3, and with the
# oracle below).  So a ray passes if its deviation is
#   <= REL_TOL x scale                                 (the north-star bound), or
#   <= NOISE_FACTOR x its own spread over the oracle's 1-ulp perturbed runs, or
#   <= ENSEMBLE_FACTOR x the largest such spread among rays of its class (bouncing / not),
# and, so that the ensemble clause cannot hide a systematic error, the MEDIAN deviation of the
# rays that never touch a boundary must meet REL_TOL itself (when there are >= 8 of them).
# Samples the reference itself produces by extrapolating a quartic more than XI_MAX step
# lengths (Q5) are only required to exist, not to agree.
REL_TOL = 1e-8
NOISE_FACTOR = 20.0
ENSEMBLE_FACTOR = 3.0
XI_MAX = 8.0


def load(name):
    return np.load(os.path.join(GOLDEN, name))


def env_from(g, prefix="env_"):
    return [g[prefix + k] for k in ENV_KEYS]


def tiled_env(g):
    nr = len(g["rin"])
    return [np.tile(g["c_row"], (nr, 1)), np.tile(g["cp_row"], (nr, 1)), g["rin"], g["zin"],
            g["depths"], g["depth_ranges"], g["bottom_angles"]]


def munk(z, sofar=1300.0, eps=0.00737):
    zh = 2 * (z - sofar) / sofar
    return 1500 * (1 + eps * (zh - 1 + np.exp(-zh)))


def munk_arrays(r_max, nr=100, z=None, bathy=5000.0, sofar_slope=0.0):
    z = np.arange(0, 6000, 1.0) if z is None else z
    r = np.linspace(0.0, r_max, nr)
    if sofar_slope:
        cin = np.array([munk(z, 1300 + sofar_slope * ri) for ri in r])
    else:
        cin = np.tile(munk(z), (nr, 1))
    cpin = np.gradient(cin, z, axis=1, edge_order=1)
    return [cin, cpin, r, z, np.full(nr, float(bathy)), r.copy(), np.zeros(nr)]


def y0_for(oracle, arrs, source_depth, source_range, theta_ode):
    c0 = oracle.bilinear(source_range, source_depth, arrs[2], arrs[3], arrs[0])
    th = np.asarray(theta_ode, float)
    return np.stack([np.zeros_like(th), np.full_like(th, source_depth), np.sin(np.radians(th)) / c0], 1)


def oracle_selfnoise(oracle, arrs, y0, x0, x1, S, **kw):
    """The oracle re-run under 1-ulp perturbations of p0 (both ways) and of rtol (both ways)."""
    outs = []
    for d in (-1, 1):
        y = y0.copy()
        y[:, 2] = np.nextafter(y[:, 2], d * np.inf)
        outs.append(oracle.shoot_fan(*arrs, y, x0, x1, S, **kw))
    rt = kw.get("rtol", 1e-9)
    kw2 = {k: v for k, v in kw.items() if k != "rtol"}
    for d in (0.0, 1.0):
        outs.append(oracle.shoot_fan(*arrs, y0, x0, x1, S, rtol=float(np.nextafter(rt, d)), **kw2))
    return outs


def assert_fan_parity(test, ref, noise_runs=None, scales=None, label="", abs_floor=None):
    """test/ref: dicts with T,z,p (N,S), n_bott, n_surf, status; ref also has xi."""
    assert np.array_equal(test["status"], ref["status"]), \
        f"{label}: status differs at {np.where(test['status'] != ref['status'])[0][:10]}"
    ok = ref["status"] == 0
    assert np.array_equal(test["n_bott"][ok], ref["n_bott"][ok]), f"{label}: bottom bounce counts"
    assert np.array_equal(test["n_surf"][ok], ref["n_surf"][ok]), f"{label}: surface bounce counts"
    for nm in "Tzp":
        assert np.array_equal(np.isnan(test[nm]), np.isnan(ref[nm])), f"{label}: NaN pattern of {nm}"
    if not ok.any():
        return {}
    good = np.abs(ref["xi"]) <= XI_MAX
    if noise_runs is not None:
        # a bounce a hair to the other side of a save point flips which quartic owns the
        # sample (Q5): only samples that are interior in every oracle run are compared
        for nr_ in noise_runs:
            if nr_.get("xi") is not None:
                good &= np.abs(nr_["xi"]) <= XI_MAX
    good[:, -1] = True
    zscale, tscale, pscale = scales if scales else (5000.0, np.nanmax(ref["T"][ok]), 1.0 / 1500.0)
    quiet = ((ref["n_bott"] + ref["n_surf"]) == 0)[ok]
    worst = {}
    for nm, scale in (("T", tscale), ("z", zscale), ("p", pscale)):
        d = np.abs(test[nm] - ref[nm])
        d = np.nan_to_num(np.where(good, d, 0.0)[ok], nan=np.inf).max(1)       # per ray
        tol = np.full(d.shape[0], REL_TOL * scale)
        if abs_floor is not None:  # coarse-grid cases: the reference's own test tolerances
            tol = np.maximum(tol, abs_floor[nm])
        if noise_runs is not None:
            spread = np.zeros(d.shape[0])
            for nr_ in noise_runs:
                okn = (nr_["status"] == 0)[ok]
                s = np.abs(nr_[nm] - ref[nm])
                s = np.nan_to_num(np.where(good, s, 0.0)[ok], nan=np.inf).max(1)
                s[~okn] = np.inf
                spread = np.maximum(spread, s)
            tol = np.maximum(tol, NOISE_FACTOR * spread)
            for cls in (quiet, ~quiet):
                fin = cls & np.isfinite(spread)
                if fin.any():
                    tol[cls] = np.maximum(tol[cls], ENSEMBLE_FACTOR * spread[fin].max())
        bad = d > tol
        worst[nm] = float(np.nanmax(d) / scale)
        assert not bad.any(), (f"{label}: {nm} differs: worst {np.nanmax(d):.3e} (rel {worst[nm]:.2e}), "
                               f"{bad.sum()} rays beyond tolerance, first {np.where(bad)[0][:5]}")
        if quiet.sum() >= 8:
            med = float(np.median(d[quiet]))
            floor = REL_TOL * scale if abs_floor is None else max(REL_TOL * scale, abs_floor[nm])
            assert med <= floor, f"{label}: median {nm} deviation of non-bouncing rays {med:.3e} > {floor:.3e}"
            worst[nm + "_median_quiet"] = med / scale
    return worst
