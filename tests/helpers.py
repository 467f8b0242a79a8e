"""Shared helpers for the parity tests (test infrastructure)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ENV_KEYS = ["cin", "cpin", "rin", "zin", "depths", "depth_ranges", "bottom_angles"]

# Parity policy (DESIGN.md "Parity"): (r, z, tau) within REL_TOL relative of the CPU reference,
# relative to the field scale (water-column depth for z, arrival time for T, 1/c for p), OR
# within NOISE_FACTOR x the reference integrator's own spread under a +-1 ulp perturbation of
# p0 (the adaptive controller on the kinked bilinear c(z) amplifies last-bit differences on
# multi-bounce rays far beyond 1e-8: SURVEY.md section 0).  Samples the reference itself
# produces by extrapolating a quartic more than XI_MAX step lengths (Q5) are only required
# to exist, not to agree.
REL_TOL = 1e-8
NOISE_FACTOR = 20.0
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
    """Spread of the oracle's own output under +-1 ulp perturbations of p0: (N,) arrays for
    T, z, p (max over well-conditioned samples)."""
    outs = []
    for d in (-1, 1):
        y = y0.copy()
        y[:, 2] = np.nextafter(y[:, 2], d * np.inf)
        outs.append(oracle.shoot_fan(*arrs, y, x0, x1, S, **kw))
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
    worst = {}
    for nm, scale in (("T", tscale), ("z", zscale), ("p", pscale)):
        d = np.abs(test[nm] - ref[nm])
        d = np.where(good, d, 0.0)[ok]
        tol = np.full(d.shape[0], REL_TOL * scale)
        if abs_floor is not None:  # coarse-grid cases: the reference's own test tolerances
            tol = np.maximum(tol, abs_floor[nm])
        if noise_runs is not None:
            spread = np.zeros(d.shape)
            for nr_ in noise_runs:
                okn = (nr_["status"] == 0)[ok]
                s = np.abs(nr_[nm] - ref[nm])
                s = np.where(good, s, 0.0)[ok]
                s[~okn] = np.inf
                spread = np.maximum(spread, np.nan_to_num(s, nan=np.inf))
            tol = np.maximum(tol[:, None], NOISE_FACTOR * spread)
            bad = d > tol
        else:
            bad = d > tol[:, None]
        worst[nm] = float(np.nanmax(d) / scale)
        assert not bad.any(), (f"{label}: {nm} differs: worst {np.nanmax(d):.3e} (rel {worst[nm]:.2e}) "
                               f"at ray {np.argwhere(bad)[0]}")
    return worst
