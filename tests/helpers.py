"""Shared helpers for the parity tests (test infrastructure)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ENV_KEYS = ["cin", "cpin", "rin", "zin", "depths", "depth_ranges", "bottom_angles"]

# Parity policy.  Two comparisons, two rules:
#
# (A) HIP vs the CPU oracle in its correctly-rounded-libm mode (oracle.MATH_CR) -- the parity test
#     proper: BIT-IDENTICAL status, bounce counts, accepted / rejected step counts, end states and
#     (with PGR_EXACT_SAMPLES) every saved sample, Q5 extrapolated ones included: assert_bit_parity.
#     EVERY ray: no allowance (round 2 closed the last exception class, the kernel's 1/sqrt at
#     s = 1 - 2^-53; `max_odd` remains as a test parameter and defaults to 0).
#
# (B) HIP (or the oracle) vs vectors produced by the REFERENCE itself (tests/golden): the reference's
#     NumPy / SciPy arithmetic is not reproducible to the bit outside its own process (BLAS dot
#     products fuse and reorder, libm is faithful but not correctly rounded) and its adaptive
#     controller amplifies a last-bit change chaotically, so this comparison is statistical:
#     a ray passes if its deviation is <= REL_TOL x scale (north_star: (r, z, tau) within 1e-8
#     relative; scale = water-column depth for z, arrival time for T, 1/c for p; r is exact) or
#     <= NOISE_FACTOR x its OWN spread over the oracle's 1-ulp perturbed runs; and the MEDIAN
#     deviation of each class of rays -- never touching a boundary / bouncing -- must meet REL_TOL
#     itself (classes of >= 8 rays; `strict_bouncing=False` for the coarse or kinked grids on which
#     the reference's own test tolerances apply, abs_floor).  No class-wide allowance.
#     Samples the reference produces by extrapolating a quartic more than XI_MAX step lengths (Q5)
#     amplify rounding by xi^4 and are compared with a tolerance scaled by that.
REL_TOL = 1e-8
NOISE_FACTOR = 10.0   # SURVEY section 7 stage 2: 10 x the measured self-noise
XI_MAX = 8.0
ODD_REL_TOL = 1e-5   # (only where a test passes max_odd > 0)


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


def oracle_selfnoise(oracle, arrs, y0, x0, x1, S, ulps=(1,), **kw):
    """The oracle re-run under k-ulp perturbations of p0 (both ways, k in `ulps`; default 1 ulp) and 1-ulp perturbations of
    rtol (both ways).  (`ulps=(1, 2, 3)` is how the reference's own self-noise was sampled for the golden vectors g11-g13,
    `selfnoise_end`: seven end states per ray.)"""
    outs = []
    for k in ulps:
        for d in (-1, 1):
            y = y0.copy()
            for _ in range(int(k)):
                y[:, 2] = np.nextafter(y[:, 2], d * np.inf)
            outs.append(oracle.shoot_fan(*arrs, y, x0, x1, S, **kw))
    rt = kw.get("rtol", 1e-9)
    kw2 = {k: v for k, v in kw.items() if k != "rtol"}
    for d in (0.0, 1.0):
        outs.append(oracle.shoot_fan(*arrs, y0, x0, x1, S, rtol=float(np.nextafter(rt, d)), **kw2))
    return outs


def assert_bit_parity(test, ref, label="", samples=True, max_odd=None):
    """(A): test (HIP) against ref (oracle.MATH_CR run of the same inputs), bit for bit.  `samples`:
    also every saved sample (the HIP fan was shot with exact_samples=True); otherwise the samples a
    step evaluates inside itself (0 <= xi <= 1, stage-major FMA form by default) may differ by
    rounding (1e-12 x scale) and all others must still be equal."""
    assert np.array_equal(test["status"], ref["status"]), \
        f"{label}: status differs at {np.where(test['status'] != ref['status'])[0][:10]}"
    ok = ref["status"] == 0
    for k in ("n_bott", "n_surf"):
        assert np.array_equal(test[k][ok], ref[k][ok]), f"{label}: {k}"
    for nm in "Tzp":
        assert np.array_equal(np.isnan(test[nm]), np.isnan(ref[nm])), f"{label}: NaN pattern of {nm}"
    if not ok.any():
        return dict(n=0, odd=0)
    end_ref = np.stack([ref["T"][:, -1], ref["z"][:, -1], ref["p"][:, -1]], 1)
    same = np.all(test["end"] == end_ref, axis=1) | ~ok
    same &= (test["n_steps"].astype(np.int64) == ref["n_steps"]) | ~ok
    if test.get("n_rej") is not None and ref.get("n_rej") is not None:
        same &= (test["n_rej"].astype(np.int64) == ref["n_rej"]) | ~ok
    xi = ref["xi"]
    inside = (xi >= 0) & (xi <= 1)
    inside[:, -1] = False
    for nm, scale in (("T", max(float(np.nanmax(np.abs(ref["T"][ok]))), 1e-9)), ("z", max(float(np.nanmax(np.abs(ref["z"][ok]))), 1.0)),
                      ("p", 1.0 / 1400.0)):
        eq = (test[nm] == ref[nm]) | np.isnan(ref[nm])
        if not samples:
            with np.errstate(invalid="ignore"):
                eq |= inside & (np.abs(test[nm] - ref[nm]) <= 1e-12 * scale)
        same &= np.all(eq, axis=1) | ~ok
    odd = np.where(~same)[0]
    n = int(ok.sum())
    allowed = 0 if max_odd is None else max_odd
    assert len(odd) <= allowed, (f"{label}: {len(odd)} of {n} rays are not bit-identical to the oracle "
                                 f"(allowed {allowed}); first {odd[:8]}")
    for k in odd:   # the odd ones out: the same ray all the same
        for j, scale in enumerate((max(float(end_ref[k, 0]), 1e-9), max(float(np.nanmax(np.abs(ref["z"][ok]))), 1.0), 1 / 1400.0)):
            assert abs(test["end"][k, j] - end_ref[k, j]) <= ODD_REL_TOL * scale, (label, int(k), j)
    return dict(n=n, odd=len(odd))


def assert_fan_parity(test, ref, noise_runs=None, scales=None, label="", abs_floor=None, strict_bouncing=True):
    """(B): test/ref: dicts with T,z,p (N,S), n_bott, n_surf, status; ref also has xi."""
    assert np.array_equal(test["status"], ref["status"]), \
        f"{label}: status differs at {np.where(test['status'] != ref['status'])[0][:10]}"
    ok = ref["status"] == 0
    assert np.array_equal(test["n_bott"][ok], ref["n_bott"][ok]), f"{label}: bottom bounce counts"
    assert np.array_equal(test["n_surf"][ok], ref["n_surf"][ok]), f"{label}: surface bounce counts"
    for nm in "Tzp":
        assert np.array_equal(np.isnan(test[nm]), np.isnan(ref[nm])), f"{label}: NaN pattern of {nm}"
    if not ok.any():
        return {}
    xi = np.abs(ref["xi"])
    if noise_runs is not None:
        # a bounce a hair to the other side of a save point flips which quartic owns the
        # sample (Q5): the largest |xi| over the oracle runs prices the sample
        for nr_ in noise_runs:
            if nr_.get("xi") is not None:
                xi = np.maximum(xi, np.abs(nr_["xi"]))
    good = xi <= XI_MAX
    good[:, -1] = True
    # Q5 samples: a quartic extrapolated xi step lengths amplifies rounding by xi^4
    amp = np.where(good, 1.0, np.minimum(np.maximum(xi, 1.0) ** 4, 1e12))
    amp[:, -1] = 1.0
    zscale, tscale, pscale = scales if scales else (5000.0, np.nanmax(ref["T"][ok]), 1.0 / 1500.0)
    quiet = ((ref["n_bott"] + ref["n_surf"]) == 0)[ok]
    worst = {}
    for nm, scale in (("T", tscale), ("z", zscale), ("p", pscale)):
        dfull = np.abs(test[nm] - ref[nm]) / amp
        d = np.nan_to_num(np.where(good, dfull, 0.0)[ok], nan=np.inf).max(1)       # per ray, interior samples
        dq5 = np.nan_to_num(np.where(good, 0.0, dfull)[ok], nan=np.inf).max(1)      # per ray, Q5 samples / xi^4
        tol = np.full(d.shape[0], REL_TOL * scale)
        if abs_floor is not None:  # coarse-grid cases: the reference's own test tolerances
            tol = np.maximum(tol, abs_floor[nm])
        base_tol = tol.copy()
        if noise_runs is not None:
            spread = np.zeros(d.shape[0])
            for nr_ in noise_runs:
                okn = (nr_["status"] == 0)[ok]
                s = np.abs(nr_[nm] - ref[nm])
                s = np.nan_to_num(np.where(good, s, 0.0)[ok], nan=np.inf).max(1)
                s[~okn] = np.inf
                spread = np.maximum(spread, s)
            tol = np.maximum(tol, NOISE_FACTOR * spread)
            # the worst deviation / self-noise ratio among the rays that NEED the noise rule (those beyond REL_TOL x scale)
            needs = d > base_tol
            with np.errstate(divide="ignore", invalid="ignore"):
                ratio = np.where(needs, d / spread, 0.0)
            worst[nm + "_noise_ratio"] = float(np.nanmax(ratio)) if needs.any() else 0.0
            worst[nm + "_rays_on_noise_rule"] = int(needs.sum())
        bad = d > tol
        worst[nm] = float(np.nanmax(d) / scale)
        assert not bad.any(), (f"{label}: {nm} differs: worst {np.nanmax(d):.3e} (rel {worst[nm]:.2e}), "
                               f"{bad.sum()} rays beyond tolerance, first {np.where(bad)[0][:5]}")
        badq = dq5 > np.maximum(tol, 1e-6 * scale)
        assert not badq.any(), f"{label}: extrapolated (Q5) samples of {nm} differ beyond xi^4 x tolerance: rays {np.where(badq)[0][:5]}"
        floor = REL_TOL * scale if abs_floor is None else max(REL_TOL * scale, abs_floor[nm])
        for cls, name in ((quiet, "non-bouncing"), (~quiet, "bouncing")):
            if cls.sum() >= 8 and (name == "non-bouncing" or strict_bouncing):
                med = float(np.median(d[cls]))
                assert med <= floor, f"{label}: median {nm} deviation of {name} rays {med:.3e} > {floor:.3e}"
                worst[nm + "_median_" + name] = med / scale
    return worst


def random_case(seed, n_rays=128):
    """A random smooth environment + shot for the bit-parity sweep (scripts/fuzz_bitparity.py and
    tests/test_hip_parity.py): uniform / power-of-two / stretched depth grids, uniform / random range
    grids starting anywhere (also negative), flat / sloping floors, range (in)dependent sound speed,
    forward frames and the mirrored frame of a backwards shot, random source range, tolerance, save grid.
    Returns (arrs, y0_args, shot kwargs, description)."""
    rng = np.random.default_rng(5000 + seed)
    zmax = rng.uniform(1500, 6000)
    nz = int(rng.integers(150, 2500))
    kind_z = int(rng.integers(0, 3))
    if kind_z == 0:
        z = np.linspace(0, zmax, nz)
    elif kind_z == 1:
        z = np.arange(0, zmax, 2.0 ** rng.integers(-1, 3))
    else:
        z = zmax * np.linspace(0, 1, nz) ** rng.uniform(1.0, 1.6)
    rmax = rng.uniform(30e3, 400e3)
    x_off = 0.0 if rng.random() < 0.5 else rng.uniform(-500e3, 500e3)     # tables need not start at range 0
    nr = int(rng.integers(3, 120))
    r = x_off + (np.linspace(0, rmax, nr) if rng.random() < 0.6 else np.sort(np.concatenate([[0, rmax], rng.uniform(0, rmax, nr - 2)])))
    slope = 0.0 if rng.random() < 0.4 else rng.uniform(-2e-3, 2e-3)
    axis = rng.uniform(0.15, 0.5) * zmax
    cin = np.array([munk(z, axis + slope * (ri - x_off)) for ri in r]) if slope else np.tile(munk(z, axis), (len(r), 1))
    cpin = np.gradient(cin, z, axis=1, edge_order=1)
    nb = int(rng.integers(4, 60))
    br = x_off + (np.linspace(0, rmax, nb) if rng.random() < 0.5 else np.sort(np.concatenate([[0, rmax], rng.uniform(0, rmax, nb - 2)])))
    floor = rng.uniform(0.7, 0.95) * zmax
    depths = np.full(nb, floor) if rng.random() < 0.4 else floor + rng.uniform(0, 0.04) * zmax * np.sin((br - x_off) / rng.uniform(20e3, 90e3))
    ba = np.degrees(np.arctan(np.gradient(depths, br)))
    arrs = [cin, cpin, r, z, depths, br, ba]
    mirrored = bool(rng.random() < 0.3)
    if mirrored:   # the mirrored frame of a backwards shot (REF/launch_rays.py:684-714): negative, reversed ranges
        arrs = [np.ascontiguousarray(cin[::-1]), np.ascontiguousarray(cpin[::-1]), -r[::-1], z,
                np.ascontiguousarray(depths[::-1]), -br[::-1], -ba[::-1]]
    rr = arrs[2]
    x0 = rr[0] + rng.uniform(0.0, 0.3) * (rr[-1] - rr[0]) * (rng.random() < 0.5)
    x1 = x0 + rng.uniform(0.3, 1.0) * (rr[-1] - x0)
    src = rng.uniform(0.05, 0.6) * zmax
    rtol = [1e-9, 1e-7, 1e-5][int(rng.integers(0, 3))]
    th = np.linspace(-rng.uniform(5, 30), rng.uniform(5, 30), n_rays)
    S = int(rng.integers(2, 60))
    tb = bool(rng.random() < 0.8)
    desc = (f"nz {len(z):5d} ({['uniform', 'pow2', 'stretched'][kind_z]}) nr {len(r):3d} nb {nb:2d} rtol {rtol:g} S {S:2d} "
            f"range-dep {bool(slope)!s:5s} mirrored {mirrored!s:5s} x0 {x0:10.0f} x1 {x1:10.0f}")
    return arrs, (src, x0, th), dict(x0=x0, x1=x1, S=S, rtol=rtol, terminate_backwards=tb), desc
