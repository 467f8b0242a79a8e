"""Pin the CPU oracle (C restatement + SciPy port) to the reference.

Golden vectors come from running pygenray itself (tests/golden/make_golden.py) plus the
reference's own committed fixture tests/fixtures/munk_regression.npz (copied as data to
tests/golden/ref_munk_regression.npz).  These tests run without a GPU.
"""
import numpy as np
import pytest
import scipy.interpolate
from scipy.optimize import brentq

import oracle
from oracle import scipy_port
from helpers import (load, env_from, tiled_env, assert_fan_parity, oracle_selfnoise, XI_MAX, NOISE_FACTOR, REL_TOL, munk_arrays)


def golden_as_ref(g, prefix="", xi=None):
    ok = g[prefix + "ok"].astype(bool)
    ref = dict(T=g[prefix + "T"], z=g[prefix + "z"], p=g[prefix + "p"], n_bott=g[prefix + "n_bott"],
               n_surf=g[prefix + "n_surf"], status=np.where(ok, 0, -1), xi=xi)
    return ref


def check_against_golden(g, arrs, x0, x1, S, prefix="", label="", step_slack=0.002, abs_floor=None, **kw):
    out = oracle.shoot_fan(*arrs, g[prefix + "y0"], x0, x1, S, **kw)
    ref = golden_as_ref(g, prefix, xi=out["xi"])
    # dropped rays: golden only records ok / not ok
    test = dict(out)
    test["status"] = np.where(out["status"] == 0, 0, -1)
    noise = oracle_selfnoise(oracle, arrs, g[prefix + "y0"], x0, x1, S, **kw)
    for n in noise:
        n["status"] = np.where(n["status"] == 0, 0, -1)
    zs = float(arrs[3][-1])
    worst = assert_fan_parity(test, ref, noise_runs=noise, scales=(zs, max(np.nanmax(ref["T"]), 1e-9), 1 / 1500.0),
                              label=label, abs_floor=abs_floor)
    okm = g[prefix + "ok"].astype(bool)
    # accepted-step counts: the adaptive controller follows SciPy's decisions exactly except
    # where a last-bit difference flips an accept/reject (coarse, kinked grids: a few %)
    dsteps = np.abs(out["n_steps"][okm] - g[prefix + "n_steps"][okm])
    assert np.all(dsteps <= np.maximum(2, step_slack * g[prefix + "n_steps"][okm])), dsteps
    return out, worst


# ----------------------------------------------------------------------------- unit level (G7)
def test_unit_vectors_bilinear_linear_derivs_events():
    g = load("g7_unit_vectors.npz")
    cin, cpin, rin, zin = g["cin"], g["cpin"], g["rin"], g["zin"]
    for k in range(len(g["xs"])):
        x, y = float(g["xs"][k]), g["ys"][k]
        b = oracle.bilinear(x, y[1], rin, zin, cin)
        assert b == g["bilinear"][k] or abs(b - g["bilinear"][k]) <= 4e-16 * abs(b)
        assert oracle.linear(x, g["depth_ranges"], g["depths"]) == pytest.approx(g["linear"][k], rel=4e-16, abs=0)
        d = oracle.derivs(x, y, cin, cpin, rin, zin)
        np.testing.assert_allclose(d, g["derivs"][k], rtol=1e-14, atol=0)
        th, c = oracle.ray_angle(x, y, cin, rin, zin)
        if np.isnan(g["angle"][k, 0]):
            assert np.isnan(th)
        else:
            assert th == pytest.approx(g["angle"][k, 0], rel=1e-14, abs=1e-13)
        assert c == pytest.approx(g["angle"][k, 1], rel=4e-16)
        ev = oracle.events(x, y, cin, rin, zin, g["depths"], g["depth_ranges"])
        np.testing.assert_array_equal(ev, g["events"][k])
    # the SciPy port's NumPy RHS restates the same functions
    tb = scipy_port.Tables(cin, cpin, rin, zin, g["depths"], g["depth_ranges"], np.zeros(len(g["depths"])))
    for k in range(0, len(g["xs"]), 7):
        np.testing.assert_allclose(tb.rhs(float(g["xs"][k]), g["ys"][k]), g["derivs"][k], rtol=1e-14)
        assert [tb.ev_surface(g["xs"][k], g["ys"][k]), tb.ev_bottom(g["xs"][k], g["ys"][k]),
                tb.ev_vertical(g["xs"][k], g["ys"][k]), tb.ev_bbox(g["xs"][k], g["ys"][k])] == list(g["events"][k])


def test_extrapolation_quirk_q4():
    # index clamped, weight not -> linear extrapolation outside the grid
    v = np.array([[1., 2, 3], [4, 5, 6], [7, 8, 9]])
    gr = np.array([0., 1, 2])
    assert oracle.bilinear(3.0, 0.0, gr, gr, v) == 10.0
    assert oracle.bilinear(0.5, 0.5, gr, gr, v) == 3.0
    assert oracle.linear(0.5, gr, np.array([1., 4, 7])) == 2.5


def test_brentq_transcription_matches_scipy():
    rng = np.random.default_rng(3)
    eps = np.finfo(float).eps
    for _ in range(300):
        a = rng.uniform(0, 1e6)
        b = a + rng.uniform(1e-3, 2e3)
        s = a + rng.uniform(0, 1) * (b - a)
        cnt = [0]

        def f(x):
            cnt[0] += 1
            return 1.0 if x > s else -1.0
        r = brentq(f, a, b, xtol=4 * eps, rtol=4 * eps)
        r2, n2 = oracle.brentq_step(a, b, s)
        assert r == r2 and cnt[0] == n2


def test_bottom_angle_notaknot_matches_interp1d():
    rng = np.random.default_rng(5)
    for n in (4, 5, 9, 100):
        x = np.sort(rng.uniform(0, 100e3, n))
        x[0], x[-1] = 0.0, 100e3
        y = rng.uniform(-3, 3, n)
        xq = np.concatenate([x, rng.uniform(0, 100e3, 200)])
        want = scipy.interpolate.interp1d(x, y, kind="cubic")(xq)
        got = oracle.bottom_angle_interp(x, y, xq)
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-9 * max(1.0, np.abs(y).max()))
    assert np.all(oracle.bottom_angle_interp(np.linspace(0, 1, 8), np.zeros(8), np.linspace(0, 1, 50)) == 0.0)
    with pytest.raises(ValueError):
        oracle.bottom_angle_interp(np.arange(3.0), np.zeros(3), [0.5])


# ----------------------------------------------------------------------------- the reference's own fixture
def test_reference_fixture_munk_regression():
    """tests/test_physics.py:310-386 of the reference, against ITS committed fixture
    (tolerances of the reference's own test: zs 0.1 m, ps 0.1, bounce counts exact; ts is held
    to 5e-6 s instead of 1e-6 s because the reference re-run in this container -- SciPy 1.15.3,
    no numba fastmath -- itself differs from its fixture by 3.1e-6 s, see g1 below)."""
    g = load("g1_fixture_case.npz")
    ref = load("ref_munk_regression.npz")
    arrs = env_from(g)
    c0 = oracle.bilinear(0.0, 1300.0, arrs[2], arrs[3], arrs[0])
    y0 = np.array([[0, 1300.0, np.sin(np.radians(a)) / c0] for a in ref["thetas"]])
    out = oracle.shoot_fan(*arrs, y0, 0.0, 50e3, 50)
    assert np.all(out["status"] == 0)
    np.testing.assert_allclose(out["T"], ref["ts"], atol=5e-6)
    np.testing.assert_allclose(-out["z"], ref["zs"], atol=0.1)
    np.testing.assert_allclose(-out["p"], ref["ps"], atol=0.1)
    np.testing.assert_array_equal(out["n_bott"], ref["n_botts"])
    np.testing.assert_array_equal(out["n_surf"], ref["n_surfs"])
    # and against the same case regenerated from the reference here, within 3x its own
    # +-1 ulp self-noise on this coarse (dz = 15 m) grid
    assert np.all(np.abs(out["T"] - g["ts"]).max(1) <= 3 * g["selfnoise_t"] + 1e-9)
    assert np.all(np.abs(-out["z"] - g["zs"]).max(1) <= 3 * g["selfnoise_z"] + 1e-6)
    # the committed fixture vs the regenerated reference: documents the reference's own drift
    assert np.abs(g["ts"] - ref["ts"]).max() < 5e-6


# ----------------------------------------------------------------------------- full rays (G2-G5)
def test_munk_100km_config0_shape():
    g = load("g2_munk_100km.npz")
    out, worst = check_against_golden(g, tiled_env(g), 0.0, 100e3, 101, label="g2")
    assert np.array_equal(out["nfev"], g["nfev"])
    quiet = (g["n_bott"] + g["n_surf"]) == 0
    assert np.abs(out["z"][quiet] - g["z"][quiet]).max() / 5000 < 1e-8
    assert np.abs(out["T"][quiet] - g["T"][quiet]).max() / 67.0 < 1e-9


def test_munk_1000km_config1_subset():
    g = load("g3_munk_1000km.npz")
    out, worst = check_against_golden(g, tiled_env(g), 0.0, 1000e3, 101, label="g3")
    # end states within the reference's recorded +-1 ulp self-noise (x10) or 1e-8 relative
    end = np.stack([out["T"][:, -1], out["z"][:, -1], out["p"][:, -1]], 1)
    gend = np.stack([g["T"][:, -1], g["z"][:, -1], g["p"][:, -1]], 1)
    tol = np.maximum(NOISE_FACTOR * g["selfnoise_end"], 1e-8 * np.array([670.0, 6000.0, 1 / 1500.0]))
    assert np.all(np.abs(end - gend) <= tol)
    quiet = (g["n_bott"] + g["n_surf"]) == 0
    assert np.abs(end - gend)[quiet][:, 1].max() / 5000 < 1e-8


def test_range_dependent_forward_and_mirrored():
    g = load("g4_range_dependent.npz")
    arrs = env_from(g)
    # coarse dz = 15 m grid + sloping bottom: the reference's own tolerances for this case
    # (tests/test_physics.py:549-551: z atol 1e-2, t atol 1e-6)
    floor = dict(T=1e-6, z=1e-2, p=1e-7)
    check_against_golden(g, arrs, 10e3, 90e3, 81, prefix="fwd_", label="g4 fwd", step_slack=0.04,
                         abs_floor=floor)
    # mirrored environment (launch_rays.py:684-714)
    arrs_m = [np.ascontiguousarray(arrs[0][::-1]), np.ascontiguousarray(arrs[1][::-1]), -arrs[2][::-1],
              arrs[3], np.ascontiguousarray(arrs[4][::-1]), -arrs[5][::-1], -arrs[6][::-1]]
    check_against_golden(g, arrs_m, -60e3, -10e3, 80, prefix="bwd_", label="g4 bwd", step_slack=0.04,
                         abs_floor=floor)


def test_irregular_range_bathymetry_and_depth_grids():
    """g9: randomly spaced rin (50 m ... 3 km cells) and bathymetry ranges, stretched zin,
    range-dependent c: every table look-up is a search; at rtol 1e-5 one step spans several cells."""
    g = load("g9_irregular_grids.npz")
    arrs = env_from(g)
    # rtol 1e-9 on a grid with kinks every few metres: accept/reject decisions sit on the last bit, the
    # step counts differ by one on some rays and the 1-ulp self-noise bound carries the comparison
    check_against_golden(g, arrs, 1e3, 69e3, 61, prefix="t9_", rtol=1e-9, label="g9 rtol 1e-9", step_slack=0.04)
    _, worst = check_against_golden(g, arrs, 1e3, 69e3, 61, prefix="t5_", rtol=1e-5, label="g9 rtol 1e-5")
    assert max(worst.values()) < 1e-9, worst


def test_range_dependent_config2_subset():
    g = load("g4_config2_subset.npz")
    from helpers import munk
    z = np.arange(0, 6000, 1.0)
    r = np.linspace(0.0, float(g["r_max"]), int(g["nr"]))
    cin = np.array([munk(z, 1300 + float(g["sofar_slope"]) * ri) for ri in r])
    cpin = np.gradient(cin, z, axis=1, edge_order=1)
    assert np.sum(cin) == pytest.approx(float(g["c_checksum"]), rel=1e-15)
    assert np.sum(cpin) == pytest.approx(float(g["cp_checksum"]), rel=1e-12)
    arrs = [cin, cpin, r, z, np.full(len(r), 5000.0), r.copy(), np.zeros(len(r))]
    check_against_golden(g, arrs, 0.0, 1000e3, 101, label="g4 config2")


def test_constant_c_and_steep_rays():
    g = load("g5_const_c.npz")
    check_against_golden(g, env_from(g), 0.0, 30e3, 60, label="g5 const c")
    g = load("g5_const_c_steep.npz")
    check_against_golden(g, env_from(g), 0.0, 1.5e3, 31, rtol=float(g["rtol"]), label="g5 steep")


def test_linear_gradient_and_flatearth_nonuniform_grid():
    g = load("g5_linear_gradient.npz")
    arrs = env_from(g)
    out = oracle.shoot_fan(*arrs, g["y0"], 0.0, 80e3, 400)
    for nm in "Tzp":
        out[nm] = out[nm][:, ::4]
    out["xi"] = out["xi"][:, ::4]
    ref = golden_as_ref(g, xi=out["xi"])
    test = dict(out)
    test["status"] = np.where(out["status"] == 0, 0, -1)
    noise = oracle_selfnoise(oracle, arrs, g["y0"], 0.0, 80e3, 400)
    for n in noise:
        n["status"] = np.where(n["status"] == 0, 0, -1)
        for nm in "Tzp":
            n[nm] = n[nm][:, ::4]
        n["xi"] = n["xi"][:, ::4][:, :-1]
    # the golden's last column is the sub-sampled grid's last point (index 396), not the end state
    good_cols = slice(0, -1)
    for d in (test, ref, *noise):
        for nm in "Tzp":
            d[nm] = d[nm][:, good_cols]
    ref["xi"] = ref["xi"][:, good_cols]
    assert_fan_parity(test, ref, noise_runs=noise, scales=(5000.0, 55.0, 1 / 1500.0), label="g5 lin")
    g = load("g5_flatearth.npz")
    check_against_golden(g, env_from(g), 0.0, 100e3, 101, label="g5 flat earth")


# ----------------------------------------------------------------------------- round 4: the wide pins at the headline range
def end_state_check(g, out, label):
    """End states against the REFERENCE's own +-1-ulp self-noise recorded per ray in the golden file (selfnoise_end):
    within max(REL_TOL x scale, NOISE_FACTOR x self-noise).  Returns the worst deviation / self-noise ratio among the rays
    that need the noise rule, and the count of such rays."""
    ok = g["ok"].astype(bool)
    end = np.stack([out["T"][:, -1], out["z"][:, -1], out["p"][:, -1]], 1)[ok]
    gend = np.stack([g["T"][:, -1], g["z"][:, -1], g["p"][:, -1]], 1)[ok]
    scale = np.array([float(np.nanmax(g["T"][ok])), float(g["zin"][-1]) if "zin" in g.files else 6000.0, 1 / 1500.0])
    d = np.abs(end - gend)
    sn = g["selfnoise_end"][ok]
    needs = d > REL_TOL * scale
    assert np.all(d <= np.maximum(NOISE_FACTOR * sn, REL_TOL * scale)), \
        f"{label}: end states beyond {NOISE_FACTOR} x the reference's self-noise: rays {np.where((d > np.maximum(NOISE_FACTOR * sn, REL_TOL * scale)).any(1))[0][:8]}"
    with np.errstate(divide="ignore", invalid="ignore"):
        ratio = float(np.nanmax(np.where(needs, d / sn, 0.0))) if needs.any() else 0.0
    return dict(end_noise_ratio=ratio, end_rays_on_noise_rule=int(needs.any(1).sum()), end_worst_rel=[float(v) for v in (d / scale).max(0)])


def test_munk_1000km_288_reference_rays():
    """g11: BASELINE configs[1] tables, 288 REFERENCE rays to 1000 km (104 bouncing, up to 64 bounces)."""
    g = load("g11_munk_1000km_288.npz")
    assert len(g["theta_ode"]) == 288 and ((g["n_bott"] + g["n_surf"]) > 0).sum() >= 100 and g["ok"].all()
    out, worst = check_against_golden(g, tiled_env(g), 0.0, 1000e3, 101, label="g11")
    worst.update(end_state_check(g, out, "g11"))
    print("\ng11 oracle vs reference:", worst)
    # (8 of the 184 rays that never touch a boundary differ by 1e-4 ... 3e-3 m at 1000 km -- as do the reference's own
    # +-1 ... 3 ulp neighbours of exactly those rays, selfnoise_end: the rule above prices them; the other 176 meet 1e-8)
    quiet = (g["n_bott"] + g["n_surf"]) == 0
    dq = np.abs(out["z"][quiet] - g["z"][quiet]).max(1) / 5000
    assert (dq < 1e-8).sum() >= 170 and np.all((dq < 1e-8) | (g["selfnoise_end"][quiet, 1] > 1e-5))


def test_config2_128_reference_rays():
    """g12: BASELINE configs[2] (range-dependent tables), 128 REFERENCE rays to 1000 km (one of them dropped by the reference)."""
    g = load("g12_config2_128.npz")
    arrs = munk_arrays(float(g["r_max"]), nr=int(g["nr"]), sofar_slope=float(g["sofar_slope"]))
    assert np.sum(arrs[0]) == pytest.approx(float(g["c_checksum"]), rel=1e-15) and np.sum(arrs[1]) == pytest.approx(float(g["cp_checksum"]), rel=1e-12)
    assert len(g["theta_ode"]) == 128 and ((g["n_bott"] + g["n_surf"]) > 0).sum() >= 39 and (g["ok"] == 0).sum() == 1
    out, worst = check_against_golden(g, arrs, 0.0, 1000e3, 101, label="g12")
    worst.update(end_state_check(g, out, "g12"))
    print("\ng12 oracle vs reference:", worst)


def test_default_environment_reference_rays():
    """g13: the reference's DEFAULT environment (flat-earth transformed Munk tables, the 4500 -> 4900 m slope, bottom angle
    from the untransformed bathymetry): 64 REFERENCE rays at 100 km, 32 at 1000 km."""
    for tag, x1, n in (("100km", 100e3, 64), ("1000km", 1000e3, 32)):
        g = load(f"g13_default_env_{tag}.npz")
        assert len(g["theta_ode"]) == n and g["ok"].all()
        assert not np.allclose(np.diff(g["zin"]), 1.0, rtol=0, atol=1e-9) and g["depths"][0] != g["depths"][-1] and np.any(g["bottom_angles"] != 0)
        out, worst = check_against_golden(g, tiled_env(g), 0.0, x1, 101, label="g13 " + tag)
        worst.update(end_state_check(g, out, "g13 " + tag))
        print(f"\ng13 {tag} oracle vs reference:", worst)
    # ... and these ARE the tables the drop-in environment builds (host side; no GPU needed)
    import pygenray_amd as pr
    arrs = pr._unpack_envi(pr.OceanEnvironment2D(), flatearth=True)
    g = load("g13_default_env_100km.npz")
    for a, b in zip(arrs, tiled_env(g)):
        assert np.array_equal(a, b)


# ----------------------------------------------------------------------------- the SciPy port
def test_scipy_port_reproduces_reference_and_c_oracle():
    """Same solve_ivp call pattern as the reference: matches the golden to rounding on
    non-bouncing rays, and pins the C restatement of RK45 / brentq / dense output."""
    g = load("g3_munk_1000km.npz")
    arrs = tiled_env(g)
    pick = [0, 4, 8, 11]
    sp = scipy_port.shoot_fan(*arrs, g["y0"][pick], 0.0, 1000e3, 101)
    co = oracle.shoot_fan(*arrs, g["y0"][pick], 0.0, 1000e3, 101)
    assert np.array_equal(sp["n_steps"], g["n_steps"][pick])
    assert np.array_equal(sp["nfev"][1:], g["nfev"][pick][1:])
    assert abs(int(sp["nfev"][0]) - int(g["nfev"][pick][0])) <= 12  # 64-bounce ray: one flipped reject
    assert np.array_equal(sp["n_bott"], g["n_bott"][pick]) and np.array_equal(sp["n_surf"], g["n_surf"][pick])
    good = np.abs(co["xi"]) <= XI_MAX
    for nm, scale in (("T", 670.0), ("z", 5000.0)):
        d = np.where(good, np.abs(sp[nm] - g[nm][pick]), 0)
        assert d[1:].max() / scale < 1e-9          # non-bouncing: rounding only
        assert d[0].max() / scale < 1e-6           # 32+32 bounces: chaos-amplified rounding
        d = np.where(good, np.abs(sp[nm] - co[nm]), 0)
        assert d[1:].max() / scale < 1e-8


def test_dropped_ray_statuses():
    """Rays the reference drops (returns None): backward bounce, bbox exit, vertical."""
    from helpers import munk_arrays, y0_for
    arrs = munk_arrays(50e3, nr=20, z=np.linspace(0, 6000, 601))
    # bathymetry with a steep up-slope wall: a downward ray bounces backwards
    arrs[4] = np.where(arrs[2] > 20e3, 1000.0, 5000.0).astype(float)
    arrs[6] = np.degrees(np.arctan(np.gradient(arrs[4], arrs[5])))
    y0 = y0_for(oracle, arrs, 500.0, 0.0, [12.0, 0.5])
    out = oracle.shoot_fan(*arrs, y0, 0.0, 50e3, 26)
    sp = scipy_port.shoot_fan(*arrs, y0, 0.0, 50e3, 26)
    assert list(out["status"]) == list(sp["status"])
    assert out["status"][0] == 3 and np.all(np.isnan(out["z"][0]))
    # table shallower than the bottom: ray leaves the bounding box
    arrs = munk_arrays(50e3, nr=20, z=np.linspace(0, 3000, 301), bathy=5000.0)
    y0 = y0_for(oracle, arrs, 500.0, 0.0, [14.0])
    out = oracle.shoot_fan(*arrs, y0, 0.0, 50e3, 26)
    sp = scipy_port.shoot_fan(*arrs, y0, 0.0, 50e3, 26)
    assert out["status"][0] == 2 and sp["status"][0] == 2
