"""Parity of the HIP path (through the C ABI) with the CPU oracle and the golden vectors.

Everything here needs a real MI355X: run with ``pytest -m gpu``.  The oracle is only the
checker; the thing under test is ``libpgr_hip.so`` reached through ctypes (pygenray_amd._lib)
and the drop-in API on top of it.
"""
import numpy as np
import pytest

import oracle
from helpers import (load, env_from, tiled_env, munk, munk_arrays, y0_for, assert_fan_parity,
                     assert_bit_parity, oracle_selfnoise, random_case, XI_MAX, NOISE_FACTOR)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    from pygenray_amd import _lib
    if _lib.ARITH != "reference":
        pytest.skip("bit parity is claimed for the reference arithmetic only (PGR_ARITH=contracted: tests/test_contracted_arith.py)")
    _lib.load()
    assert _lib.device_count() >= 1
    return _lib


def gpu_vs_oracle(lib, arrs, y0, x0, x1, S, label, max_odd=None, **kw):
    """The parity test proper: the HIP fan against the oracle in its correctly-rounded-libm mode,
    BIT FOR BIT -- status, bounce counts, accepted and rejected step counts, end states and every
    saved sample in SciPy's evaluation order (Q5 extrapolated ones included); the default sample
    form (stage-major FMAs inside a step) within 1e-12 of them."""
    env = lib.EnvHandle(*arrs)
    o = oracle.shoot_fan(*arrs, y0, x0, x1, S, math=oracle.MATH_CR, **kw)
    g = env.shoot_fan(y0, x0, x1, S, exact_samples=True, **kw)
    stats = assert_bit_parity(g, o, label=label, max_odd=max_odd)
    g2 = env.shoot_fan(y0, x0, x1, S, **kw)
    assert_bit_parity(g2, o, label=label + " (default sample form)", samples=False, max_odd=max_odd)
    ok = o["status"] == 0
    # the last column is the exact final state and equals end_state
    assert np.array_equal(g["end"][ok, 1], g["z"][ok, -1])
    env.close()
    return g, o, stats


# ------------------------------------------------------------------ unit level: a1-a8 on the device
def test_device_unit_vectors(lib):
    g = load("g7_unit_vectors.npz")
    nb = len(g["depths"])
    env = lib.EnvHandle(g["cin"], g["cpin"], g["rin"], g["zin"], g["depths"], g["depth_ranges"], np.zeros(nb))
    out = env.eval_points(g["xs"], g["ys"])
    np.testing.assert_allclose(out[:, 0:3], g["derivs"], rtol=1e-14, atol=0)
    np.testing.assert_allclose(out[:, 3], g["bilinear"], rtol=4e-16, atol=0)
    nan = np.isnan(g["angle"][:, 0])
    assert np.array_equal(np.isnan(out[:, 4]), nan)
    np.testing.assert_allclose(out[~nan, 4], g["angle"][~nan, 0], rtol=1e-13, atol=1e-12)
    np.testing.assert_array_equal(out[:, 5:9], g["events"])
    np.testing.assert_allclose(out[:, 9], g["linear"], rtol=4e-16, atol=0)
    # extrapolation quirk Q4 on the device
    gr = np.array([0.0, 1, 2, 3])
    v = np.arange(16.0).reshape(4, 4)
    e2 = lib.EnvHandle(v, v, gr, gr, np.zeros(4), gr, np.zeros(4))
    o2 = e2.eval_points(np.array([4.0, 0.5]), np.array([[0, 0.0, 0], [0, 0.5, 0]]))
    assert o2[0, 3] == oracle.bilinear(4.0, 0.0, gr, gr, v) and o2[1, 3] == oracle.bilinear(0.5, 0.5, gr, gr, v)


# ------------------------------------------------------------------ golden fans (reference outputs)
def golden_check(lib, g, arrs, x0, x1, S, prefix="", label="", abs_floor=None, strict_bouncing=True, bit_parity=True,
                 noise_ulps=(1,), **kw):
    """HIP against vectors the REFERENCE itself produced (statistical policy (B) of helpers.py) and,
    on the same inputs, against the oracle bit for bit (A).  (`bit_parity=False`: rule (B) alone -- what
    tests/test_contracted_arith.py asks of the PGR_ARITH=contracted library, which makes no bit-parity claim.)"""
    env = lib.EnvHandle(*arrs)
    out = env.shoot_fan(g[prefix + "y0"], x0, x1, S, exact_samples=True, **kw)
    o = oracle.shoot_fan(*arrs, g[prefix + "y0"], x0, x1, S, math=oracle.MATH_CR, **kw)
    if bit_parity:
        assert_bit_parity(out, o, label=label + " vs oracle")
    ok = g[prefix + "ok"].astype(bool)
    ref = dict(T=g[prefix + "T"], z=g[prefix + "z"], p=g[prefix + "p"], n_bott=g[prefix + "n_bott"],
               n_surf=g[prefix + "n_surf"], status=np.where(ok, 0, -1), xi=o["xi"])
    test = dict(out)
    test["status"] = np.where(out["status"] == 0, 0, -1)
    noise = oracle_selfnoise(oracle, arrs, g[prefix + "y0"], x0, x1, S, ulps=noise_ulps, **kw)
    for n in noise:
        n["status"] = np.where(n["status"] == 0, 0, -1)
    worst = assert_fan_parity(test, ref, noise_runs=noise,
                              scales=(float(arrs[3][-1]), float(np.nanmax(ref["T"])), 1 / 1500.0), label=label,
                              abs_floor=abs_floor, strict_bouncing=strict_bouncing)
    print(f"\n{label} HIP vs reference:", {k: v for k, v in worst.items() if "median" not in k})
    env.close()
    return out


def test_golden_munk_100km(lib):
    g = load("g2_munk_100km.npz")
    out = golden_check(lib, g, tiled_env(g), 0.0, 100e3, 101, label="g2")
    quiet = (g["n_bott"] + g["n_surf"]) == 0
    assert np.abs(out["z"][quiet] - g["z"][quiet]).max() / 5000 < 1e-8
    assert np.abs(out["T"][quiet] - g["T"][quiet]).max() / 67.0 < 1e-8
    assert np.array_equal(out["n_steps"][quiet], g["n_steps"][quiet])


def test_golden_irregular_range_bathymetry_and_depth_grids(lib):
    """g9 (reference output): randomly spaced rin / bathymetry ranges, stretched zin, range-dependent c,
    at rtol 1e-9 and at 1e-5 (steps wider than several range cells)."""
    g = load("g9_irregular_grids.npz")
    arrs = env_from(g)
    golden_check(lib, g, arrs, 1e3, 69e3, 61, prefix="t9_", rtol=1e-9, label="g9 rtol 1e-9", strict_bouncing=False)
    out = golden_check(lib, g, arrs, 1e3, 69e3, 61, prefix="t5_", rtol=1e-5, label="g9 rtol 1e-5", strict_bouncing=False)
    assert np.array_equal(out["n_steps"], g["t5_n_steps"])
    assert np.array_equal(out["n_bott"], g["t5_n_bott"]) and np.array_equal(out["n_surf"], g["t5_n_surf"])


def test_golden_munk_1000km(lib):
    g = load("g3_munk_1000km.npz")
    out = golden_check(lib, g, tiled_env(g), 0.0, 1000e3, 101, label="g3")
    end = out["end"]
    gend = np.stack([g["T"][:, -1], g["z"][:, -1], g["p"][:, -1]], 1)
    tol = np.maximum(NOISE_FACTOR * g["selfnoise_end"], 1e-8 * np.array([670.0, 6000.0, 1 / 1500.0]))
    assert np.all(np.abs(end - gend) <= tol)


def test_golden_wide_pins_at_the_headline_range(lib):
    """Round 4's reference-produced vectors (tests/golden/make_golden.py g11-g13): 288 rays of configs[1], 128 of configs[2],
    64 + 32 of the reference's default (flat-earth, sloping bottom) environment -- HIP against the reference (rule B at
    NOISE_FACTOR = 10) and against the oracle bit for bit (rule A)."""
    from test_oracle_golden import end_state_check
    g = load("g11_munk_1000km_288.npz")
    out = golden_check(lib, g, tiled_env(g), 0.0, 1000e3, 101, label="g11 configs[1] x 288")
    print("g11 end states:", end_state_check(g, out, "g11"))
    g = load("g12_config2_128.npz")
    arrs = munk_arrays(float(g["r_max"]), nr=int(g["nr"]), sofar_slope=float(g["sofar_slope"]))
    out = golden_check(lib, g, arrs, 0.0, 1000e3, 101, label="g12 configs[2] x 128")
    print("g12 end states:", end_state_check(g, out, "g12"))
    for tag, x1 in (("100km", 100e3), ("1000km", 1000e3)):
        g = load(f"g13_default_env_{tag}.npz")
        e = lib.EnvHandle(*tiled_env(g))
        assert e.query(5) == 1 and e.lds_path     # the cubic-index kernel (ZM = 5)
        e.close()
        out = golden_check(lib, g, tiled_env(g), 0.0, x1, 101, label="g13 default environment " + tag)
        print(f"g13 {tag} end states:", end_state_check(g, out, "g13 " + tag))


def test_golden_range_dependent_and_mirrored(lib):
    g = load("g4_range_dependent.npz")
    arrs = env_from(g)
    floor = dict(T=1e-6, z=1e-2, p=1e-7)  # the reference's own tolerances on this coarse grid
    golden_check(lib, g, arrs, 10e3, 90e3, 81, prefix="fwd_", label="g4 fwd", abs_floor=floor, strict_bouncing=False)
    arrs_m = [np.ascontiguousarray(arrs[0][::-1]), np.ascontiguousarray(arrs[1][::-1]), -arrs[2][::-1],
              arrs[3], np.ascontiguousarray(arrs[4][::-1]), -arrs[5][::-1], -arrs[6][::-1]]
    golden_check(lib, g, arrs_m, -60e3, -10e3, 80, prefix="bwd_", label="g4 bwd", abs_floor=floor, strict_bouncing=False)


def test_golden_config2_subset(lib):
    g = load("g4_config2_subset.npz")
    arrs = munk_arrays(float(g["r_max"]), nr=int(g["nr"]), sofar_slope=float(g["sofar_slope"]))
    golden_check(lib, g, arrs, 0.0, 1000e3, 101, label="g4 config2")


def test_golden_const_c_steep_linear_flatearth(lib):
    g = load("g5_const_c.npz")
    golden_check(lib, g, env_from(g), 0.0, 30e3, 60, label="const c")
    g = load("g5_const_c_steep.npz")
    golden_check(lib, g, env_from(g), 0.0, 1.5e3, 31, rtol=float(g["rtol"]), label="steep")
    g = load("g5_flatearth.npz")
    golden_check(lib, g, env_from(g), 0.0, 100e3, 101, label="flat earth (non-uniform zin)", strict_bouncing=False)


def test_reference_fixture_through_dropin_api(lib):
    """The reference's TestMunkRegression (tests/test_physics.py:310-386) run through the
    drop-in API against the reference's committed fixture."""
    import pygenray_amd as pr
    z = np.linspace(0.0, 6000.0, 400)
    r = np.linspace(0.0, 50e3, 30)
    ssp = pr.DataArray(np.outer(np.ones(30), pr.munk_ssp(z)), dims=["range", "depth"],
                       coords={"range": r, "depth": z})
    bathy = pr.DataArray(np.full(30, 5000.0), dims=["range"], coords={"range": r})
    env = pr.OceanEnvironment2D(sound_speed=ssp, bathymetry=bathy, flat_earth_transform=False)
    rf = pr.shoot_rays(1300.0, 0.0, [-8.0, -4.0, 0.0, 4.0, 8.0], 50e3, 50, env, n_processes=1,
                       debug=False, flatearth=False)
    ref = load("ref_munk_regression.npz")
    np.testing.assert_allclose(rf.ts, ref["ts"], atol=5e-6)
    np.testing.assert_allclose(rf.zs, ref["zs"], atol=0.1)
    np.testing.assert_allclose(rf.ps, ref["ps"], atol=0.1)
    np.testing.assert_array_equal(rf.n_botts, ref["n_botts"])
    np.testing.assert_array_equal(rf.n_surfs, ref["n_surfs"])
    np.testing.assert_array_equal(rf.thetas, ref["thetas"])


# ------------------------------------------------------------------ HIP vs oracle on seeded inputs
def test_fan_vs_oracle_munk_lds_path(lib):
    arrs = munk_arrays(200e3)
    theta = np.linspace(-20, 20, 333)  # ragged: not a multiple of 64
    y0 = y0_for(oracle, arrs, 1000.0, 0.0, theta)
    g, o, stats = gpu_vs_oracle(lib, arrs, y0, 0.0, 200e3, 201, "munk 200 km")
    assert ((o["n_bott"] + o["n_surf"]) > 0).sum() > 50 and stats["n"] >= 330


def test_fan_vs_oracle_range_dependent_sloping_bottom(lib):
    z = np.arange(0, 5500, 2.0)
    r = np.linspace(0, 150e3, 61)
    cin = np.array([munk(z, 1300 + 2e-3 * ri) + 2.0 * np.sin(ri / 30e3) * np.exp(-z / 800.0) for ri in r])
    cpin = np.gradient(cin, z, axis=1, edge_order=1)
    br = np.linspace(0, 150e3, 31)
    depths = 4800 + 300 * np.sin(br / 20e3)
    ba = np.degrees(np.arctan(np.gradient(depths, br)))
    arrs = [cin, cpin, r, z, depths, br, ba]
    y0 = y0_for(oracle, arrs, 700.0, 5e3, np.linspace(-18, 18, 130))
    g, o, worst = gpu_vs_oracle(lib, arrs, y0, 5e3, 140e3, 136, "range dependent")
    assert (o["n_bott"] > 0).any() and (o["n_surf"] > 0).any()


def test_nonuniform_grids_generic_search_path(lib):
    rng = np.random.default_rng(2)
    z = 6000.0 * np.linspace(0, 1, 1501) ** 1.3          # smooth, stretched: no exact-uniform path
    r = 80e3 * np.linspace(0, 1, 27) ** 0.8
    cin = np.array([munk(z, 1300 + 1e-3 * ri) for ri in r])
    cpin = np.gradient(cin, z, axis=1, edge_order=1)
    br = np.sort(np.concatenate([[0.0, 80e3], rng.uniform(0, 80e3, 9)]))
    depths = 4700 + 100 * np.cos(br / 9e3)
    arrs = [cin, cpin, r, z, depths, br, np.degrees(np.arctan(np.gradient(depths, br)))]
    env = lib.EnvHandle(*arrs)
    assert not env.query(1) and not env.query(2)
    env.close()
    y0 = y0_for(oracle, arrs, 900.0, 0.0, np.linspace(-15, 15, 70))
    gpu_vs_oracle(lib, arrs, y0, 0.0, 80e3, 81, "non-uniform grids")


def test_power_of_two_depth_steps_other_than_one(lib):
    """zin = j dz with dz a power of two takes the index-by-scaling path; dz = 1 m (the reference's
    default grid) has its own kernel instance without the scalings.  dz = 2 m and 0.25 m here."""
    for dz in (2.0, 0.25):
        z = np.arange(0, 1500.0 + dz / 2, dz)
        arrs = munk_arrays(40e3, nr=9, z=z, bathy=1400.0)
        env = lib.EnvHandle(*arrs)
        assert env.query(0) and env.query(1) and env.query(3)
        env.close()
        y0 = y0_for(oracle, arrs, 700.0, 0.0, np.linspace(-12, 12, 66))
        gpu_vs_oracle(lib, arrs, y0, 0.0, 40e3, 33, f"dz = {dz} m")


def test_depth_grids_too_large_for_the_lds(lib):
    """A 0.5 m depth grid (12 001 nodes, 192 KB of {c, cp}) does not fit the 160 KB LDS: the table
    stays in HBM/L2 -- also for a non-uniform version of the grid, whose depth search (zin + bucket
    table, 123 KB) still runs from LDS.  Same results as the oracle either way."""
    z = np.arange(0, 6000.25, 0.5)
    arrs = munk_arrays(60e3, nr=12, z=z)
    env = lib.EnvHandle(*arrs)
    assert env.query(0) and not env.query(3)          # range independent, but no LDS table
    env.close()
    y0 = y0_for(oracle, arrs, 1000.0, 0.0, np.linspace(-18, 18, 70))
    gpu_vs_oracle(lib, arrs, y0, 0.0, 60e3, 31, "0.5 m uniform grid (HBM table)")
    zs = z * (1 + 1e-5 * z / 6000.0)                  # smoothly stretched: non-uniform
    arrs2 = munk_arrays(60e3, nr=12, z=zs)
    gpu_vs_oracle(lib, arrs2, y0_for(oracle, arrs2, 1000.0, 0.0, np.linspace(-18, 18, 70)), 0.0, 60e3, 31,
                  "0.5 m non-uniform grid (HBM table, LDS depth search)")


def test_steps_wider_than_range_cells(lib):
    """The range weights of a step's five stage abscissae are computed together from the cached
    cell, its right neighbour, or cell by cell: a 100 m range grid under loose-tolerance steps
    of several hundred metres (every step straddles cells, many span more than two), uniform
    and non-uniform rin, range independent and dependent."""
    r_u = np.linspace(0.0, 100e3, 1001)
    r_n = np.sort(np.concatenate([[0.0, 100e3], np.random.default_rng(5).uniform(0, 100e3, 700)]))
    z = np.arange(0, 6000, 1.0)
    for r, slope in ((r_u, 0.0), (r_u, 1e-3), (r_n, 1e-3)):
        cin = np.array([munk(z, 1300 + slope * ri) for ri in r])
        arrs = [cin, np.gradient(cin, z, axis=1, edge_order=1), r, z, np.full(len(r), 5000.0), r.copy(),
                np.zeros(len(r))]
        y0 = y0_for(oracle, arrs, 1000.0, 0.0, np.linspace(-18, 18, 130))
        for rtol in (1e-5, 1e-9):
            gpu_vs_oracle(lib, arrs, y0, 0.0, 100e3, 41, f"fine range grid rtol={rtol}", rtol=rtol)


def test_dropped_rays_and_statuses(lib):
    # backward bounce off a wall, bbox exit, near-vertical rays: same statuses as the oracle
    arrs = munk_arrays(50e3, nr=20, z=np.linspace(0, 6000, 601))
    arrs[4] = np.where(arrs[2] > 20e3, 1000.0, 5000.0).astype(float)
    arrs[6] = np.degrees(np.arctan(np.gradient(arrs[4], arrs[5])))
    y0 = y0_for(oracle, arrs, 500.0, 0.0, [12.0, 0.5, -3.0, 14.0])
    g, o, _ = gpu_vs_oracle(lib, arrs, y0, 0.0, 50e3, 26, "wall")
    assert g["status"][0] == 3 and np.all(np.isnan(g["z"][0])) and np.all(np.isnan(g["end"][0]))
    arrs = munk_arrays(50e3, nr=20, z=np.linspace(0, 3000, 301), bathy=5000.0)
    y0 = y0_for(oracle, arrs, 500.0, 0.0, [14.0, 2.0])
    g, o, _ = gpu_vs_oracle(lib, arrs, y0, 0.0, 50e3, 26, "bbox")
    assert g["status"][0] == 2 and g["status"][1] == 0
    # the fan's span [x0, x1] reaches beyond the tables (the kernel then keeps the range tests of the
    # bounding-box event and of the bottom pre-filter on every step): receiver behind the last table column
    # -> every ray leaves the box in range; bathymetry table shorter than the span -> extrapolated sea floor
    arrs = munk_arrays(50e3, nr=20, z=np.linspace(0, 6000, 601))
    y0 = y0_for(oracle, arrs, 500.0, 0.0, [14.0, 2.0, -9.0])
    g, o, _ = gpu_vs_oracle(lib, arrs, y0, 0.0, 60e3, 26, "receiver beyond the table")
    assert np.all(g["status"] == 2)
    arrs = munk_arrays(50e3, nr=20, z=np.linspace(0, 6000, 601))
    arrs[5] = np.linspace(5e3, 30e3, 20)                     # depth_ranges cover 5..30 km of the 0..50 km shot
    arrs[4] = np.linspace(4000.0, 4600.0, 20)
    arrs[6] = np.degrees(np.arctan(np.gradient(arrs[4], arrs[5])))
    y0 = y0_for(oracle, arrs, 500.0, 0.0, np.linspace(-16, 16, 40))
    g, o, _ = gpu_vs_oracle(lib, arrs, y0, 0.0, 50e3, 26, "bathymetry table shorter than the shot")
    assert (g["n_bott"] > 0).any()


def test_near_vertical_rays_do_not_crash(lib):
    """tests/test_physics.py:394-455 of the reference: steep launches must not blow up (Q8)."""
    import pygenray_amd as pr
    z = np.linspace(0.0, 5000.0, 200)
    r = np.linspace(0.0, 100e3, 20)
    ssp = pr.DataArray(np.full((20, 200), 1500.0), dims=["range", "depth"], coords={"range": r, "depth": z})
    bathy = pr.DataArray(np.full(20, 4500.0), dims=["range"], coords={"range": r})
    env = pr.OceanEnvironment2D(ssp, bathy, flat_earth_transform=False)
    for ang in (-85.0, -89.0):
        ray = pr.shoot_ray(200.0, 0.0, ang, 2e3, 50, env, rtol=1e-6, flatearth=False, debug=False)
        assert ray is None or np.all(np.isfinite(ray.z))
    # exactly vertical / 89.9 deg bounce ~forever: the step guard ends them (status 5), no hang
    handle = lib.EnvHandle(*pr._unpack_envi(env, flatearth=False))
    y0 = np.array([[0, 200.0, np.sin(np.radians(a)) / 1500.0] for a in (89.9, 90.0)])
    out = handle.shoot_fan(y0, 0.0, 10e3, 50, rtol=1e-6, max_steps=20000)
    assert np.all(np.isin(out["status"], [0, 1, 3, 4, 5]))


def test_layouts_and_end_state_only(lib):
    arrs = munk_arrays(100e3)
    y0 = y0_for(oracle, arrs, 1000.0, 0.0, np.linspace(-19, 19, 200))
    env = lib.EnvHandle(*arrs)
    a = env.shoot_fan(y0, 0.0, 100e3, 64)
    b = env.shoot_fan(y0, 0.0, 100e3, 64, sample_major=True)
    c = env.shoot_fan(y0, 0.0, 100e3, 64, save=False)
    for nm in "Tzp":
        assert np.array_equal(a[nm], b[nm].T)
    assert np.array_equal(a["end"], c["end"]) and np.array_equal(a["n_steps"], c["n_steps"])
    assert c["T"] is None
    # every workgroup shape gives identical results (rays are independent)
    for w in (1, 3, 8):
        env.set_option("waves_per_block", w)
        d = env.shoot_fan(y0, 0.0, 100e3, 64)
        assert np.array_equal(a["z"], d["z"]) and np.array_equal(a["status"], d["status"])
    env.set_option("waves_per_block", 0)
    # no sentinel survives: every sample of an OK ray is written
    assert np.all(np.isfinite(a["T"][a["status"] == 0]))
    # empty fan and argument errors
    e = env.shoot_fan(np.zeros((0, 3)), 0.0, 100e3, 8)
    assert e["status"].shape == (0,)
    with pytest.raises(lib.PgrError):
        env.shoot_fan(y0, 100e3, 0.0, 8)
    with pytest.raises(lib.PgrError):
        lib.EnvHandle(arrs[0], arrs[1], arrs[2][::-1].copy(), arrs[3], arrs[4], arrs[5], arrs[6])
    with pytest.raises(lib.PgrError):
        lib.EnvHandle(arrs[0], arrs[1], arrs[2], arrs[3], arrs[4][:3], arrs[5][:3], arrs[6][:3])


def test_tolerances_outside_the_usual_range(lib):
    """rtol from 1e-3 to 1e-12 and atol from 1e-12 to 1e-2 (SciPy's controller sees error norms from 1e-7 to
    1e3 there, steps of metres to kilometres): bit for bit, bouncing rays included."""
    arrs = munk_arrays(120e3, nr=13)
    y0 = y0_for(oracle, arrs, 800.0, 0.0, np.linspace(-19.5, 19.5, 48))
    # (the last one is below 100 EPS: solve_ivp raises it to 2.2e-14, SCIPY/common.py:44-51)
    for rtol, atol in ((1e-3, 1e-6), (1e-12, 1e-6), (1e-11, 1e-12), (1e-6, 1e-2), (3e-14, 1e-9), (1e-15, 1e-9)):
        gpu_vs_oracle(lib, arrs, y0, 0.0, 120e3, 25, f"rtol={rtol} atol={atol}", rtol=rtol, atol=atol)


def test_two_host_threads_two_environments(lib):
    """The boundary holds no process-wide state (SURVEY 8(b) "callable from one host thread per GPU"): two host
    threads, each with its own environment (different tables, different scheduling options, its own stream),
    shoot fans at the same time; every result equals the single-threaded one bit for bit.  A third thread
    shares the first thread's environment (the library serialises its launches)."""
    import threading
    arrs_a = munk_arrays(100e3)
    arrs_b = munk_arrays(100e3, nr=41, sofar_slope=2e-4)                     # range dependent: the HBM-table kernels
    y0_a = y0_for(oracle, arrs_a, 1000.0, 0.0, np.linspace(-19, 19, 700))
    y0_b = y0_for(oracle, arrs_b, 600.0, 0.0, np.linspace(-17, 17, 900))
    env_a, env_b = lib.EnvHandle(*arrs_a), lib.EnvHandle(*arrs_b)
    env_b.set_option("park", 32, 8)
    env_b.set_option("placement", 1)
    ref_a = env_a.shoot_fan(y0_a, 0.0, 100e3, 33)
    ref_b = env_b.shoot_fan(y0_b, 0.0, 100e3, 17, sample_major=True)
    errors = []

    def work(env, y0, S, kw, ref, n):
        try:
            for _ in range(n):
                g = env.shoot_fan(y0, 0.0, 100e3, S, **kw)
                for k in ("T", "z", "p", "end", "status", "n_bott", "n_surf", "n_steps", "n_rej"):
                    if not np.array_equal(g[k], ref[k], equal_nan=(g[k].dtype.kind == "f")):
                        errors.append(f"{k} differs")
        except Exception as e:   # noqa: BLE001 -- reported by the main thread
            errors.append(repr(e))

    ts = [threading.Thread(target=work, args=(env_a, y0_a, 33, {}, ref_a, 6)),
          threading.Thread(target=work, args=(env_b, y0_b, 17, dict(sample_major=True), ref_b, 6)),
          threading.Thread(target=work, args=(env_a, y0_a, 33, {}, ref_a, 4))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors[:5]
    assert_bit_parity(ref_a, oracle.shoot_fan(*arrs_a, y0_a, 0.0, 100e3, 33, math=oracle.MATH_CR), "thread test, env a", samples=False)


def test_bucketed_depth_search_equals_binary_search(lib):
    """Non-uniform zin (the reference's default flat-earth grid, and a strongly graded one): the
    bucket-table cell search in LDS returns the cells of np.searchsorted -- every output bit equals
    the binary-search build of the same kernel, for the LDS-table and the HBM-table variant."""
    import pygenray_amd as pr
    from pygenray_amd.environment import _unpack_envi
    zz = np.arange(0, 6000, 1.0)
    rr = np.linspace(0, 200e3, 21)
    cases = []
    for slope in (0.0, 2e-4):   # range independent / dependent
        c2 = np.array([pr.munk_ssp(zz, 1300 + slope * ri) for ri in rr])
        env = pr.OceanEnvironment2D(pr.DataArray(c2, dims=["range", "depth"], coords={"range": rr, "depth": zz}),
                                    pr.DataArray(np.full(21, 5000.0), dims=["range"], coords={"range": rr}),
                                    flat_earth_transform=True)
        cases.append(_unpack_envi(env, flatearth=True))
    zg = np.concatenate([np.linspace(0, 300, 301), np.linspace(300, 5500, 261)[1:]])   # 1 m and 20 m cells
    cg = np.tile(munk(zg), (21, 1))
    cases.append((cg, np.gradient(cg, zg, axis=1), rr, zg, np.full(21, 5200.0), rr.copy(), np.zeros(21)))
    for arrs in cases:
        assert not np.allclose(np.diff(arrs[3]), np.diff(arrs[3])[0], rtol=1e-12, atol=0)  # really non-uniform
        y0 = y0_for(oracle, arrs, 400.0, 0.0, np.linspace(-19, 19, 500))
        env = lib.EnvHandle(*arrs)
        a = env.shoot_fan(y0, 0.0, 200e3, 41)           # automatic: index polynomial or bucket table
        for mode in (1, 2):                              # binary search; bucket table
            try:
                env.set_option("depth_search", mode)
                b = env.shoot_fan(y0, 0.0, 200e3, 41)
            finally:
                env.set_option("depth_search", 0)
            for k in ("T", "z", "p", "end"):
                assert np.array_equal(a[k], b[k], equal_nan=True), (mode, k)
            for k in ("status", "n_steps", "n_rej", "n_bott", "n_surf"):
                assert np.array_equal(a[k], b[k]), (mode, k)
        # and a depth below / above the grid still clamps to the edge cells (Q4)
        pts = env.eval_points(np.array([10e3, 10e3]), np.array([[0, arrs[3][0] - 5.0, 0], [0, arrs[3][-1] + 5.0, 0]]))
        assert np.all(np.isfinite(pts[:, 3]))
        env.close()


def test_save_grid_loaded_or_recomputed(lib):
    """The save grid is np.linspace: the kernel variant that recomputes r_save[j] per index
    (PGR_SAVE_LINSPACE) and the one that loads it give the same trajectories bit for bit, and
    the end-state-only variant the same end states."""
    import torch
    from pygenray_amd.device_fan import DeviceFan
    arrs = munk_arrays(150e3)
    env = lib.EnvHandle(*arrs)
    y0 = y0_for(oracle, arrs, 1000.0, 0.0, np.linspace(-20, 20, 700))
    outs = []
    for clear in (0, lib.PGR_SAVE_LINSPACE):
        fan = DeviceFan(env, y0, 0.0, 150e3, 151, save=True, sample_major=True)
        fan.flags &= ~clear
        fan.run(); torch.cuda.synchronize()
        outs.append([t.cpu().numpy() for t in (fan.T, fan.Z, fan.P, fan.end, fan.n_steps, fan.status)])
    for a, b in zip(*outs):
        assert np.array_equal(a, b, equal_nan=True)
    fan = DeviceFan(env, y0, 0.0, 150e3, 151, save=False)
    fan.run(); torch.cuda.synchronize()
    assert np.array_equal(fan.end.cpu().numpy(), outs[0][3], equal_nan=True)
    assert np.array_equal(fan.n_steps.cpu().numpy(), outs[0][4])


def test_compact_and_stored_sign_flags(lib):
    """PGR_COMPACT squeezes dropped rays out of the [S][N] trajectories on the device and
    PGR_STORED_SIGN stores -z, -p: both are exactly the host-side post-processing they replace."""
    arrs = munk_arrays(100e3)
    th = np.concatenate([np.linspace(-19, 19, 150), [89.9995, -89.9995, 89.9999], np.linspace(-5, 5, 20)])
    y0 = y0_for(oracle, arrs, 1000.0, 0.0, th)
    env = lib.EnvHandle(*arrs)
    a = env.shoot_fan(y0, 0.0, 100e3, 33, sample_major=True)
    b = env.shoot_fan(y0, 0.0, 100e3, 33, sample_major=True, compact=True, stored_sign=True)
    keep = a["status"] == 0
    assert 0 < keep.sum() < len(th)                     # some rays really are dropped
    assert np.array_equal(a["status"], b["status"]) and np.array_equal(a["end"], b["end"], equal_nan=True)
    assert b["T"].shape == (33, int(keep.sum()))
    assert np.array_equal(b["T"], a["T"][:, keep])
    assert np.array_equal(b["z"], -a["z"][:, keep]) and np.array_equal(b["p"], -a["p"][:, keep])
    # nothing dropped -> same layout as without the flag
    c = env.shoot_fan(y0[:150], 0.0, 100e3, 33, sample_major=True, compact=True)
    assert c["T"].shape == (33, 150) and np.array_equal(c["z"], a["z"][:, :150])


def test_reused_output_buffers_keep_what_the_call_does_not_write(lib):
    """The host entry faults the caller's output pages in while the kernel runs (pgr_shoot_fan); with
    PGR_COMPACT and dropped rays only [S][M] of the caller's [S][N] buffers is written, and the tail
    of a REUSED buffer must come back untouched (it used to lose the first byte of every page)."""
    arrs = munk_arrays(100e3)
    th = np.concatenate([np.linspace(-19, 19, 69_000), np.full(1000, 89.9999)])   # 1000 rays are dropped
    y0 = y0_for(oracle, arrs, 1000.0, 0.0, th)
    env = lib.EnvHandle(*arrs)
    N, S = len(th), 64                                   # 35.8 MB per array: above the prefault threshold
    bufs = [np.full((S, N), 7.25) for _ in range(3)]
    out = env.shoot_fan(y0, 0.0, 100e3, S, sample_major=True, compact=True, buffers=bufs)
    M = int((out["status"] == 0).sum())
    assert 0 < M < N and out["T"].shape == (S, M)
    for b in bufs:
        assert np.all(b.reshape(-1)[S * M:] == 7.25)
    ref = env.shoot_fan(y0, 0.0, 100e3, S, sample_major=True, compact=True)
    assert np.array_equal(out["z"], ref["z"])
    env.close()


def test_sample_evaluation_orders_agree(lib):
    """Default (stage-major FMA) and PGR_EXACT_SAMPLES (SciPy's Q = K.T @ P order) evaluate
    the SAME quartics of the SAME integration: steps, end states and the exact last column are
    bit-identical, interior samples differ by rounding only (they never feed back)."""
    arrs = munk_arrays(300e3)
    env = lib.EnvHandle(*arrs)
    S = 301
    for lo, hi in ((-10, 10), (-20, 20)):   # without / with bounces
        y0 = y0_for(oracle, arrs, 1000.0, 0.0, np.linspace(lo, hi, 333))
        a = env.shoot_fan(y0, 0.0, 300e3, S)
        b = env.shoot_fan(y0, 0.0, 300e3, S, exact_samples=True)
        assert np.array_equal(a["end"], b["end"], equal_nan=True)
        assert np.array_equal(a["n_steps"], b["n_steps"]) and np.array_equal(a["status"], b["status"])
        ok = a["status"] == 0
        for nm, scale in (("T", 300e3 / 1500.0), ("z", 5000.0), ("p", 1 / 1500.0)):
            assert np.array_equal(a[nm][ok, -1], b[nm][ok, -1])
            d = np.abs(a[nm][ok] - b[nm][ok]) / scale
            assert np.median(d) < 1e-15, (nm, np.median(d))
            # the stage-major form is used inside a step only (0 <= xi <= 1); the extrapolated
            # end-of-segment samples (Q5) keep SciPy's order and are bit-identical
            assert d.max() < 1e-12, (nm, d.max())


# ------------------------------------------------------------------ full size: size-independent properties
def test_full_size_config1_properties(lib):
    """BASELINE configs[1] at full size (1e5 rays, 1000 km): the oracle cannot run this in
    seconds, so check properties: (i) every 10th ray (10 000 rays, ~5 s of oracle on the GPU box's 16 cores; ALL
    1e5 rays: scripts/bitparity.py - 1 1, profiles/r03_bitparity_all_rays.txt) equals the oracle BIT FOR BIT
    (helpers.py, rule (A)), (ii) determinism, (iii) the range-independent Hamiltonian
    sqrt(1/c^2 - p^2) is conserved along every ray, (iv) up/down symmetry of step counts is
    not required but bounce counts are monotone in |angle| at the fan edges."""
    arrs = munk_arrays(1000e3)
    theta = np.linspace(-20, 20, 100_000)
    y0 = y0_for(oracle, arrs, 1000.0, 0.0, -theta)
    env = lib.EnvHandle(*arrs)
    a = env.shoot_fan(y0, 0.0, 1000e3, 101)
    env.set_option("placement", 0)  # cost-aware wave scheduling off: same rays, same bits, different SIMDs
    b = env.shoot_fan(y0, 0.0, 1000e3, 101)
    env.set_option("placement", 1)
    b1 = env.shoot_fan(y0[:80000], 0.0, 1000e3, 3)
    env.set_option("placement", 2)
    b2 = env.shoot_fan(y0[:80000], 0.0, 1000e3, 3)
    assert np.array_equal(b1["end"], b2["end"], equal_nan=True) and np.array_equal(b1["end"], a["end"][:80000], equal_nan=True)
    assert np.array_equal(a["end"], b["end"], equal_nan=True) and np.array_equal(a["n_steps"], b["n_steps"])
    assert np.array_equal(a["z"], b["z"], equal_nan=True) and np.array_equal(a["status"], b["status"])
    ok = a["status"] == 0
    assert ok.mean() > 0.999
    sub = np.arange(0, 100_000, 10)
    o = oracle.shoot_fan(*arrs, y0[sub], 0.0, 1000e3, 101, math=oracle.MATH_CR)
    gsub = {k: (v[sub] if isinstance(v, np.ndarray) and v.shape[:1] == (100_000,) else v) for k, v in a.items()}
    st = assert_bit_parity(gsub, o, label="config1, every 10th ray", samples=False)
    assert st["n"] > 9900 and (o["n_bott"] + o["n_surf"] > 0).sum() > 2500
    # Hamiltonian at the end state vs at the source
    zc = np.clip(a["end"][ok, 1], 0, 5998.999)
    c_end = np.interp(zc, arrs[3], arrs[0][0])
    H_end = np.sqrt(np.maximum(1 / c_end**2 - a["end"][ok, 2]**2, 0))
    c0 = oracle.bilinear(0.0, 1000.0, arrs[2], arrs[3], arrs[0])
    H0 = np.sqrt(1 / c0**2 - y0[ok, 2]**2)
    # (the dc/dz table is a finite difference, not the slope of the piecewise-linear c, so H
    # drifts at the 1e-5 level over 1000 km; the reference's own bound is 1e-3 at 100 km)
    assert np.abs(H_end / H0 - 1).max() < 1e-4
    total = int(a["n_steps"][ok].sum())
    assert 1.2e8 < total < 1.7e8


def _subset(out, sub, n):
    return {k: (v[sub] if isinstance(v, np.ndarray) and v.shape[:1] == (n,) else v) for k, v in out.items()}


def test_full_size_config2_properties(lib):
    """BASELINE configs[2] at full size (range-dependent c(r, z): sofar axis + 2e-4 r over 101 columns, 1e5 rays,
    1000 km; the tables stay in HBM / L2): every 10th ray against the oracle BIT FOR BIT -- with trajectories in
    SciPy's sample order, with the default sample form, and end state only (three different kernel instances) --
    and the three runs agree with each other on every one of the 1e5 rays."""
    arrs = munk_arrays(1000e3, nr=101, sofar_slope=2e-4)
    n = 100_000
    y0 = y0_for(oracle, arrs, 1000.0, 0.0, -np.linspace(-20, 20, n))
    env = lib.EnvHandle(*arrs)
    assert not env.lds_path and not env.range_independent
    a = env.shoot_fan(y0, 0.0, 1000e3, 101, exact_samples=True)
    b = env.shoot_fan(y0, 0.0, 1000e3, 101)
    c = env.shoot_fan(y0, 0.0, 1000e3, 101, save=False)
    for k in ("end", "n_steps", "n_rej", "n_bott", "n_surf", "status"):
        assert np.array_equal(a[k], b[k], equal_nan=True) and np.array_equal(a[k], c[k], equal_nan=True), k
    sub = np.arange(0, n, 10)
    o = oracle.shoot_fan(*arrs, y0[sub], 0.0, 1000e3, 101, math=oracle.MATH_CR)
    st = assert_bit_parity(_subset(a, sub, n), o, label="config2, every 10th ray, SciPy sample order")
    assert_bit_parity(_subset(b, sub, n), o, label="config2, every 10th ray, default samples", samples=False)
    assert st["n"] > 9800 and (o["n_bott"] + o["n_surf"] > 0).sum() > 2500
    ok = a["status"] == 0
    assert 0.99 < ok.mean() < 1.0 and 1.2e8 < int(a["n_steps"][ok].sum()) < 1.7e8
    env.close()


def test_full_size_default_flat_earth_environment(lib):
    """The reference's DEFAULT path at size: OceanEnvironment2D() (Munk profile on arange(0, 6000, 1), 4500 -> 4900 m
    slope, flat_earth_transform=True: a smoothly NON-uniform zin -> the cubic-index depth look-up, kernel ZM = 5)
    and shoot_rays(..., flatearth=True) with 1e5 launch angles over its 100 km.  EVERY ray against the
    oracle (which finds the depth cell by binary search, as np.searchsorted does) BIT FOR BIT, end state and all
    samples; and the same fan through the three-node search of round 2 and through the binary search on the device:
    equal on every ray."""
    import pygenray_amd as pr
    env_obj = pr.OceanEnvironment2D()
    arrs = pr._unpack_envi(env_obj, flatearth=True)
    assert not np.allclose(np.diff(arrs[3]), 1.0, rtol=0, atol=1e-9)      # the transformed grid is not uniform
    n = 100_000
    theta = np.linspace(-20, 20, n)
    y0 = y0_for(oracle, arrs, 1000.0, 0.0, -theta)
    env = lib.EnvHandle(*arrs)
    assert env.query(5) == 1 and env.lds_path                            # cubic index estimate verified by the host
    a = env.shoot_fan(y0, 0.0, 100e3, 101, exact_samples=True)
    o = oracle.shoot_fan(*arrs, y0, 0.0, 100e3, 101, math=oracle.MATH_CR)     # (~6 s on the GPU box's 16 cores)
    st = assert_bit_parity(a, o, label="default flat-earth environment, every ray")
    assert st["n"] > 99_000 and (o["n_bott"] + o["n_surf"] > 0).sum() > 20_000
    for mode in (3, 2, 1):   # quadratic estimate + three nodes, bucket table, binary search
        env.set_option("depth_search", mode)
        b = env.shoot_fan(y0, 0.0, 100e3, 101, exact_samples=True)
        for k in ("T", "z", "p", "end", "n_steps", "n_rej", "n_bott", "n_surf", "status"):
            assert np.array_equal(a[k], b[k], equal_nan=True), (mode, k)
    env.close()
    # ... and through the drop-in API with the reference's default arguments (flatearth=True)
    fan = pr.shoot_rays(1000.0, 0.0, theta, 100e3, 101, env_obj, debug=False)
    keep = a["status"] == 0
    assert len(fan) == int(keep.sum())
    assert np.array_equal(fan.ts[:, -1], a["end"][keep, 0]) and np.array_equal(-fan.zs[:, -1], a["end"][keep, 1])
    assert np.array_equal(fan.n_botts, a["n_bott"][keep])


# ------------------------------------------------------------------ the benchmark AS IT IS BENCHMARKED: S = 1001 at 1000 km
def _as_benchmarked(lib, arrs, label, every=50, exact_too=True, blocked=False):
    """bench.py's pass over one workload -- DeviceFan(env, y0, 0, 1000 km, S = 1001, sample-major [S][N] outputs left in
    HBM, ODE signs, linspace save grid recomputed on the device, default wave placement), 1e5 launch angles -- and every
    `every`-th ray of it against the oracle (MATH_CR): status / bounces / accepted and rejected steps / end state bit-equal;
    the default sample form (stage-major FMAs inside a step) within 1e-12 x scale inside a step and BIT-EQUAL on the
    extrapolated (Q5) samples behind and beyond it and in the exact last column; with PGR_EXACT_SAMPLES every one of
    the 1001 samples of every compared ray bit-equal.  One sample per km = 0.7 samples per accepted step: most steps own
    one sample, many two, every bounce re-samples (REF/launch_rays.py:745-784, the idx1 == idx2 skip included)."""
    import torch
    from pygenray_amd.device_fan import DeviceFan, fan_y0
    n, S, x1 = 100_000, 1001, 1000e3
    theta = np.linspace(-20, 20, n)
    y0 = fan_y0(arrs, 1000.0, 0.0, -theta)
    assert np.array_equal(y0, y0_for(oracle, arrs, 1000.0, 0.0, -theta))
    sub = np.arange(0, n, every)
    o = oracle.shoot_fan(*arrs, y0[sub], 0.0, x1, S, math=oracle.MATH_CR)
    env = lib.EnvHandle(*arrs)
    tsub = torch.from_numpy(sub).cuda()
    stats = None
    for exact in ((False, True) if exact_too else (False,)):
        fan = DeviceFan(env, y0, 0.0, x1, S, save=True, sample_major=True, exact_samples=exact, sample_blocked=(blocked and not exact))
        fan.run()
        torch.cuda.synchronize()
        if fan.sample_blocked:     # [S/4][N][4] -> the compared rays' (n, S) trajectories
            pick = lambda t: t[:, tsub, :].permute(1, 0, 2).reshape(len(sub), -1)[:, :S].cpu().numpy()   # noqa: E731
        else:
            pick = lambda t: t[:, tsub].T.cpu().numpy()   # noqa: E731
        g = {"T": pick(fan.T), "z": pick(fan.Z), "p": pick(fan.P), "end": fan.end[tsub].cpu().numpy()}
        for k in ("n_bott", "n_surf", "status", "n_steps", "n_rej"):
            g[k] = getattr(fan, k)[tsub].cpu().numpy()
        st = assert_bit_parity(g, o, label=f"{label}, S = 1001, every {every}th ray" + (", SciPy sample order" if exact else ", default sample form"),
                               samples=exact)
        if not exact:
            stats = st
            # what the default form may differ by, measured: samples inside a step against SciPy's order
            ok = o["status"] == 0
            inside = (o["xi"] >= 0) & (o["xi"] <= 1)
            inside[:, -1] = False
            dz = np.abs(g["z"] - o["z"])[ok][inside[ok]].max() / 5000.0
            dt = np.abs(g["T"] - o["T"])[ok][inside[ok]].max() / np.nanmax(o["T"][ok])
            outside = ~inside & ok[:, None]
            assert np.array_equal(g["z"][outside], o["z"][outside]) and np.array_equal(g["T"][outside], o["T"][outside])
            stats.update(rel_dz_inside=float(dz), rel_dt_inside=float(dt), q5_samples=int((outside[:, :-1]).sum()),
                         samples_per_step=float(S * ok.sum() / o["n_steps"][ok].sum()),
                         bouncing=int(((o["n_bott"] + o["n_surf"]) > 0).sum()), total_steps=int(fan.ray_steps()))
            assert dz < 1e-12 and dt < 1e-12
        del fan
    env.close()
    print(f"\n{label} as benchmarked: {stats}")
    return stats


def test_config1_as_benchmarked_S1001(lib):
    """BASELINE configs[1] exactly as bench.py runs it (headline line): pgr_fan_kernel<true, 4, 1>."""
    import bench       # (bench.py's own table producer: its bottom angle is arctan(np.gradient(5000 m)) = 1e-15 degrees, not 0)
    st = _as_benchmarked(lib, bench.munk_tables(1000e3)[1], "configs[1]")
    assert st["n"] >= 1990 and st["bouncing"] > 500 and st["q5_samples"] > 1000 and 0.6 < st["samples_per_step"] < 0.8


def test_config2_as_benchmarked_S1001(lib):
    """BASELINE configs[2] exactly as bench.py's `range_dependent` leg runs it: `trajectories` = pgr_fan_kernel<false, 4, 3>
    (the sample-blocked layout, PGR_SAMPLE_BLOCKED) and `trajectories_row_layout` = pgr_fan_kernel<false, 4, 1>."""
    import bench
    arrs = bench.munk_tables(1000e3, nr=101, sofar_slope=2e-4)[1]
    st = _as_benchmarked(lib, arrs, "configs[2], sample-blocked layout", blocked=True)
    assert st["n"] >= 1980 and st["bouncing"] > 500 and st["q5_samples"] > 1000
    st = _as_benchmarked(lib, arrs, "configs[2], row layout", exact_too=False)
    assert st["n"] >= 1980


def test_sample_blocked_layout_holds_the_same_bits_as_the_row_layout(lib):
    """PGR_SAMPLE_BLOCKED (HBM-table kernels, SAVE = 3): [ceil(S/4)][N][4] un-blocked equals the [S][N] rows of the
    plain kernel bit for bit -- every S mod 4, dropped rays (NaN columns), bouncing rays (the re-sample after a bounce
    rewrites a sample of a block already stored), every depth-search instance, the full configs[2] fan -- and the flag
    is refused where it does not apply."""
    import torch
    from pygenray_amd.device_fan import DeviceFan, fan_y0
    import pygenray_amd as pr
    from pygenray_amd import _lib

    def both(arrs, y0, x1, S):
        env = lib.EnvHandle(*arrs)
        assert not env.lds_path
        out = []
        for blocked in (False, True):
            fan = DeviceFan(env, y0, 0.0, x1, S, save=True, sample_major=True, sample_blocked=blocked)
            if blocked:
                for t in (fan.T, fan.Z, fan.P):
                    t.fill_(-7.0)                  # (padding rows must not matter; rows < S must all be written)
            fan.run()
            torch.cuda.synchronize()
            out.append([fan.rows(t).cpu().numpy() for t in (fan.T, fan.Z, fan.P)] + [fan.end.cpu().numpy(), fan.status.cpu().numpy(),
                                                                                   fan.n_steps.cpu().numpy()])
            assert out[-1][0].shape == (S, len(y0))
        env.close()
        for a_, b_ in zip(*out):
            assert np.array_equal(a_, b_, equal_nan=True)
        return out[0]

    # range-dependent Munk, table shallower than the sea floor: deep rays leave the table (dropped -> NaN columns)
    zt = np.arange(0, 4200, 1.0)
    arrs = munk_arrays(300e3, nr=41, z=zt, bathy=5000.0, sofar_slope=1e-3)
    y0 = y0_for(oracle, arrs, 1000.0, 0.0, np.linspace(-20, 20, 700))
    for S in (1, 2, 3, 4, 5, 6, 7, 8, 301, 302, 303, 304):
        o = both(arrs, y0, 300e3, S)
    assert 20 < (o[4] != 0).sum() < 600 and np.isnan(o[0][:, o[4] != 0]).all() and not np.isnan(o[0][:, o[4] == 0]).any()
    # sloping bottom + every depth-search instance (uniform non-power-of-two grid, stretched grid, flat-earth grid)
    for z in (np.linspace(0, 5500, 1377), 5500 * np.linspace(0, 1, 1200) ** 1.3, pr.eflat(np.arange(0, 5600, 2.0), 35.0)[0]):
        r = np.linspace(0, 200e3, 33)
        cin = np.array([munk(z, 1300 + 5e-4 * ri) for ri in r])
        br = np.linspace(0, 200e3, 9)
        depths = 4800 + 300 * np.sin(br / 40e3)
        arrs2 = [cin, np.gradient(cin, z, axis=1, edge_order=1), r, z, depths, br, np.degrees(np.arctan(np.gradient(depths, br)))]
        both(arrs2, y0_for(oracle, arrs2, 800.0, 0.0, np.linspace(-18, 18, 500)), 200e3, 203)
    # configs[2] at full size
    import bench
    arrs3 = bench.munk_tables(1000e3, nr=101, sofar_slope=2e-4)[1]
    o = both(arrs3, fan_y0(arrs3, 1000.0, 0.0, -np.linspace(-20, 20, 100_000)), 1000e3, 1001)
    assert (o[4] != 0).sum() > 100 and int(o[5][o[4] == 0].sum()) > 1.2e8
    # round 6: the blocked layout is what DeviceFan picks BY ITSELF for an HBM-table environment (pgr_env_query(env, 8)) -- and
    # rows for an LDS-table one, for the SciPy sample order and for ray-major output; api_blocked = 0 turns the choice off
    env_h, env_l = lib.EnvHandle(*arrs), lib.EnvHandle(*munk_arrays(100e3))
    assert env_h.blocked_layout and not env_l.blocked_layout
    auto = DeviceFan(env_h, y0[:130], 0.0, 300e3, 10, save=True, sample_major=True)
    rows_ = DeviceFan(env_h, y0[:130], 0.0, 300e3, 10, save=True, sample_major=True, sample_blocked=False)
    assert auto.sample_blocked and tuple(auto.T.shape) == (3, 130, 4) and not rows_.sample_blocked
    auto.run(); rows_.run(); torch.cuda.synchronize()
    for k in ("T", "Z", "P"):
        assert np.array_equal(auto.rows(getattr(auto, k)).cpu().numpy(), getattr(rows_, k).cpu().numpy(), equal_nan=True)
    assert env_h.last_instance()["save"] == 1      # (the row-layout fan ran last)
    assert not DeviceFan(env_l, y0[:64], 0.0, 100e3, 10, save=True, sample_major=True).sample_blocked
    assert not DeviceFan(env_h, y0[:64], 0.0, 300e3, 10, save=True, sample_major=True, exact_samples=True).sample_blocked
    assert not DeviceFan(env_h, y0[:64], 0.0, 300e3, 10, save=True, sample_major=False).sample_blocked
    assert not DeviceFan(env_h, y0[:64], 0.0, 300e3, 10, save=False, sample_major=True).sample_blocked
    env_h.set_option("api_blocked", 0)
    assert not env_h.blocked_layout and not DeviceFan(env_h, y0[:64], 0.0, 300e3, 10, save=True, sample_major=True).sample_blocked
    env_h.close(); env_l.close()
    # refused: an LDS-table environment; the SciPy sample order; ray-major output
    env1 = lib.EnvHandle(*munk_arrays(100e3))
    y1 = y0_for(oracle, munk_arrays(100e3), 1000.0, 0.0, np.linspace(-5, 5, 64))
    with pytest.raises(_lib.PgrError):
        DeviceFan(env1, y1, 0.0, 100e3, 11, save=True, sample_major=True, sample_blocked=True).run()
    env2 = lib.EnvHandle(*arrs)
    with pytest.raises(_lib.PgrError):
        DeviceFan(env2, y0[:64], 0.0, 300e3, 11, save=True, sample_major=True, sample_blocked=True, exact_samples=True).run()
    with pytest.raises(ValueError):
        DeviceFan(env2, y0[:64], 0.0, 300e3, 11, save=True, sample_major=False, sample_blocked=True)
    # ... and in the host-pointer entry, whose buffers are [S][N]
    import ctypes
    L = _lib.load()
    S_, n_ = 11, 64
    r_ = np.linspace(0.0, 300e3, S_)
    bufs = [np.empty((S_, n_)) for _ in range(3)]
    ints = [np.zeros(n_, np.int32) for _ in range(5)]
    end_ = np.empty((n_, 3))
    vp = lambda a_: a_.ctypes.data_as(ctypes.c_void_p)   # noqa: E731
    rc = L.pgr_shoot_fan(env2._h, vp(np.ascontiguousarray(y0[:n_])), n_, 0.0, 300e3, vp(r_), S_, 1e-9, 1e-6,
                         _lib.PGR_SAMPLE_MAJOR | _lib.PGR_SAMPLE_BLOCKED, 10**6, vp(bufs[0]), vp(bufs[1]), vp(bufs[2]), vp(end_),
                         vp(ints[0]), vp(ints[1]), vp(ints[2]), vp(ints[3]), vp(ints[4]))
    assert rc != 0 and b"PGR_SAMPLE_BLOCKED" in L.pgr_last_error()


def test_flat_earth_leg_as_benchmarked_S1001(lib):
    """bench.py's `flatearth_default` leg: the configs[1] tables after OceanEnvironment2D's default flat-earth transform
    (smoothly non-uniform zin; kernel pgr_fan_kernel<true, 5, 1>, cubic index estimate) at 1000 km."""
    import bench
    import pygenray_amd as pr
    env_obj, _ = bench.munk_tables(1000e3)
    env_obj.flat_earth_transform(lat=35)
    arrs = pr._unpack_envi(env_obj, flatearth=True)
    assert not np.allclose(np.diff(arrs[3]), 1.0, rtol=0, atol=1e-9)
    e = lib.EnvHandle(*arrs)
    assert e.query(5) == 1 and e.lds_path
    e.close()
    st = _as_benchmarked(lib, arrs, "flat-earth leg")
    assert st["n"] >= 1980 and st["bouncing"] > 500


def test_cubic_index_estimate_qualifies_only_smooth_grids(lib):
    """pgr_env_create verifies the cubic index estimate node by node (fit_cubic_index): the flat-earth map of a uniform
    grid and a grid of accumulated steps (linear up to rounding, not bitwise uniform) qualify; a power-law stretch, a grid with a kink and a random
    grid do not (they keep the three-node / bin-table / binary search) -- and all of them integrate to the oracle's bits."""
    import pygenray_amd as pr
    rng = np.random.default_rng(7)
    zu = np.arange(0, 5600, 2.0)
    grids = {
        "flat-earth of arange": (pr.eflat(zu, 35.0)[0], 1),
        "flat-earth at 80 N": (pr.eflat(zu, 80.0)[0], 1),
        "accumulated steps: linear to rounding, not bitwise j * dz": (np.cumsum(np.r_[0.0, np.full(2800, 2.0001)]), 1),
        "power-law stretch": (5600 * np.linspace(0, 1, 1500) ** 1.5, 0),
        "kinked": (np.concatenate([np.linspace(0, 1000, 500), np.linspace(1000, 5600, 800)[1:]]), 0),
        "random": (np.sort(np.concatenate([[0, 5600], rng.uniform(0, 5600, 1200)])), 0),
        "uniform (closed form, no estimate needed)": (zu, 0),
    }
    r = np.linspace(0, 80e3, 12)
    th = np.linspace(-16, 16, 96)
    for name, (z, want) in grids.items():
        cin = np.tile(munk(z), (len(r), 1))
        arrs = [cin, np.gradient(cin, z, axis=1, edge_order=1), r, z, np.full(12, 5000.0), r.copy(), np.zeros(12)]
        env = lib.EnvHandle(*arrs)
        assert env.query(5) == want, name
        env.close()
        y0 = y0_for(oracle, arrs, 900.0, 0.0, th)
        gpu_vs_oracle(lib, arrs, y0, 0.0, 80e3, 17, label=name)


def test_arithmetic_building_blocks(lib):
    """The kernel's divide / sqrt expansions must be correctly rounded on the operand ranges that
    occur (they replace the compiler's IEEE expansions), 10*ulp(t) exact, and the three libm
    functions of the reference -- err ** -0.2, (0.01/d) ** 0.2, arcsin, sin -- CORRECTLY ROUNDED
    (csrc/pgr_crmath.h): equal to the oracle's binary128 evaluation rounded once."""
    rng = np.random.default_rng(0)
    M = 1_000_000
    a = rng.uniform(-1e4, 1e4, M) * 10.0 ** rng.integers(-8, 8, M)
    b = rng.uniform(0.1, 10, M) * 10.0 ** rng.integers(-12, 12, M)
    b[:300000] = 10 ** rng.uniform(np.log10(5e-6), np.log10(1800), 300000)
    o = lib.debug_math(a, b)
    assert np.array_equal(o[:, 0], a / b)
    assert np.array_equal(o[:, 1], 1 / b)
    assert np.array_equal(o[:, 2], 1 / np.sqrt(b))
    assert np.array_equal(o[:, 3], np.sqrt(b))
    assert np.array_equal(o[:, 5], 10 * np.abs(np.nextafter(a, np.inf) - a))
    # the correctly rounded functions, on the ranges the integrator feeds them
    x = np.exp(rng.uniform(np.log(1e-7), np.log(1e4), M))                       # error norms
    v = np.concatenate([rng.uniform(-1, 1, M // 2), rng.uniform(-1, 1, M // 4) ** 5,
                        np.sign(rng.uniform(-1, 1, M - M // 2 - M // 4)) * (1 - 10 ** rng.uniform(-16, -0.3, M - M // 2 - M // 4))])
    o = lib.debug_math(v, x)
    assert np.array_equal(o[:, 4], oracle.math_fn("pow_m02", x))
    assert np.array_equal(o[:, 6], oracle.math_fn("pow_p02", x))
    assert np.array_equal(o[:, 7], oracle.math_fn("asin", v))
    assert np.array_equal(o[:, 8], oracle.math_fn("sin", v))
    w = rng.uniform(-6.5, 6.5, M)                                               # radians(theta_b), |theta_b| <= 270 deg
    assert np.array_equal(lib.debug_math(w, x)[:, 8], oracle.math_fn("sin", w))
    # (the platform libm is NOT this: glibc differs from the correctly rounded value in ~0.1 % of calls)
    assert 0 < np.mean(oracle.math_fn("pow_m02", x, math=oracle.MATH_LIBM) != o[:, 4]) < 0.01
    # 1/sqrt next to the powers of two, where a divisor with an all-ones significand (s = RN(sqrt x) = 1 - 2^-53)
    # is the classic exception of Newton-Raphson reciprocals: correctly rounded there too (pgr_device.h: frsqrt)
    k = np.arange(0, 100_000, dtype=np.float64)
    for base in (1.0, 0.25, 4.0):
        for near in (base * (1.0 - k * 2.0 ** -53), base * (1.0 + k * 2.0 ** -52)):
            got = lib.debug_math(np.ones_like(near), near)[:, 2]
            wrong = np.where(got != 1 / np.sqrt(near))[0]
            assert len(wrong) == 0, (base, wrong[:8])
    # ... and the quotients by such divisors (all-ones significand and its neighbours), random numerators
    for e in (-40, -1, 0, 1, 30):
        bb = np.repeat(2.0 ** e * (2.0 - np.arange(1, 9) * 2.0 ** -52), 25_000)
        aa = rng.uniform(0.5, 2.0, len(bb)) * 10.0 ** rng.integers(-8, 8, len(bb))
        o = lib.debug_math(aa, bb)
        assert np.array_equal(o[:, 0], aa / bb), (e, np.where(o[:, 0] != aa / bb)[0][:8])
        # (column 1, the bare Newton reciprocal, only ever SEEDS such quotients in the kernel: not required here)


def test_powers_hard_cases_on_the_device(lib):
    """The Ziv evaluation of err ** -0.2 and (0.01 / d) ** 0.2 ON THE DEVICE (csrc/pgr_crmath.h): every argument the
    rounding test flagged in 6e9 / 1.5e9 random draws (tests/golden/g14_pow_hard_cases.npz; expected values from mpmath at
    400 bits), among them >= 24 whose exact power lies within 2^-80 of a rounding boundary -- bit-equal; padded to whole
    waves with easy arguments, and alone in a wave (the second level runs behind a wave-uniform ballot)."""
    g = load("g14_pow_hard_cases.npz")
    rng = np.random.default_rng(14)
    for tag, col in (("m02", 4), ("p02", 6)):
        xs, want, dist = g[tag + "_x"], g[tag + "_want"], g[tag + "_log2_dist"]
        assert (dist < -80).sum() >= (24 if tag == "m02" else 6)
        out = lib.debug_math(np.ones(len(xs)), xs)[:, col]
        assert np.array_equal(out, want), (tag, int((out != want).sum()))
        # one hard argument per wave among 63 easy ones
        hard = xs[np.argsort(dist)[:64]]
        b = rng.uniform(0.05, 1.0, 64 * 64)
        b[::64] = hard
        out = lib.debug_math(np.ones(len(b)), b)[:, col]
        assert np.array_equal(out, oracle.math_fn("pow_" + tag, b)) and np.array_equal(out[::64], want[np.argsort(dist)[:64]])


def test_step_probe_reproduces_the_oracle_trace(lib):
    """Every step ATTEMPT of a ray, one at a time: the device's rk_step + error norm + controller
    power from the oracle's (t, y, h) must give the oracle's y_new, f_new, error_norm and power bit
    for bit (pgr_debug_step: the tool that found the ERR_HI and 1/sqrt(1 - 2^-53) cases)."""
    arrs = munk_arrays(300e3)
    env = lib.EnvHandle(*arrs)
    y0 = y0_for(oracle, arrs, 1000.0, 0.0, [-17.3, -6.0, 0.4, 11.9])
    for k in range(len(y0)):
        tr = oracle.trace_ray(*arrs, y0[k], 0.0, 300e3, math=oracle.MATH_CR)
        d = env.debug_step(tr[:, 0], tr[:, 2:5], tr[:, 1])
        assert np.array_equal(d[:, 8:11], tr[:, 5:8])          # f at the start of the attempt
        assert np.array_equal(d[:, 6], tr[:, 8])               # error norm
        assert np.array_equal(d[:, 7], 0.9 * oracle.math_fn("pow_m02", tr[:, 8]))
        nxt = np.arange(1, len(tr))
        cont = (tr[:-1, 9] == 1) & (tr[nxt, 11] == tr[:-1, 11]) & (tr[nxt, 0] == tr[:-1, 0] + tr[:-1, 1])
        assert cont.sum() > 100
        assert np.array_equal(d[:-1][cont, 0:3], tr[nxt][cont, 2:5])   # y_new = the next attempt's y
        assert np.array_equal(d[:-1][cont, 3:6], tr[nxt][cont, 5:8])   # f_new = its f (FSAL)
    env.close()


def test_exact_bisection_flag_agrees_with_default_locator(lib):
    """PGR_EXACT_BISECTION runs brentq's ~42-step bisection with the true +-1 event at every
    iterate; the default locator replays the same iterates and evaluates the event only inside the
    rounding-noise band around the Newton root.  Both return brentq's root: every output bit equal,
    and equal to the oracle's (which calls a transcription of brentq itself) -- flat and sloping
    floors, range dependent tables, the first bounces near x = 0 included."""
    arrs = munk_arrays(300e3)
    y0 = y0_for(oracle, arrs, 1000.0, 0.0, np.linspace(15.0, 20.0, 128))
    z = np.arange(0, 5500, 2.0)
    r = np.linspace(0, 150e3, 61)
    cin = np.array([munk(z, 1300 + 2e-3 * ri) for ri in r])
    br = np.linspace(0, 150e3, 31)
    depths = 4800 + 300 * np.sin(br / 20e3)
    arrs2 = [cin, np.gradient(cin, z, axis=1, edge_order=1), r, z, depths, br, np.degrees(np.arctan(np.gradient(depths, br)))]
    y02 = y0_for(oracle, arrs2, 700.0, 0.0, np.linspace(-18, 18, 130))
    for ar, yy, x1 in ((arrs, y0, 300e3), (arrs2, y02, 140e3)):
        env = lib.EnvHandle(*ar)
        a = env.shoot_fan(yy, 0.0, x1, 31, exact_samples=True)
        b = env.shoot_fan(yy, 0.0, x1, 31, exact_samples=True, exact_bisection=True)
        o = oracle.shoot_fan(*ar, yy, 0.0, x1, 31, math=oracle.MATH_CR)
        assert (o["n_bott"] + o["n_surf"]).max() >= 5
        for k in ("T", "z", "p", "end"):
            assert np.array_equal(a[k], b[k], equal_nan=True), k
        for k in ("status", "n_steps", "n_rej", "n_bott", "n_surf"):
            assert np.array_equal(a[k], b[k]), k
        assert_bit_parity(b, o, label="exact bisection vs oracle")
        env.close()


def test_edge_shapes_and_inputs(lib):
    arrs = munk_arrays(50e3, nr=12)
    env = lib.EnvHandle(*arrs)
    y1 = y0_for(oracle, arrs, 800.0, 0.0, [3.0])
    # one ray, one / two save points
    for S in (1, 2, 3):
        g = env.shoot_fan(y1, 0.0, 50e3, S)
        o = oracle.shoot_fan(*arrs, y1, 0.0, 50e3, S)
        assert g["T"].shape == (1, S) and g["status"][0] == 0
        np.testing.assert_allclose(g["z"], o["z"], rtol=0, atol=5e-5)
        assert g["z"][0, -1] == g["end"][0, 1]
    # exactly one wave, one wave + 1, and a ragged multi-block fan give the same per-ray results
    th = np.linspace(-18, 18, 193)
    y = y0_for(oracle, arrs, 800.0, 0.0, th)
    full = env.shoot_fan(y, 0.0, 50e3, 11)
    for n in (64, 65, 129):
        part = env.shoot_fan(y[:n], 0.0, 50e3, 11)
        assert np.array_equal(part["z"], full["z"][:n]) and np.array_equal(part["n_steps"], full["n_steps"][:n])
    # a NaN / inf initial state cannot hang the kernel: the ray is dropped with a status
    bad = y[:4].copy()
    bad[0, 2] = np.nan
    bad[1, 1] = np.inf
    bad[2, 2] = 1.0          # |p c| >> 1: clamp path (Q8), then bbox / step failure
    g = env.shoot_fan(bad, 0.0, 50e3, 5, max_steps=5000)
    assert np.all(g["status"][:3] != 0) and g["status"][3] == 0
    assert np.all(np.isnan(g["z"][:3])) and np.all(np.isfinite(g["z"][3]))
    # source beyond the table's last range column: bounding-box event fires immediately
    g = env.shoot_fan(y[:2], 49e3, 80e3, 5)
    assert np.all(g["status"] == 2)
    # not terminate_backwards: a backwards bounce continues (reference default is True)
    arrs2 = munk_arrays(50e3, nr=20, z=np.linspace(0, 6000, 601))
    arrs2[4] = np.where(arrs2[2] > 20e3, 1000.0, 5000.0).astype(float)
    arrs2[6] = np.degrees(np.arctan(np.gradient(arrs2[4], arrs2[5])))
    yb = y0_for(oracle, arrs2, 500.0, 0.0, [12.0])
    e2 = lib.EnvHandle(*arrs2)
    assert e2.shoot_fan(yb, 0.0, 50e3, 5)["status"][0] == 3
    g = e2.shoot_fan(yb, 0.0, 50e3, 5, terminate_backwards=False, max_steps=20000)
    o = oracle.shoot_fan(*arrs2, yb, 0.0, 50e3, 5, terminate_backwards=False, max_steps=20000)
    assert g["status"][0] == o["status"][0]


def test_kernel_packed_end_records(lib):
    """PGR_PACKED_END: the kernel writes the all-gather's 40-byte end records itself; they equal
    pack_end_records() of the separate outputs, padding rows stay zero."""
    import torch
    from pygenray_amd.device_fan import DeviceFan
    from pygenray_amd.distributed import pack_end_records, start_all_gather_records
    arrs = munk_arrays(80e3)
    env = lib.EnvHandle(*arrs)
    th = np.concatenate([np.linspace(-20, 20, 150), [89.9995]])
    y0 = y0_for(oracle, arrs, 1000.0, 0.0, th)
    plain = DeviceFan(env, y0, 0.0, 80e3, 2, save=False)
    plain.run()
    packed = DeviceFan(env, y0, 0.0, 80e3, 2, save=False, packed_end=True, n_pad=160)
    packed.run(); torch.cuda.synchronize()
    want = pack_end_records(plain.end, plain.n_bott, plain.n_surf, plain.status, 160)
    assert packed.records.shape == (160, 5)
    assert torch.equal(packed.records.view(torch.int64), want.view(torch.int64))   # bitwise, NaNs included
    assert torch.equal(packed.end.nan_to_num(), plain.end.nan_to_num()) and torch.equal(packed.status, plain.status)
    e, nb, ns, st = start_all_gather_records(packed.records[:151].contiguous(), 151).finish()  # one rank: a copy
    assert torch.equal(st, plain.status) and torch.equal(nb, plain.n_bott)


def test_sharded_hip_compute_single_rank(lib):
    """distributed.shoot_fan_sharded with the HIP compute callback (world size 1 here; the
    collective itself is covered under gloo in tests/test_host.py)."""
    import torch
    from pygenray_amd.distributed import shoot_fan_sharded, hip_compute, arrival_time_histogram
    arrs = munk_arrays(100e3)
    y0 = y0_for(oracle, arrs, 1000.0, 0.0, np.linspace(-15, 15, 300))
    env = lib.EnvHandle(*arrs)
    end, nb, ns, st = shoot_fan_sharded(hip_compute(env, 0.0, 100e3), y0)
    ref = env.shoot_fan(y0, 0.0, 100e3, 1, save=False)
    assert np.array_equal(end.cpu().numpy(), ref["end"], equal_nan=True)
    assert np.array_equal(nb.cpu().numpy(), ref["n_bott"]) and np.array_equal(st.cpu().numpy(), ref["status"])
    h = arrival_time_histogram(end[:, 0], st, 32, 66.0, 68.0)
    assert int(h.sum().item()) == int(((ref["status"] == 0) & (ref["end"][:, 0] >= 66) & (ref["end"][:, 0] <= 68)).sum())


def test_sharded_api_single_rank_rccl_equals_the_single_process_api(lib):
    """pygenray_amd.distributed's API over RCCL (backend "nccl") with ONE rank on this one-GPU box: the gathered
    end-state fan equals the last column of pr.shoot_rays' fan, find_eigenrays_sharded equals pr.find_eigenrays
    (same brackets, same eigenrays to the bit: both run pgr_eigen_refine), the all-reduced histogram equals
    np.histogram.  (Two ranks, with brackets straddling them and a dropped ray inside a bracket: the gloo test in
    tests/test_host.py.)"""
    import os
    import socket
    import torch
    import torch.distributed as dist
    import pygenray_amd as pr
    from pygenray_amd.distributed import shoot_rays_sharded, find_eigenrays_sharded, arrival_histogram_sharded
    z = np.arange(0, 6000, 1.0); r = np.linspace(0, 100e3, 100)
    env = pr.OceanEnvironment2D(pr.DataArray(np.tile(pr.munk_ssp(z), (100, 1)), dims=["range", "depth"], coords={"range": r, "depth": z}),
                                pr.DataArray(np.full(100, 5000.0), dims=["range"], coords={"range": r}), flat_earth_transform=False)
    theta = np.linspace(-19.9, 19.9, 2001)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        fan = shoot_rays_sharded(1000.0, 0.0, theta, 100e3, env, flatearth=False)
        again = shoot_rays_sharded(1000.0, 0.0, theta[::2], 100e3, env, flatearth=False)     # re-uses the cached buffers
        er = find_eigenrays_sharded(fan, [800.0, 1000.0], 1000.0, 0.0, 100e3, 41, env, debug=False, flatearth=False, quiet=True)
        h, edges = arrival_histogram_sharded(1000.0, 0.0, theta, 100e3, env, 64, 66.0, 68.0, flatearth=False)
    finally:
        dist.destroy_process_group()
    ref = pr.shoot_rays(1000.0, 0.0, theta, 100e3, 3, env, debug=False, flatearth=False)
    assert len(fan) == len(ref) and np.array_equal(fan.thetas, ref.thetas)
    for a_, b_ in ((fan.ts, ref.ts), (fan.zs, ref.zs), (fan.ps, ref.ps)):
        assert a_.shape == (len(ref), 1) and np.array_equal(a_[:, 0], b_[:, -1])
    assert np.array_equal(fan.n_botts, ref.n_botts) and np.array_equal(fan.n_surfs, ref.n_surfs)
    ref2 = pr.shoot_rays(1000.0, 0.0, theta[::2], 100e3, 3, env, debug=False, flatearth=False)
    assert np.array_equal(again.thetas, ref2.thetas) and np.array_equal(again.zs[:, 0], ref2.zs[:, -1])
    er1 = pr.find_eigenrays(ref, [800.0, 1000.0], 1000.0, 0.0, 100e3, 41, env, debug=False, flatearth=False, quiet=True)
    for k in (0, 1):
        assert er.num_eigenrays_found[k] == er1.num_eigenrays_found[k] >= 3
        assert np.array_equal(er.launch_angles[k], er1.launch_angles[k]) and np.array_equal(er.zs[k], er1.zs[k])
        assert np.array_equal(er.ts[k], er1.ts[k]) and er.failed_eray_theta_brackets[k] == er1.failed_eray_theta_brackets[k]
    want = np.histogram(ref.ts[:, -1], bins=64, range=(66.0, 68.0))[0]
    assert np.array_equal(h, want) and h.sum() > 0 and len(edges) == 65


def test_device_resident_fan_equals_the_eager_fan(lib):
    """pgr_fan_* (a fan whose results stay in HBM) and the RayFan on top of it: the same numbers as the host-pointer
    entry -- per-ray arrays, trajectories with and without the compaction of dropped rays, each array fetched on its
    own -- and pr.shoot_rays(device_resident=True) equals device_resident=False attribute by attribute, dropped rays
    gone, the trajectories crossing PCIe only when read."""
    import pygenray_amd as pr
    arrs = munk_arrays(100e3, z=np.arange(0, 4000, 1.0), bathy=5000.0)     # table ends above the sea floor: deep rays leave it
    theta = np.linspace(-20, 20, 300)
    y0 = y0_for(oracle, arrs, 1000.0, 0.0, -theta)
    env = lib.EnvHandle(*arrs)
    ref = env.shoot_fan(y0, 0.0, 100e3, 41, sample_major=True, stored_sign=True)
    drop = ref["status"] != 0
    assert 10 < drop.sum() < 200
    h = lib.FanHandle(env, 0.0, 100e3, 41, y0=y0, stored_sign=True)
    n, m = h.wait()
    assert (n, m) == (300, int((~drop).sum()))
    rays = h.fetch_rays()
    for k in ("end", "n_bott", "n_surf", "status", "n_steps", "n_rej"):
        assert np.array_equal(rays[k], ref[k], equal_nan=True), k
    full = h.fetch_samples(compact=False)
    for k in "Tzp":
        assert full[k].shape == (41, 300) and np.array_equal(full[k], ref[k], equal_nan=True), k
    rc_ = h.fetch_rays_compact(per_ray=theta)
    assert np.array_equal(rc_["end"], ref["end"][~drop]) and np.array_equal(rc_["per_ray"], theta[~drop])
    assert rc_["n_bott"].dtype == np.int64 and np.array_equal(rc_["n_bott"], ref["n_bott"][~drop]) and np.array_equal(rc_["n_surf"], ref["n_surf"][~drop])
    hp = lib.FanHandle(env, 0.0, 100e3, 0, p0=y0[:, 2], source_depth=1000.0)         # [0, z_s, p0] assembled on the device
    assert np.array_equal(hp.fetch_rays()["end"], ref["end"], equal_nan=True)
    hp.close()
    only_z = h.fetch_samples(("z",))
    assert list(only_z) == ["z"] and np.array_equal(only_z["z"], ref["z"][:, ~drop])
    h.close()
    # end states only, initial states computed on the device from the angles (correctly rounded sine)
    c0 = oracle.bilinear(0.0, 1000.0, arrs[2], arrs[3], arrs[0])
    h2 = lib.FanHandle(env, 0.0, 100e3, 0, ode_angles_deg=-theta, source_depth=1000.0, c_source=c0)
    r2 = h2.fetch_rays()
    p0 = oracle.math_fn("sin", np.radians(-theta)) / c0
    ref2 = env.shoot_fan(np.stack([np.zeros(300), np.full(300, 1000.0), p0], 1), 0.0, 100e3, 1, save=False)
    assert np.array_equal(r2["end"], ref2["end"], equal_nan=True) and np.array_equal(r2["status"], ref2["status"])
    with pytest.raises(lib.PgrError):
        h2.fetch_samples()
    h2.close()
    env.close()
    # the drop-in API, both ways
    z = np.arange(0, 4000, 1.0); r = np.linspace(0, 100e3, 100)
    eo = pr.OceanEnvironment2D(pr.DataArray(np.tile(pr.munk_ssp(z), (100, 1)), dims=["range", "depth"], coords={"range": r, "depth": z}),
                               pr.DataArray(np.full(100, 5000.0), dims=["range"], coords={"range": r}), flat_earth_transform=False)
    a = pr.shoot_rays(1000.0, 0.0, theta, 100e3, 41, eo, debug=False, flatearth=False, device_resident=False)
    b = pr.shoot_rays(1000.0, 0.0, theta, 100e3, 41, eo, debug=False, flatearth=False, device_resident=True)
    assert not a.device_resident and b.device_resident and len(a) == len(b) == m
    assert np.array_equal(b.zs_end, a.zs[:, -1]) and np.array_equal(b.ts_end, a.ts[:, -1]) and b.device_resident
    for k in ("thetas", "n_botts", "n_surfs", "source_depths", "rs", "zs", "ts", "ps", "ray_ids"):
        assert np.array_equal(getattr(a, k), getattr(b, k)), k
    assert not b.device_resident and b[3:7].zs.shape == (4, 41) and len(a + b) == 2 * m
    # backwards shot, device resident
    c_ = pr.shoot_rays(1000.0, 100e3, theta, 0.0, 41, eo, debug=False, flatearth=False, device_resident=True)
    d_ = pr.shoot_rays(1000.0, 100e3, theta, 0.0, 41, eo, debug=False, flatearth=False, device_resident=False)
    assert np.array_equal(c_.rs, d_.rs) and np.array_equal(c_.zs, d_.zs) and np.array_equal(c_.thetas, d_.thetas)


def test_device_resident_fan_pickles_and_releases(lib):
    """a device-resident RayFan (what pr.shoot_rays returns for large fans) holds a ctypes device pointer --
    pickle / copy.deepcopy fetch it first and give a plain host fan; to_host() does the same in place; release() gives the
    HBM back without fetching (unread arrays are gone, end states stay)."""
    import copy
    import pickle
    import pygenray_amd as pr
    z = np.arange(0, 6000, 1.0); r = np.linspace(0, 100e3, 100)
    env = pr.OceanEnvironment2D(pr.DataArray(np.tile(pr.munk_ssp(z), (100, 1)), dims=["range", "depth"], coords={"range": r, "depth": z}),
                                pr.DataArray(np.full(100, 5000.0), dims=["range"], coords={"range": r}), flat_earth_transform=False)
    theta = np.linspace(-20, 20, 400)
    eager = pr.shoot_rays(1000.0, 0.0, theta, 100e3, 31, env, debug=False, flatearth=False, device_resident=False)
    fan = pr.shoot_rays(1000.0, 0.0, theta, 100e3, 31, env, debug=False, flatearth=False, device_resident=True)
    assert fan.device_resident
    _ = fan.zs                                     # one of the three read: still resident
    assert fan.device_resident
    b = pickle.loads(pickle.dumps(fan))
    assert not fan.device_resident and not b.device_resident     # the pickle fetched the rest and closed the handle
    c = copy.deepcopy(pr.shoot_rays(1000.0, 0.0, theta, 100e3, 31, env, debug=False, flatearth=False, device_resident=True))
    for x in (b, c, fan):
        for k in ("thetas", "rs", "ts", "zs", "ps", "n_botts", "n_surfs", "source_depths", "ray_ids"):
            assert np.array_equal(getattr(x, k), getattr(eager, k)), k
    d = pr.shoot_rays(1000.0, 0.0, theta, 100e3, 31, env, debug=False, flatearth=False, device_resident=True)
    assert d.to_host() is d and not d.device_resident and np.array_equal(d.ps, eager.ps)
    e = pr.shoot_rays(1000.0, 0.0, theta, 100e3, 31, env, debug=False, flatearth=False, device_resident=True)
    zs = e.zs
    e.release()
    assert not e.device_resident and np.array_equal(zs, eager.zs) and np.array_equal(e.zs_end, eager.zs[:, -1])
    with pytest.raises(AttributeError):
        e.ts
    # ... and the reference's pure functions are not served from tables uploaded before an in-place edit
    arrs = [np.array(a) for a in pr._unpack_envi(env, flatearth=False)]
    y = np.array([0.0, 1000.0, 1e-4])
    d0 = pr.derivsrd(10e3, y, *arrs[:6])
    arrs[0] += 1.0                                  # same array objects, new content
    d1 = pr.derivsrd(10e3, y, *arrs[:6])
    fresh = [a.copy() for a in arrs]
    d2 = pr.derivsrd(10e3, y, *fresh[:6])
    assert not np.array_equal(d0, d1) and np.array_equal(d1, d2)
    from pygenray_amd import host_physics
    host_physics.clear_eval_cache()


def test_environment_closed_before_its_device_resident_fan(lib):
    """pgr_env_destroy while a device-resident fan of the environment is still in flight: the release is deferred to the
    fan's own destruction (the fan uses the environment's stream, tables and buffer pool), so the fan still delivers --
    and equals the fan shot through the host entry."""
    arrs = munk_arrays(100e3)
    y0 = y0_for(oracle, arrs, 1000.0, 0.0, -np.linspace(-15, 15, 4096))
    env = lib.EnvHandle(*arrs)
    ref = env.shoot_fan(y0, 0.0, 100e3, 51, sample_major=True)
    for rep in range(3):
        e2 = lib.EnvHandle(*arrs)
        fan = lib.FanHandle(e2, 0.0, 100e3, 51, y0=y0)
        e2.close()                      # kernel possibly still running
        assert e2._h is None
        got = fan.fetch_rays()
        got.update(fan.fetch_samples(compact=False))
        for name in ("end", "n_bott", "n_surf", "status", "n_steps", "T", "z", "p"):
            assert np.array_equal(got[name], ref[name], equal_nan=True), (rep, name)
        fan.close()                     # ... and the environment goes with it
    env.close()


def test_two_device_resident_fans_in_flight_fetched_from_two_threads(lib):
    """Two fans launched back to back on ONE environment (both kernels in flight on its stream, buffers from its pool) and
    fetched concurrently from two host threads -- per-ray arrays, then all three trajectory arrays through the pipelined
    copy (helper threads faulting and page-locking two sets of destination buffers at once): each equals the fan shot alone."""
    import threading
    arrs = munk_arrays(300e3)
    env = lib.EnvHandle(*arrs)
    ys = [y0_for(oracle, arrs, 1000.0, 0.0, -np.linspace(-18, 18, n)) for n in (40_000, 30_000)]
    S = 301                                   # 96 MB / 72 MB per array: above the pipelined copy's threshold
    alone = [env.shoot_fan(y, 0.0, 300e3, S, sample_major=True, stored_sign=True) for y in ys]
    for rep in range(2):
        fans = [lib.FanHandle(env, 0.0, 300e3, S, y0=y, stored_sign=True) for y in ys]
        got = [None, None]

        def fetch(k):
            r = fans[k].fetch_rays()
            r.update(fans[k].fetch_samples(compact=False))
            got[k] = r
        th = [threading.Thread(target=fetch, args=(k,)) for k in (0, 1)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        for k in (0, 1):
            for name in ("end", "n_bott", "n_surf", "status", "n_steps", "T", "z", "p"):
                assert np.array_equal(got[k][name], alone[k][name], equal_nan=True), (rep, k, name)
            fans[k].close()
    env.close()


def test_many_fans_in_flight_on_user_streams_keep_their_wave_maps(lib):
    """VERDICT r02 'robustness': 12 fans in flight at once on 12 user streams through pgr_shoot_fan_device on ONE
    environment, each big enough for the cost-aware wave placement (a per-launch map in device memory): every fan's
    result equals the one-at-a-time result -- no launch's map is handed to another while its kernel may read it."""
    import torch
    from pygenray_amd.device_fan import DeviceFan, fan_y0
    arrs = munk_arrays(200e3)
    env = lib.EnvHandle(*arrs)
    fans, streams = [], []
    for k in range(12):
        n = 90_000 + 1000 * k       # 1407 .. 1579 waves: single-round placement, a different map each
        y0 = fan_y0(arrs, 1000.0, 0.0, -np.linspace(-20 + 0.1 * k, 20, n))
        fans.append(DeviceFan(env, y0, 0.0, 200e3, 1, save=False))
        streams.append(torch.cuda.Stream())
    ref = []
    for f in fans:
        f.run()
        torch.cuda.synchronize()
        ref.append((f.end.clone(), f.n_steps.clone(), f.status.clone()))
        f.end.zero_(); f.n_steps.zero_()
    torch.cuda.synchronize()
    for rep in range(2):
        for f, st in zip(fans, streams):
            with torch.cuda.stream(st):
                f.run()
        torch.cuda.synchronize()
        for f, (e, ns_, s_) in zip(fans, ref):
            assert torch.equal(torch.nan_to_num(f.end), torch.nan_to_num(e)) and torch.equal(f.n_steps, ns_) and torch.equal(f.status, s_)
    env.close()


def test_persistent_fans_in_flight_on_user_streams_keep_their_packet_queues(lib):
    """Four fans of more than one round -- persistent waves, each launch with its own cost-sorted list and its own queue counter
    in device memory -- in flight at once on four user streams of ONE environment (an LDS-table and an HBM-table one):
    every fan's result equals the one-at-a-time result, twice over (the slots are re-used by the second round)."""
    import torch
    from pygenray_amd.device_fan import DeviceFan, fan_y0
    for arrs in (munk_arrays(60e3, nr=12), munk_arrays(60e3, nr=13, sofar_slope=1e-3)):
        env = lib.EnvHandle(*arrs)
        fans, streams = [], []
        for k in range(4):
            n = 135_000 + 40_000 * k      # 2110 .. 3985 packets: one to two rounds (the cheap-end rule) -- and, last, beyond
            y0 = fan_y0(arrs, 900.0, 0.0, -np.linspace(-19.5 + 0.1 * k, 19.5, n))
            fans.append(DeviceFan(env, y0, 0.0, 60e3, 1, save=False))
            streams.append(torch.cuda.Stream())
        y0 = fan_y0(arrs, 900.0, 0.0, -np.linspace(-19.5, 19.5, 300_000))
        fans.append(DeviceFan(env, y0, 0.0, 60e3, 1, save=False)); streams.append(torch.cuda.Stream())
        ref = []
        for f in fans:
            f.run()
            torch.cuda.synchronize()
            ref.append((f.end.clone(), f.n_steps.clone(), f.status.clone()))
            f.end.zero_(); f.n_steps.zero_(); f.status.fill_(-7)
        torch.cuda.synchronize()
        for rep in range(2):
            for f, st in zip(fans, streams):
                with torch.cuda.stream(st):
                    f.run()
            torch.cuda.synchronize()
            for f, (e, ns_, s_) in zip(fans, ref):
                assert torch.equal(torch.nan_to_num(f.end), torch.nan_to_num(e)) and torch.equal(f.n_steps, ns_) and torch.equal(f.status, s_)
                f.end.zero_(); f.n_steps.zero_(); f.status.fill_(-7)
        env.close()


_TWO_RANK_WORKER = r"""
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
import pygenray_amd as pr
from pygenray_amd.distributed import shoot_rays_sharded, find_eigenrays_sharded, arrival_histogram_sharded
rank = int(sys.argv[1])
torch.cuda.set_device(0)                     # both ranks share the box's one GPU: the collective goes through gloo
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%(port)d", rank=rank, world_size=2)
z = np.arange(0, 6000, 1.0); r = np.linspace(0, 100e3, 100)
env = pr.OceanEnvironment2D(pr.DataArray(np.tile(pr.munk_ssp(z), (100, 1)), dims=["range", "depth"], coords={"range": r, "depth": z}),
                            pr.DataArray(np.full(100, 5000.0), dims=["range"], coords={"range": r}), flat_earth_transform=False)
theta = np.linspace(-19.9, 19.9, 2001)
fan = shoot_rays_sharded(1000.0, 0.0, theta, 100e3, env, flatearth=False, device=0)
er = find_eigenrays_sharded(fan, [800.0, 1000.0], 1000.0, 0.0, 100e3, 41, env, debug=False, flatearth=False, quiet=True, device=0)
h, edges = arrival_histogram_sharded(1000.0, 0.0, theta, 100e3, env, 64, 66.0, 68.0, flatearth=False, device=0)
out = dict(thetas=fan.thetas, ts=fan.ts, zs=fan.zs, ps=fan.ps, nb=fan.n_botts, ns=fan.n_surfs, hist=h)
for k in (0, 1):
    out[f"e{k}_th"] = er.launch_angles[k]; out[f"e{k}_ts"] = er.ts[k]; out[f"e{k}_zs"] = er.zs[k]; out[f"e{k}_nb"] = er.n_botts[k]
    out[f"e{k}_failed"] = np.array(er.failed_eray_theta_brackets[k], dtype=float).reshape(-1, 2)
np.savez(sys.argv[2], **out)
dist.barrier(); dist.destroy_process_group()
print("RANK_OK")
"""


def test_sharded_api_two_ranks_on_one_gpu_equal_the_single_process_api(lib, tmp_path):
    """The sharded path with the REAL HIP fan and the real device refinement on TWO ranks: both processes drive this
    box's one GPU (the kernel writes the packed end records, each rank refines its share of the brackets with
    pgr_eigen_refine), the collectives go through gloo (RCCL wants one GPU per rank).  Every rank's fan, EigenRays and
    histogram equal pr.shoot_rays / pr.find_eigenrays / np.histogram in this process, to the bit."""
    import socket
    import subprocess
    import sys
    import os
    import pygenray_amd as pr
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    w = tmp_path / "worker.py"
    w.write_text(_TWO_RANK_WORKER % dict(root=root, port=port))
    procs = [subprocess.Popen([sys.executable, str(w), str(rk), str(tmp_path / f"rank{rk}.npz")], stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for rk in range(2)]
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0 and "RANK_OK" in o, e[-3000:]
    z = np.arange(0, 6000, 1.0); r = np.linspace(0, 100e3, 100)
    env = pr.OceanEnvironment2D(pr.DataArray(np.tile(pr.munk_ssp(z), (100, 1)), dims=["range", "depth"], coords={"range": r, "depth": z}),
                                pr.DataArray(np.full(100, 5000.0), dims=["range"], coords={"range": r}), flat_earth_transform=False)
    theta = np.linspace(-19.9, 19.9, 2001)
    ref = pr.shoot_rays(1000.0, 0.0, theta, 100e3, 3, env, debug=False, flatearth=False)
    er1 = pr.find_eigenrays(ref, [800.0, 1000.0], 1000.0, 0.0, 100e3, 41, env, debug=False, flatearth=False, quiet=True)
    want_h = np.histogram(ref.ts[:, -1], bins=64, range=(66.0, 68.0))[0]
    for rk in range(2):
        g = np.load(tmp_path / f"rank{rk}.npz")
        assert np.array_equal(g["thetas"], ref.thetas) and np.array_equal(g["zs"][:, 0], ref.zs[:, -1])
        assert np.array_equal(g["ts"][:, 0], ref.ts[:, -1]) and np.array_equal(g["ps"][:, 0], ref.ps[:, -1])
        assert np.array_equal(g["nb"], ref.n_botts) and np.array_equal(g["ns"], ref.n_surfs) and np.array_equal(g["hist"], want_h)
        for k in (0, 1):
            assert len(g[f"e{k}_th"]) == er1.num_eigenrays_found[k] >= 3
            assert np.array_equal(g[f"e{k}_th"], er1.launch_angles[k]) and np.array_equal(g[f"e{k}_zs"], er1.zs[k])
            assert np.array_equal(g[f"e{k}_ts"], er1.ts[k]) and np.array_equal(g[f"e{k}_nb"], er1.n_botts[k])
            assert np.array_equal(g[f"e{k}_failed"], np.array(er1.failed_eray_theta_brackets[k], dtype=float).reshape(-1, 2))


def test_bench_two_ranks_on_one_gpu_carries_the_multi_gpu_legs(lib):
    """`python bench.py --gpus 2` end to end, the way the driver's N > 1 runs go (the launcher starts the ranks, rank 0 prints ONE
    JSON line) -- rehearsed on this box's one GPU over gloo (PGR_BENCH_ONE_GPU=1: RCCL wants one GPU per rank): the line
    carries `ranks_joined`, the configs[4] leg (1e6 rays per rank, end records all-gathered, histogram all-reduced: every
    surviving ray counted once) and the sharded eigenray search; its whole-job value is steps x passes / wall."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PGR_BENCH_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "2", "--warmup", "1",
                        "--eigen-rays", "200000"], capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_joined"] == 2 and d["scaling"] == "weak" and "REHEARSAL" in d["config"]["sharding"]
    assert abs(d["value"] - d["config"]["ray_steps_per_pass"] * d["steps"] / (d["ms_per_step"] * 1e-3 * d["steps"])) < 1e-6 * d["value"]
    assert 2.8e8 < d["config"]["ray_steps_per_pass"] < 3.0e8          # 2 x 1e5 rays of the 2e5-angle fan
    c4 = d["legs"]["config4"]
    assert c4["ranks"] == 2 and c4["gathered_rays"] == 2_000_000 and c4["histogram_counted_rays"] == c4["gathered_ok"]
    assert c4["gathered_ok"] == 2_000_000 - c4["dropped_rays"] and 2.8e9 < c4["ray_steps_per_pass"] < 3.0e9
    es = d["eigenray_sharded"]
    assert es["ranks"] == 2 and es["found"] + es["failed"] == es["brackets"] and es["found"] > 10
    assert "cpu_baseline" not in d and "api" not in d.get("legs", {})      # rank-0-at-N=1-only legs stay out


def test_arrival_time_histogram_equals_numpy(lib):
    """pgr_arrival_histogram_device (BASELINE configs[4]) against np.histogram, count for count:
    values on bin edges and on the range ends, NaN, dropped rays, strided views, packed end records."""
    import torch
    from pygenray_amd.distributed import arrival_time_histogram
    rng = np.random.default_rng(4)
    n, bins, lo, hi = 300_000, 4096, 660.0, 690.0
    t = rng.uniform(lo - 2, hi + 2, n)
    edges = np.linspace(lo, hi, bins + 1)
    t[:5000] = rng.choice(edges, 5000)                        # exactly on edges, both range ends included
    t[5000:5100] = np.nextafter(rng.choice(edges, 100), np.inf)
    t[5100:5200] = np.nextafter(rng.choice(edges, 100), -np.inf)
    t[5200:5300] = np.nan
    st = np.where(rng.random(n) < 0.05, rng.integers(1, 6, n), 0).astype(np.int32)
    want = np.histogram(t[(st == 0) & ~np.isnan(t)], bins=bins, range=(lo, hi))[0]
    # (a) end-state layout [N][3], time in column 0: a strided view, read in place
    end = torch.zeros(n, 3, dtype=torch.float64, device="cuda")
    end[:, 0] = torch.from_numpy(t).cuda()
    std = torch.from_numpy(st).cuda()
    h = arrival_time_histogram(end[:, 0], std, bins, lo, hi)
    assert h.dtype == torch.int64 and np.array_equal(h.cpu().numpy(), want)
    # (b) the packed 40-byte end records of the multi-GPU path: T in slot 0, status in the low half of slot 4
    rec = torch.zeros(n, 5, dtype=torch.float64, device="cuda")
    rec[:, 0] = end[:, 0]
    rec.view(torch.int32).view(n, 10)[:, 8] = std
    h2 = arrival_time_histogram(rec[:, 0], rec.view(torch.int32).view(n, 10)[:, 8], bins, lo, hi)
    assert np.array_equal(h2.cpu().numpy(), want)
    # (c) few bins, empty input, argument errors
    h3 = arrival_time_histogram(end[:, 0], std, 7, lo, hi)
    assert np.array_equal(h3.cpu().numpy(), np.histogram(t[(st == 0) & ~np.isnan(t)], bins=7, range=(lo, hi))[0])
    h4 = arrival_time_histogram(end[:0, 0], std[:0], 16, lo, hi)
    assert h4.shape == (16,) and int(h4.sum()) == 0
    with pytest.raises(lib.PgrError):
        arrival_time_histogram(end[:, 0], std, 16, hi, lo)
    with pytest.raises(lib.PgrError):
        arrival_time_histogram(end[:, 0], std, 100_000, lo, hi)


def test_wave_scheduler_is_a_pure_permutation(lib):
    """Cost-aware scheduling (placement, priorities, cost-sorted workgroups) only decides where
    and when a wave runs: for fan sizes around every regime boundary the results are bit-identical
    to the plain strided deal and every ray is integrated exactly once."""
    arrs = munk_arrays(60e3, nr=12)
    env = lib.EnvHandle(*arrs)
    cus = 256
    sizes = [4 * cus * 64 + 1, 5 * cus * 64, 7 * cus * 64 - 63, 8 * cus * 64, 8 * cus * 64 + 1, 150_000]
    for n in sizes:
        y0 = y0_for(oracle, arrs, 900.0, 0.0, np.linspace(-19, 19, n))
        env.set_option("placement", 0)
        ref = env.shoot_fan(y0, 0.0, 60e3, 1, save=False)
        for mode in (1, 2):
            env.set_option("placement", mode)
            got = env.shoot_fan(y0, 0.0, 60e3, 1, save=False)
            assert np.array_equal(got["end"], ref["end"], equal_nan=True), (n, mode)
            assert np.array_equal(got["n_steps"], ref["n_steps"]) and np.array_equal(got["status"], ref["status"])
    env.set_option("placement", 2)
    # range-dependent (no LDS table) path as well; options belong to an environment: env's do not leak into env2
    arrs2 = munk_arrays(60e3, nr=13, sofar_slope=1e-3)
    env2 = lib.EnvHandle(*arrs2)
    y0 = y0_for(oracle, arrs2, 900.0, 0.0, np.linspace(-19, 19, 90_000))
    env2.set_option("placement", 0)
    ref = env2.shoot_fan(y0, 0.0, 60e3, 1, save=False)
    env2.set_option("placement", 2)
    got = env2.shoot_fan(y0, 0.0, 60e3, 1, save=False)
    assert np.array_equal(got["end"], ref["end"], equal_nan=True) and np.array_equal(got["status"], ref["status"])
    with pytest.raises(lib.PgrError):
        env2.set_option("park", 0, 5)


def test_persistent_waves_hold_the_same_bits_as_the_static_deal(lib):
    """Fans of several rounds (more 64-ray packets than the chip holds waves) run as PERSISTENT waves that claim packets
    from the cost-sorted list (PGR_OPT_PERSISTENT 1, the default); the static deal of whole workgroups (0) and the
    plain strided deal (placement 0) integrate the same rays: every output array bit-identical, every ray integrated
    exactly once -- LDS-table and HBM-table kernels, end state only, rows, the sample-blocked layout, the cubic-index
    look-up -- and every 400th ray of the largest fan bit-identical to the oracle."""
    from pygenray_amd.device_fan import DeviceFan
    import torch
    cus = 256

    def run(env, y0, x1, S, save, blocked=False):
        fan = DeviceFan(env, y0, 0.0, x1, S, save=save, sample_major=True, sample_blocked=blocked)
        fan.run(); torch.cuda.synchronize()
        out = {k: getattr(fan, k).cpu().numpy() for k in ("end", "n_bott", "n_surf", "status", "n_steps", "n_rej")}
        if save:
            out.update({k: fan.rows(getattr(fan, k)).cpu().numpy() for k in ("T", "Z", "P")})
        return out

    arrs = munk_arrays(80e3, nr=12)
    arrs_rd = munk_arrays(80e3, nr=13, sofar_slope=1e-3)
    import pygenray_amd as pr
    from pygenray_amd.environment import _unpack_envi
    z = np.arange(0, 6000, 1.0); r = np.linspace(0, 80e3, 9)
    eo = pr.OceanEnvironment2D(pr.DataArray(np.tile(pr.munk_ssp(z), (9, 1)), dims=["range", "depth"], coords={"range": r, "depth": z}),
                               pr.DataArray(np.full(9, 5000.0), dims=["range"], coords={"range": r}), flat_earth_transform=True)
    arrs_fe = _unpack_envi(eo, flatearth=True)
    cases = [(arrs, 8 * cus * 64 + 65, 1, False, False), (arrs, 200_001, 7, True, False), (arrs, 16 * cus * 64 + 1, 1, False, False), (arrs_rd, 160_000, 1, False, False),
             (arrs_rd, 140_000, 10, True, True), (arrs_rd, 140_000, 9, True, False), (arrs_fe, 150_000, 5, True, False)]
    for arrs_, n, S, save, blocked in cases:
        env = lib.EnvHandle(*arrs_)
        y0 = y0_for(oracle, arrs_, 900.0, 0.0, np.linspace(-19.5, 19.5, n))
        env.set_option("persistent", 0)
        ref = run(env, y0, 80e3, S, save, blocked)
        for mode in (2, 3, 1):     # every packet from the list's head / waves 4 .. 7 start at its cheap end / the default rule
            env.set_option("persistent", mode)
            got = run(env, y0, 80e3, S, save, blocked)
            for k in ref:
                assert np.array_equal(got[k], ref[k], equal_nan=True), (n, S, blocked, k, mode)
        env.set_option("placement", 0)
        plain = run(env, y0, 80e3, S, save, blocked)
        for k in ref:
            assert np.array_equal(plain[k], ref[k], equal_nan=True), (n, S, blocked, k, "strided")
        if arrs_ is arrs and save:
            sub = np.arange(0, n, 400)
            o = oracle.shoot_fan(*arrs_, y0[sub], 0.0, 80e3, S, math=oracle.MATH_CR)
            assert np.array_equal(o["status"], got["status"][sub]) and np.array_equal(o["n_steps"], got["n_steps"][sub])
            ok = o["status"] == 0
            end_o = np.stack([o["T"][:, -1], o["z"][:, -1], o["p"][:, -1]], 1)
            assert np.array_equal(end_o[ok], got["end"][sub][ok]) and np.array_equal(o["n_rej"][ok], got["n_rej"][sub][ok])
        env.close()


def test_every_kernel_instance_is_launched_and_bit_identical(lib, capsys):
    """The library holds 72 instances of pgr_fan_kernel<LDS_TAB, ZM, SAVE, PERSIST> (csrc/pgr_launch.h: select_variant x sv x
    persistent).  This walks ALL of them: for every (table home, depth look-up) an environment built to select it, for every
    SAVE a fan shaped to need it, small fans for the one-packet-per-wave instances and fans of more packets than the chip
    holds waves for the persistent ones -- each launch is confirmed by pgr_debug_last_instance to be the instance it was
    meant to be, and checked against the oracle by rule (A): status, bounce counts, accepted AND rejected steps, end states,
    samples (SAVE 2: every sample bit for bit; SAVE 1 / 3: bit-equal outside a step, 1e-12 inside)."""
    import torch
    import pygenray_amd as pr
    from pygenray_amd.device_fan import DeviceFan
    z1 = np.arange(0, 6000, 1.0)
    depf = pr.eflat(z1, 35.0, pr.munk_ssp(z1))[0]               # the flat-earth image of a uniform grid: smooth, non-uniform
    grids = {4: (z1, 0), 1: (np.arange(0, 6000, 2.0), 0), 5: (depf, 0), 3: (depf, 3), 2: (depf, 2), 0: (depf, 1)}   # zm -> (zin, PGR_OPT_DEPTH_SEARCH)
    X_SMALL, S_SMALL, N_SMALL = 30e3, 13, 200
    X_BIG, S_BIG, N_BIG = 6e3, 5, 8 * 256 * 64 + 65
    every = 300
    seen = set()

    def env_for(zin, lds):
        nr = 7
        r = np.linspace(0.0, 40e3, nr)
        cin = np.array([munk(zin, 1300.0 + (0.0 if lds else 4e-3 * ri)) for ri in r])
        cpin = np.gradient(cin, zin, axis=1, edge_order=1)
        return [cin, cpin, r, zin, np.full(nr, 4800.0), r.copy(), np.zeros(nr)]

    def launch(env, y0, x1, S, sv, want):
        fan = DeviceFan(env, y0, 0.0, x1, S, save=(sv != 0), sample_major=True, exact_samples=(sv == 2), sample_blocked=(sv == 3))
        fan.run(); torch.cuda.synchronize()
        li = env.last_instance()
        got = (li["lds_tab"], li["zm"], li["save"], li["persist"])
        assert got == want, (got, want, li)
        seen.add(got)
        out = {k: getattr(fan, k).cpu().numpy() for k in ("end", "n_bott", "n_surf", "status", "n_steps", "n_rej")}
        if sv:
            out.update({k.lower() if k != "T" else k: fan.rows(getattr(fan, k)).cpu().numpy().T.copy() for k in ("T", "Z", "P")})
        return out

    def check(out, o, sub, sv, label):
        t = {k: (v[sub] if v.ndim and len(v) == len(out["status"]) else v) for k, v in out.items()}
        if sv:
            assert_bit_parity(t, o, label=label, samples=(sv == 2))
        else:
            assert np.array_equal(t["status"], o["status"]), label
            ok = o["status"] == 0
            end_o = np.stack([o["T"][:, -1], o["z"][:, -1], o["p"][:, -1]], 1)
            for a_, b_ in ((t["end"], end_o), (t["n_bott"], o["n_bott"]), (t["n_surf"], o["n_surf"]),
                           (t["n_steps"].astype(np.int64), o["n_steps"]), (t["n_rej"].astype(np.int64), o["n_rej"])):
                assert np.array_equal(a_[ok], b_[ok]), label

    for lds in (1, 0):
        for zm, (zin, search) in grids.items():
            arrs = env_for(zin, lds)
            env = lib.EnvHandle(*arrs)
            env.set_option("depth_search", search)
            y0s = y0_for(oracle, arrs, 900.0, 0.0, np.linspace(-19.5, 19.5, N_SMALL))
            y0b = y0_for(oracle, arrs, 900.0, 0.0, np.linspace(-19.5, 19.5, N_BIG))
            sub = np.arange(0, N_BIG, every)
            o_small = oracle.shoot_fan(*arrs, y0s, 0.0, X_SMALL, S_SMALL, math=oracle.MATH_CR)
            o_big = oracle.shoot_fan(*arrs, y0b[sub], 0.0, X_BIG, S_BIG, math=oracle.MATH_CR)
            assert ((o_small["n_bott"] + o_small["n_surf"]) > 0).sum() > 20 and ((o_big["n_bott"] + o_big["n_surf"]) > 0).sum() > 20
            for sv in ((0, 1, 2) if lds else (0, 1, 2, 3)):
                out = launch(env, y0s, X_SMALL, S_SMALL, sv, (lds, zm, sv, 0))
                check(out, o_small, np.arange(N_SMALL), sv, f"<{lds},{zm},{sv},0>")
                if sv != 2:
                    out = launch(env, y0b, X_BIG, S_BIG, sv, (lds, zm, sv, 1))
                    check(out, o_big, sub, sv, f"<{lds},{zm},{sv},1>")
            env.close()
    want_all = {(lds, zm, sv, pv) for lds in (1, 0) for zm in range(6) for sv in ((0, 1, 2) if lds else (0, 1, 2, 3))
                for pv in ((0,) if sv == 2 else (0, 1))}
    assert seen == want_all and len(seen) == 72, sorted(want_all - seen)
    with capsys.disabled():
        print(f"\n{len(seen)} / {len(want_all)} reachable instances of pgr_fan_kernel launched and bit-identical to the oracle")


def test_api_callers_of_hbm_table_environments_get_the_sample_blocked_kernel(lib):
    """pgr_shoot_fan (sample-major) and the fan handles integrate trajectory fans of environments whose tables stay in HBM / L2
    with the sample-blocked kernel and un-block in the pass that squeezes dropped rays out (PGR_OPT_API_BLOCKED 1, the
    default): the caller's arrays are the row kernel's, bit for bit -- with and without compaction, for every S mod 4,
    with dropped rays, eager and device resident, through pr.shoot_rays too (REF/launch_rays.py:166-186 is what the caller
    gets either way)."""
    import pygenray_amd as pr
    arrs = munk_arrays(100e3, nr=21, z=np.arange(0, 4000, 1.0), bathy=5000.0, sofar_slope=5e-4)   # range dependent; deep rays leave the table
    theta = np.linspace(-20, 20, 700)
    y0 = y0_for(oracle, arrs, 1000.0, 0.0, -theta)
    env = lib.EnvHandle(*arrs)
    assert not env.lds_path
    for S in (1, 2, 3, 4, 41, 42, 43, 44):
        for compact in (False, True):
            env.set_option("api_blocked", 0)
            ref = env.shoot_fan(y0, 0.0, 100e3, S, sample_major=True, stored_sign=True, compact=compact)
            env.set_option("api_blocked", 1)
            got = env.shoot_fan(y0, 0.0, 100e3, S, sample_major=True, stored_sign=True, compact=compact)
            assert 10 < (ref["status"] != 0).sum() < 600
            for k in ("T", "z", "p", "end", "status", "n_bott", "n_surf", "n_steps", "n_rej"):
                assert got[k].shape == ref[k].shape and np.array_equal(got[k], ref[k], equal_nan=True), (S, compact, k)
    # the oracle on the blocked path's output (rule A: bit for bit, every sample in SciPy's order is not what the default
    # form gives inside a step -- end states and counts are)
    o = oracle.shoot_fan(*arrs, y0[::7], 0.0, 100e3, 44, math=oracle.MATH_CR)
    assert np.array_equal(o["status"], got["status"][::7]) and np.array_equal(o["n_steps"], got["n_steps"][::7])
    ok = o["status"] == 0
    assert np.array_equal(np.stack([o["T"][:, -1], o["z"][:, -1], o["p"][:, -1]], 1)[ok], got["end"][::7][ok])
    # fan handles: compact and full fetches, one array at a time
    drop = ref["status"] != 0
    for S in (41, 43):
        env.set_option("api_blocked", 0)
        ref = env.shoot_fan(y0, 0.0, 100e3, S, sample_major=True, stored_sign=True)
        env.set_option("api_blocked", 1)
        h = lib.FanHandle(env, 0.0, 100e3, S, y0=y0, stored_sign=True)
        full = h.fetch_samples(compact=False)
        for k in "Tzp":
            assert full[k].shape == (S, 700) and np.array_equal(full[k], ref[k], equal_nan=True), (S, k)
        only_p = h.fetch_samples(("p",))
        assert list(only_p) == ["p"] and np.array_equal(only_p["p"], ref["p"][:, ~drop])
        h.close()
    env.close()
    # the drop-in API on a range-dependent environment, eager and device resident
    z = np.arange(0, 4000, 1.0); r = np.linspace(0, 100e3, 21)
    c2 = np.array([pr.munk_ssp(z, 1300.0 + 5e-4 * ri) for ri in r])
    eo = pr.OceanEnvironment2D(pr.DataArray(c2, dims=["range", "depth"], coords={"range": r, "depth": z}),
                               pr.DataArray(np.full(21, 5000.0), dims=["range"], coords={"range": r}), flat_earth_transform=False)
    a = pr.shoot_rays(1000.0, 0.0, theta, 100e3, 41, eo, debug=False, flatearth=False, device_resident=False)
    b = pr.shoot_rays(1000.0, 0.0, theta, 100e3, 41, eo, debug=False, flatearth=False, device_resident=True)
    assert len(a) == len(b) == int((~drop).sum())
    for k in ("thetas", "n_botts", "n_surfs", "rs", "zs", "ts", "ps"):
        assert np.array_equal(getattr(a, k), getattr(b, k)), k


def test_plain_c_example_runs(lib, tmp_path):
    """examples/shoot_fan.c (C99, no Python, no torch) drives the library and gets the oracle's rays."""
    import re
    import subprocess
    from test_host import _build_c_example
    exe = _build_c_example(tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    rows = re.findall(r"ray\s+(\d+): status (\d+), T = ([0-9.]+) s, z = ([-0-9.]+) m, (\d+) steps", out.stdout)
    assert len(rows) == 8
    assert "device-resident fan: 64 of 64 rays kept, 11 samples each, equal to the host-entry fan: yes" in out.stdout
    arrs = munk_arrays(100e3, nr=20)
    th = -15.0 + 30.0 * np.arange(64) / 63
    y0 = y0_for(oracle, arrs, 1000.0, 0.0, th)
    o = oracle.shoot_fan(*arrs, y0, 0.0, 100e3, 2)
    for k, st, T, z, nsteps in rows:
        k = int(k)
        assert int(st) == o["status"][k] == 0
        assert abs(float(T) - o["T"][k, -1]) < 1e-8 * 70 and abs(float(z) - o["z"][k, -1]) < 2e-5
        assert abs(int(nsteps) - o["n_steps"][k]) <= 2


# ------------------------------------------------------------------ BASELINE configs[3] and configs[4] at full size
def _config1_env(pr):
    z = np.arange(0, 6000, 1.0)
    r = np.linspace(0.0, 1000e3, 100)
    return pr.OceanEnvironment2D(pr.DataArray(np.tile(pr.munk_ssp(z), (100, 1)), dims=["range", "depth"], coords={"range": r, "depth": z}),
                                 pr.DataArray(np.full(100, 5000.0), dims=["range"], coords={"range": r}), flat_earth_transform=False)


def test_config3_eigenray_search_on_the_1e6_angle_fan(lib):
    """BASELINE configs[3]: fixed source / receiver, a fan of 1 000 000 launch angles (end state only)
    and pygenray's regula falsi on every bracket, through the drop-in API.  Checked against (i) the
    oracle, bit for bit, on every 200th ray of the fan (5 000 rays); (ii) the REFERENCE's own rays on three
    48-angle windows of the same grid (golden g10): end depths, arrival times, bracket positions;
    (iii) the reference's _find_single_eigenray on those three brackets and on 500 times coarser
    brackets around them: launch angle, arrival time, end depth of the eigenray."""
    import pygenray_amd as pr
    from pygenray_amd.eigenrays import _regula_falsi_batch
    g = load("g10_eigenrays_1000km.npz")
    N, rd, zs, x1 = int(g["n_grid"]), float(g["receiver_depth"]), float(g["source_depth"]), float(g["receiver_range"])
    env = _config1_env(pr)
    grid = np.linspace(-20, 20, N)
    fan = pr.shoot_rays(zs, 0.0, grid, x1, 2, env, debug=False, flatearth=False)
    assert N - 2000 < len(fan) <= N                      # the near-vertical-free fan loses only a few rays (Q12)
    pos = np.searchsorted(fan.thetas, grid)              # grid index -> position in the fan (dropped rays vanish)
    alive = (pos < len(fan)) & (fan.thetas[np.minimum(pos, len(fan) - 1)] == grid)
    # (i) every 200th ray against the oracle
    sub = np.arange(0, N, 200)
    arrs = pr._unpack_envi(env, flatearth=False)
    y0 = y0_for(oracle, arrs, zs, 0.0, -grid[sub])
    o = oracle.shoot_fan(*arrs, y0, 0.0, x1, 2, math=oracle.MATH_CR)
    assert np.array_equal(o["status"] == 0, alive[sub])
    ok = o["status"] == 0
    same = (fan.ts[pos[sub][ok], -1] == o["T"][ok, -1]) & (-fan.zs[pos[sub][ok], -1] == o["z"][ok, -1])
    assert same.all(), (int((~same).sum()), int(same.size))   # every sampled ray bit-identical
    assert np.array_equal(fan.n_botts[pos[sub][ok]], o["n_bott"][ok]) and np.array_equal(fan.n_surfs[pos[sub][ok]], o["n_surf"][ok])
    # (ii) the reference's rays on three windows of the grid
    for j in range(3):
        idx = g[f"w{j}_idx"]
        assert np.all(alive[idx]) and np.all(g[f"w{j}_ok"] == 1)
        # (the reference's own rays: NumPy / SciPy arithmetic, chaotic at the last bit over 1000 km --
        # policy (B) of helpers.py: the median within 1e-8 of the water column, no ray beyond 1e-6)
        dz = np.abs(fan.zs[pos[idx], -1] - g[f"w{j}_z_end"]) / 5000.0
        assert np.median(dz) <= 1e-8 and dz.max() <= 1e-6, (j, np.median(dz), dz.max())
        np.testing.assert_allclose(fan.ts[pos[idx], -1], g[f"w{j}_t_end"], rtol=0, atol=1e-6)
        assert np.array_equal(fan.n_botts[pos[idx]], g[f"w{j}_n_bott"]) and np.array_equal(fan.n_surfs[pos[idx]], g[f"w{j}_n_surf"])
        assert np.array_equal(np.where(np.diff(np.sign(fan.zs[pos[idx], -1] + rd)))[0], g[f"w{j}_starts"])
    # (iii) the search itself
    er = pr.find_eigenrays(fan, [rd], zs, 0.0, x1, 2, env, ztol=1, max_iter=20, debug=False, flatearth=False, quiet=True)
    n_br = int(np.count_nonzero(np.diff(np.sign(fan.zs[:, -1] + rd))))
    assert er.num_eigenrays[rd] == n_br and 60 <= n_br <= 120
    assert er.num_eigenrays_found[0] + len(er.failed_eray_theta_brackets[0]) == n_br
    assert er.num_eigenrays_found[0] >= 0.9 * n_br
    assert np.all(np.abs(er.zs[0][:, -1] + rd) < 1.0)
    for j in range(3):
        ref = g[f"w{j}_eigen"]
        k = int(np.argmin(np.abs(er.launch_angles[0] - ref[0])))
        assert abs(er.launch_angles[0][k] - ref[0]) < 1e-8          # the first false-position angle hits: same angle
        assert abs(er.ts[0][k, -1] - ref[1]) < 1e-6 and abs(er.zs[0][k, -1] - ref[2]) < 5e-3
        assert (int(er.n_botts[0][k]), int(er.n_surfs[0][k])) == (int(ref[4]), int(ref[5]))
        # the 500 times coarser bracket around it (the ends are fan rays 501 grid steps apart)
        ic = g[f"w{j}_coarse_idx"]
        z12 = fan.zs[pos[ic], -1]
        np.testing.assert_allclose(z12, g[f"w{j}_coarse_z"], rtol=0, atol=5e-3)
        found, th, r, T, Z, P, nb, ns = _regula_falsi_batch(z12[:1], z12[1:], grid[ic[:1]], grid[ic[1:]], rd, zs, 0.0, x1, 2, env,
                                                            1, 20, dict(debug=False, flatearth=False, quiet=True))
        refc = g[f"w{j}_coarse_eigen"]
        assert found[0] and abs(th[0] - refc[0]) < 1e-7 and abs(T[0, -1] - refc[1]) < 1e-6 and abs(Z[0, -1] - refc[2]) < 5e-3


def test_receiver_depths_searched_together_equal_one_search_each(lib):
    """find_eigenrays loops over its receiver depths (REF/eigenrays.py:62); here the brackets of ALL depths go through
    the device loop together (pgr_eigen_refine_depths: a receiver depth per bracket).  The EigenRays of one call with
    four depths equals, depth by depth and bit for bit, four calls with one depth each -- in as many fan launches as
    the longest of the four searches alone."""
    import pygenray_amd as pr
    from pygenray_amd import eigenrays as er_mod
    z = np.arange(0, 6000, 1.0)
    r = np.linspace(0.0, 300e3, 60)
    env = pr.OceanEnvironment2D(pr.DataArray(np.tile(pr.munk_ssp(z), (60, 1)), dims=["range", "depth"], coords={"range": r, "depth": z}),
                                pr.DataArray(np.full(60, 5000.0), dims=["range"], coords={"range": r}), flat_earth_transform=False)
    fan = pr.shoot_rays(1000.0, 0.0, np.linspace(-18, 18, 3001), 300e3, 2, env, debug=False, flatearth=False)
    depths = [400.0, 1000.0, 1800.0, 3000.0]
    kw = dict(ztol=1, max_iter=20, debug=False, flatearth=False, quiet=True)
    er_mod.LAST_SEARCH_STATS.clear()
    all4 = pr.find_eigenrays(fan, depths, 1000.0, 0.0, 300e3, 51, env, **kw)
    launches_together = er_mod.LAST_SEARCH_STATS["launches"]
    # the re-shot eigenrays (trajectory kernel) end on the bits of the accepted trial rays (end-state kernel)
    assert er_mod.LAST_SEARCH_STATS["reshot_differs"] == 0
    launches_alone = []
    for k, rd in enumerate(depths):
        er_mod.LAST_SEARCH_STATS.clear()
        one = pr.find_eigenrays(fan, [rd], 1000.0, 0.0, 300e3, 51, env, **kw)
        launches_alone.append(er_mod.LAST_SEARCH_STATS.get("launches", 0))
        assert all4.num_eigenrays[rd] == one.num_eigenrays[rd] and all4.num_eigenrays_found[k] == one.num_eigenrays_found[0]
        assert all4.failed_eray_theta_brackets[k] == one.failed_eray_theta_brackets[0]
        assert one.num_eigenrays_found[0] >= 3
        for name in ("launch_angles", "ts", "zs", "ps", "rs", "n_botts", "n_surfs"):
            assert np.array_equal(getattr(all4, name)[k], getattr(one, name)[0]), (rd, name)
        assert np.all(np.abs(all4.zs[k][:, -1] + rd) < 1.0)
    # (each search: its loop + one re-shoot of the eigenrays found; together: the longest loop + one re-shoot)
    assert launches_together == max(launches_alone) and sum(launches_alone) > 2 * launches_together
    # ONE arithmetic for the initial slowness (NumPy's sin(radians(.)) / c, REF/launch_rays.py:284-285) in the fan, in the
    # search's trial rays and in the eigenrays handed back: pr.shoot_ray(theta) of an eigenray's launch angle IS that eigenray
    for k in (0, 2):
        for q in range(min(3, len(all4.launch_angles[k]))):
            ray = pr.shoot_ray(1000.0, 0.0, float(all4.launch_angles[k][q]), 300e3, 51, env, debug=False, flatearth=False)
            assert ray is not None and np.array_equal(ray.z, all4.zs[k][q]) and np.array_equal(ray.t, all4.ts[k][q]) \
                and np.array_equal(ray.p, all4.ps[k][q]), (k, q)
            assert ray.launch_angle == -all4.launch_angles[k][q]       # (Q2: shoot_ray stores the negated angle)


def test_config4_end_records_and_arrival_time_histogram_of_1e6_rays(lib):
    """BASELINE configs[4], one GPU's share: 1 000 000 rays, end state only, the kernel writes the
    40-byte end records of the all-gather (PGR_PACKED_END); the 4096-bin arrival-time histogram on the
    device (pgr_arrival_histogram_device, read straight from the records) equals np.histogram of the
    same end states count for count; the single-rank all-gather hands the fan back in launch order."""
    import torch
    from pygenray_amd.device_fan import DeviceFan, fan_y0
    from pygenray_amd.distributed import start_all_gather_records, arrival_time_histogram
    arrs = munk_arrays(1000e3)
    n = 1_000_000
    y0 = fan_y0(arrs, 1000.0, 0.0, -np.linspace(-20, 20, n))
    env = lib.EnvHandle(*arrs)
    fan = DeviceFan(env, y0, 0.0, 1000e3, 2, save=False, packed_end=True, n_pad=n)
    fan.run()
    torch.cuda.synchronize()
    t_end = fan.records[:n, 0]
    bins, lo, hi = 4096, 1000e3 / 1560.0, 1000e3 / 1400.0
    h = arrival_time_histogram(t_end, fan.status, bins, lo, hi).cpu().numpy()
    st = fan.status.cpu().numpy()
    T = fan.records[:n, 0].cpu().numpy()
    want = np.histogram(T[(st == 0) & ~np.isnan(T)], bins=bins, range=(lo, hi))[0]
    assert np.array_equal(h, want) and h.sum() > 0.99 * n
    # the records carry the same end states, counts and status as the plain outputs of an unpacked run
    plain = DeviceFan(env, y0, 0.0, 1000e3, 2, save=False)
    plain.run()
    end, nb, ns, stg = start_all_gather_records(fan.records, n).finish()
    torch.cuda.synchronize()
    assert torch.equal(stg, plain.status) and torch.equal(nb, plain.n_bott) and torch.equal(ns, plain.n_surf)
    ok = plain.status == 0
    assert torch.equal(end[ok], plain.end[ok]) and bool(torch.isnan(end[~ok]).all())
    # a strided oracle subset pins the end states the histogram was built from
    sub = np.arange(0, n, 500)
    o = oracle.shoot_fan(*arrs, y0[sub], 0.0, 1000e3, 2, math=oracle.MATH_CR)
    okk = o["status"] == 0
    assert np.array_equal(okk, st[sub] == 0)
    assert np.all(T[sub][okk] == o["T"][okk, -1]), int(np.sum(T[sub][okk] != o["T"][okk, -1]))
    env.close()


def test_eigen_refine_device_loop_follows_the_reference_loop(lib):
    """pgr_eigen_refine runs pygenray's _find_single_eigenray (REF/eigenrays.py:206-268) for all brackets
    on the device.  The checker is the same loop written out in NumPy around host-launched fans (one
    fan per iteration): same outcome per bracket (found / trial ray dropped / iteration limit), same
    number of trial rays, the same trial angles to 1e-10 degrees (the device computes sin(radians(.)) of
    the initial slowness correctly rounded, NumPy with the platform libm: one ulp apart now and then).
    Coarse fans so that the false position takes several steps, and a fan across bounce-count
    discontinuities (false brackets that run into the iteration limit)."""
    from pygenray_amd.host_physics import bilinear_interp
    arrs = munk_arrays(200e3)
    env = lib.EnvHandle(*arrs)
    zs, rd, x1 = 1000.0, 1200.0, 200e3
    c0 = bilinear_interp(0.0, zs, arrs[2], arrs[3], arrs[0])

    def shoot(theta_user):
        y0 = np.stack([np.zeros(len(theta_user)), np.full(len(theta_user), zs), np.sin(np.radians(-theta_user)) / c0], 1)
        o = env.shoot_fan(y0, 0.0, x1, 2, save=False)
        return o["status"], -o["end"][:, 1]

    for grid in (np.linspace(-12, 12, 25), np.linspace(-19.5, 19.5, 40)):
        st, zend = shoot(grid)
        ok = st == 0
        th, ze = grid[ok], zend[ok]
        starts = np.where(np.diff(np.sign(ze + rd)))[0]
        assert len(starts) >= 3
        th1, th2, z1, z2 = th[starts].copy(), th[starts + 1].copy(), ze[starts].copy(), ze[starts + 1].copy()
        got = env.eigen_refine(th1, th2, z1, z2, rd, zs, 0.0, x1, c0, ztol=1.0, max_iter=20)
        # the reference's loop, bracket by bracket in lock step
        n = len(starts)
        state = np.zeros(n, int); ntrial = np.zeros(n, int)
        theta = th1 - (z1 + rd) * (th2 - th1) / (z2 - z1)
        it = 0
        while (state == 0).any():
            a = np.where(state == 0)[0]
            s_, z_ = shoot(theta[a])
            ntrial[a] += 1
            for q, k in enumerate(a):
                if s_[q] != 0:
                    state[k] = 2
                elif abs(z_[q] + rd) < 1.0:
                    state[k] = 1
                else:
                    if np.sign(z_[q] + rd) == np.sign(z1[k] + rd):
                        z1[k], th1[k] = z_[q], theta[k]
                    else:
                        z2[k], th2[k] = z_[q], theta[k]
                    theta[k] = th1[k] - (z1[k] + rd) * (th2[k] - th1[k]) / (z2[k] - z1[k])
                    if it > 20:
                        state[k] = 3
            it += 1
        assert np.array_equal(got["state"], state) and np.array_equal(got["n_trial"], ntrial)
        assert (state == 1).sum() >= 2
        np.testing.assert_allclose(got["theta"][state == 1], theta[state == 1], rtol=0, atol=1e-10)
        assert np.all(np.abs(got["z_end"][state == 1] + rd) < 1.0)
        assert got["launches"] == ntrial.max()
        # The trial fan gives every bracket a wave of its own while there is a wave per SIMD for each (<= 1024 brackets);
        # more brackets share waves 2, 4, ... 64 to a wave: the same brackets 150 and 3 000 times over come out the same
        # every time, bit for bit, whatever the spread.
        th1o, th2o, z1o, z2o = th[starts], th[starts + 1], ze[starts], ze[starts + 1]
        for reps in (150, 3000):
            big = env.eigen_refine(np.tile(th1o, reps), np.tile(th2o, reps), np.tile(z1o, reps), np.tile(z2o, reps), rd, zs, 0.0, x1, c0,
                                   ztol=1.0, max_iter=20)
            assert big["launches"] == got["launches"]
            for name in ("state", "n_trial", "theta", "z_end", "t_end"):
                assert np.array_equal(big[name].reshape(reps, n), np.tile(got[name], (reps, 1)), equal_nan=True), (reps, name)
    assert ntrial.max() >= 4
    # nothing to do / bad arguments
    e = env.eigen_refine(np.zeros(0), np.zeros(0), np.zeros(0), np.zeros(0), rd, zs, 0.0, x1, c0)
    assert e["launches"] == 0 and e["state"].shape == (0,)
    with pytest.raises(lib.PgrError):
        env.eigen_refine(th1, th2, z1, z2, rd, zs, 0.0, x1, -1.0)
    env.close()


def test_random_environments_are_bit_identical_to_the_oracle(lib):
    """A slice of scripts/fuzz_bitparity.py (2 x 2 200 random environments, 0 rays not bit-identical when it was last run
    in full): random depth / range / bathymetry grids, table offsets, mirrored (negative-range) frames, sloping floors,
    rtol 1e-5 ... 1e-9, terminate_backwards on and off -- 48 environments as they are and 24 more passed through the
    reference's flat-earth map first (smoothly non-uniform depth grid: the cubic-index look-up where the grid qualifies,
    the three-node / bin-table forms where it does not)."""
    from pygenray_amd.environment import eflat
    n_rays = n_odd = n_cubic = 0
    for seed, flat in [(k, False) for k in range(48)] + [(k, True) for k in range(100, 124)]:
        arrs, (src, x0, th), kw, desc = random_case(seed, n_rays=96)
        if flat:
            cin, cpin, rin, zin, depths, dr, ba = arrs
            zf = eflat(zin, 35.0)[0]
            cf = np.array([eflat(zin, 35.0, row)[1] for row in cin])
            arrs = [cf, np.gradient(cf, zf, axis=1, edge_order=1), rin, zf, eflat(depths, 35.0)[0], dr, ba]
        y0 = y0_for(oracle, arrs, src, x0, th)
        env = lib.EnvHandle(*arrs)
        n_cubic += env.query(5)
        g = env.shoot_fan(y0, kw["x0"], kw["x1"], kw["S"], rtol=kw["rtol"], exact_samples=True,
                          terminate_backwards=kw["terminate_backwards"])
        env.close()
        o = oracle.shoot_fan(*arrs, y0, kw["x0"], kw["x1"], kw["S"], rtol=kw["rtol"], math=oracle.MATH_CR,
                             terminate_backwards=kw["terminate_backwards"])
        st = assert_bit_parity(g, o, label=f"seed {seed}{' (flat earth)' if flat else ''}: {desc}")
        n_rays += st["n"]; n_odd += st["odd"]
    assert n_rays > 6000 and n_odd == 0 and n_cubic >= 8, (n_rays, n_odd, n_cubic)
