"""The reference's own physics tests (tests/test_physics.py of pygenray), re-stated against the
drop-in API running on the HIP path, plus eigenray parity against vectors captured from the
reference (it has no eigenray tests of its own).  Needs a GPU."""
import numpy as np
import pytest

import pygenray_amd as pr
from pygenray_amd import shoot_ray, shoot_rays, munk_ssp, OceanEnvironment2D, DataArray
from helpers import load

pytestmark = pytest.mark.gpu


def _env(c_2d, z, r, bathy_vals):
    ssp = DataArray(c_2d, dims=["range", "depth"], coords={"range": r, "depth": z})
    bathy = DataArray(bathy_vals, dims=["range"], coords={"range": r})
    return OceanEnvironment2D(sound_speed=ssp, bathymetry=bathy, flat_earth_transform=False)


def _const_c_env(c0=1500.0, z_max=5000.0, r_max=100e3, bathy_depth=4500.0, nz=200, nr=20):
    z = np.linspace(0.0, z_max, nz)
    r = np.linspace(0.0, r_max, nr)
    return _env(np.full((nr, nz), c0), z, r, np.full(nr, bathy_depth))


def _linear_gradient_env(c0=1500.0, g=0.05, z_max=5000.0, r_max=100e3, bathy_depth=4500.0, nz=500, nr=50):
    z = np.linspace(0.0, z_max, nz)
    r = np.linspace(0.0, r_max, nr)
    return _env(np.outer(np.ones(nr), c0 + g * z), z, r, np.full(nr, bathy_depth))


def _munk_env(r_max=100e3, nr=50, nz=600, bathy_depth=5000.0):
    z = np.linspace(0.0, 6000.0, nz)
    r = np.linspace(0.0, r_max, nr)
    return _env(np.outer(np.ones(nr), munk_ssp(z)), z, r, np.full(nr, bathy_depth))


# A. Snell invariant (REF tests/test_physics.py:72-101)
@pytest.mark.parametrize("user_angle", [-5.0, -10.0, -15.0])
def test_p_constant_along_ray(user_angle):
    ray = shoot_ray(200.0, 0.0, user_angle, 30e3, 60, _const_c_env(), rtol=1e-9, flatearth=False, debug=False)
    assert ray is not None
    abs_p = np.abs(ray.p)
    assert np.std(abs_p) / np.mean(abs_p) < 1e-5


# B. straight lines in constant c (REF tests/test_physics.py:109-170)
def test_constant_c_straight_line():
    c0, z0, R, th = 1500.0, 200.0, 20e3, 10.0
    ray = shoot_ray(z0, 0.0, -th, R, 50, _const_c_env(c0=c0, r_max=R + 1e3), rtol=1e-9, flatearth=False, debug=False)
    assert ray is not None
    t_an = R / (c0 * np.cos(np.radians(th)))
    assert abs(ray.t[-1] - t_an) / t_an < 1e-3
    z_expected = -(z0 + R * np.tan(np.radians(th)))
    assert abs(ray.z[-1] - z_expected) / abs(z_expected) < 1e-3
    np.testing.assert_allclose(ray.p, -np.sin(np.radians(th)) / c0, rtol=1e-5, atol=0)


# C. linear gradient (REF tests/test_physics.py:178-249)
def test_linear_gradient_turning_depth_and_hamiltonian():
    C0, G, ZS, TH = 1500.0, 0.05, 200.0, 20.0
    ray = shoot_ray(ZS, 0.0, -TH, 80e3, 400, _linear_gradient_env(c0=C0, g=G), rtol=1e-9, flatearth=False, debug=False)
    assert ray is not None
    cs = C0 + G * ZS
    z_turn = (cs / np.cos(np.radians(TH)) - C0) / G
    assert abs(-np.min(ray.z) - z_turn) < 50.0
    H = np.sqrt(1.0 / (C0 + G * (-ray.z)) ** 2 - (-ray.p) ** 2)
    assert np.std(H) / np.mean(H) < 1e-4


# D. Munk Hamiltonian (REF tests/test_physics.py:257-302)
@pytest.mark.parametrize("user_angle", [-5.0, -10.0, -15.0])
def test_hamiltonian_conserved_munk(user_angle):
    ray = shoot_ray(1000.0, 0.0, user_angle, 100e3, 200, _munk_env(r_max=100e3), rtol=1e-9, flatearth=False, debug=False)
    assert ray is not None
    arg = np.clip(1.0 / munk_ssp(-ray.z) ** 2 - (-ray.p) ** 2, 0.0, None)
    H = np.sqrt(arg)
    H = H[H > 1e-6 / 1500.0]
    assert np.std(H) / np.mean(H) < 1e-3


# G. backwards shooting (REF tests/test_physics.py:463-579)
def test_backwards_endpoints_and_p_constant():
    ray = shoot_ray(200.0, 30e3, -10.0, 0.0, 60, _const_c_env(), rtol=1e-9, flatearth=False, debug=False)
    assert ray is not None
    assert ray.r[0] == 30e3 and ray.r[-1] == 0.0
    assert np.std(np.abs(ray.p)) / np.mean(np.abs(ray.p)) < 1e-5


def test_backwards_matches_manually_mirrored_environment():
    z = np.linspace(0.0, 6000.0, 400)
    r = np.linspace(0.0, 100e3, 80)
    c_2d = np.array([munk_ssp(z, sofar_depth=1300 + 0.01 * ri) for ri in r])
    bathy_vals = np.linspace(4500.0, 4900.0, len(r))
    env = _env(c_2d, z, r, bathy_vals)
    env_m = _env(c_2d[::-1, :], z, r, bathy_vals[::-1])
    bwd = shoot_ray(200.0, 60e3, -15.0, 10e3, 80, env, rtol=1e-9, flatearth=False, debug=False)
    fwd = shoot_ray(200.0, 40e3, -15.0, 90e3, 80, env_m, rtol=1e-9, flatearth=False, debug=False)
    assert bwd is not None and fwd is not None
    assert (bwd.n_bottom, bwd.n_surface) == (fwd.n_bottom, fwd.n_surface)
    assert bwd.n_bottom + bwd.n_surface > 0
    np.testing.assert_allclose(bwd.z, fwd.z, rtol=1e-4, atol=1e-2)
    np.testing.assert_allclose(bwd.t, fwd.t, rtol=1e-4, atol=1e-6)
    # and against the reference's own backwards shot of this case
    g = load("g4_range_dependent.npz")
    assert (bwd.n_bottom, bwd.n_surface) == (int(g["api_bwd_nb"]), int(g["api_bwd_ns"]))
    assert bwd.launch_angle == float(g["api_bwd_launch_angle"])
    np.testing.assert_allclose(bwd.r, g["api_bwd_r"], rtol=0, atol=1e-9)
    # coarse dz = 15 m grid: the reference's own tolerance for this case is rtol 1e-4
    # (tests/test_physics.py:549-551); samples next to a bounce are extrapolated (Q5)
    np.testing.assert_allclose(bwd.t, g["api_bwd_t"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(bwd.z, g["api_bwd_z"], rtol=1e-4, atol=2e-2)


def test_shoot_rays_backwards_matches_shoot_ray():
    env = _munk_env(r_max=50e3)
    angles = np.linspace(-15.0, 15.0, 80)  # the >= 70 branch of the reference
    rf = shoot_rays(200.0, 40e3, angles, 5e3, 60, env, rtol=1e-9, flatearth=False, debug=False)
    assert len(rf) == len(angles)
    assert np.allclose(rf.rs[:, 0], 40e3) and np.allclose(rf.rs[:, -1], 5e3)
    idx = np.argmin(np.abs(rf.thetas - 7.0))
    single = shoot_ray(200.0, 40e3, rf.thetas[idx], 5e3, 60, env, rtol=1e-9, flatearth=False, debug=False)
    np.testing.assert_allclose(rf.zs[idx], single.z, atol=1e-6)


def test_sign_conventions_q1_q2_q3():
    """shoot_ray(user) integrates ODE angle -user and stores launch_angle = -user (Q2); a fan
    of < 70 angles integrates +user, a fan of >= 70 integrates -user (Q1); z, p are stored
    negated (Q3).  Checked against the reference's shoot_ray output for three angles."""
    z = np.arange(0, 6000, 1.0)
    r = np.linspace(0, 100e3, 100)
    env = _env(np.tile(munk_ssp(z), (100, 1)), z, r, np.full(100, 5000.0))
    g = load("g2_shoot_ray_api.npz")
    import oracle
    from helpers import y0_for, XI_MAX
    arrs = pr._unpack_envi(env, flatearth=False)
    for k, a in enumerate(g["user_angles"]):
        ray = shoot_ray(1000.0, 0.0, float(a), 100e3, 101, env, debug=False, flatearth=False)
        assert ray.launch_angle == g["launch_angle"][k] == -a
        assert (ray.n_bottom, ray.n_surface) == (g["n_bottom"][k], g["n_surface"][k])
        # samples the reference extrapolates far outside a step (Q5) are ill-conditioned
        xi = oracle.shoot_fan(*arrs, y0_for(oracle, arrs, 1000.0, 0.0, [-a]), 0.0, 100e3, 101)["xi"][0]
        good = np.abs(xi) <= XI_MAX
        np.testing.assert_allclose(ray.t[good], g["t"][k][good], rtol=0, atol=2e-7)
        np.testing.assert_allclose(ray.z[good], g["z"][k][good], rtol=0, atol=2e-3)
        assert np.all(np.isfinite(ray.z)) and ray.z[0] == -1000.0
    small = shoot_rays(1000.0, 0.0, [3.0], 100e3, 11, env, debug=False, flatearth=False)
    big = shoot_rays(1000.0, 0.0, np.linspace(3.0, 3.0, 70), 100e3, 11, env, debug=False, flatearth=False)
    assert small.thetas[0] == 3.0 and big.thetas[0] == 3.0
    assert small.zs[0, 1] < -1000.0 < big.zs[0, 1]  # +user goes down in the small fan, up in the big one


def test_debug_messages_and_dropped_rays(capsys):
    z = np.linspace(0, 3000, 301)
    r = np.linspace(0, 50e3, 20)
    env = _env(np.tile(munk_ssp(z), (20, 1)), z, r, np.full(20, 5000.0))
    assert shoot_ray(500.0, 0.0, -14.0, 50e3, 20, env, flatearth=False, debug=True) is None
    assert "bounding box" in capsys.readouterr().out
    fan = shoot_rays(500.0, 0.0, [-14.0, -2.0], 50e3, 20, env, flatearth=False, debug=False)
    assert len(fan) == 1  # dropped rays vanish (Q12)


def test_flat_earth_default_environment():
    env = OceanEnvironment2D()  # default Munk + sloping default bathymetry, flat earth on
    fan = shoot_rays(1000.0, 0.0, np.linspace(-10, 10, 21), 90e3, 91, env, debug=False)
    assert len(fan) == 21 and np.all(np.isfinite(fan.zs))
    with pytest.raises(Exception, match="Flat earth"):
        shoot_ray(1000.0, 0.0, 1.0, 90e3, 10, OceanEnvironment2D(flat_earth_transform=False), debug=False)


# Eigenrays: parity against the reference's _find_single_eigenray (captured in g6)
def test_find_eigenrays_matches_reference():
    g = load("g6_eigenrays.npz")
    z = np.arange(0, 6000, 1.0)
    r = np.linspace(0, 100e3, 100)
    env = _env(np.tile(munk_ssp(z), (100, 1)), z, r, np.full(100, 5000.0))
    fan = shoot_rays(1000.0, 0.0, g["fan_angles_user"], 100e3, 21, env, debug=False, flatearth=False)
    np.testing.assert_array_equal(fan.thetas, g["fan_thetas"])
    np.testing.assert_allclose(fan.zs[:, -1], g["fan_z_end"], atol=1e-4)
    np.testing.assert_allclose(fan.ts[:, -1], g["fan_t_end"], atol=1e-8)
    er = pr.find_eigenrays(fan, [float(g["receiver_depth"])], 1000.0, 0.0, 100e3, 21, env, ztol=1,
                           max_iter=20, debug=False, flatearth=False)
    ref = g["eigen"]
    assert er.num_eigenrays[float(g["receiver_depth"])] == len(g["bracket_starts"])
    assert er.num_eigenrays_found[0] == int(np.sum(~np.isnan(ref[:, 0])))
    np.testing.assert_allclose(er.launch_angles[0], ref[:, 0], rtol=0, atol=1e-6)
    np.testing.assert_allclose(er.ts[0][:, -1], ref[:, 1], rtol=0, atol=1e-6)
    np.testing.assert_allclose(er.zs[0][:, -1], ref[:, 2], rtol=0, atol=1e-3)
    assert np.all(np.abs(er.zs[0][:, -1] + 1000.0) < 1.0)
    assert er.rs[0].shape == (4, 21) and er.received_angles[0].shape == (4,)
    assert er.failed_eray_theta_brackets[0] == []


def test_find_eigenrays_several_receiver_depths_and_structure():
    z = np.arange(0, 6000, 1.0)
    r = np.linspace(0, 100e3, 100)
    env = _env(np.tile(munk_ssp(z), (100, 1)), z, r, np.full(100, 5000.0))
    fan = shoot_rays(1000.0, 0.0, np.linspace(-14, 14, 141), 100e3, 31, env, debug=False, flatearth=False)
    rds = [500.0, 1000.0, 3000.0]
    er = pr.find_eigenrays(fan, rds, 1000.0, 0.0, 100e3, 31, env, ztol=1, max_iter=20, debug=False,
                           flatearth=False)
    for k, rd in enumerate(rds):
        n_br = er.num_eigenrays[rd]
        # brackets = sign changes of (z_end + receiver_depth) between neighbouring fan rays
        assert n_br == int(np.count_nonzero(np.diff(np.sign(fan.zs[:, -1] + rd))))
        assert er.num_eigenrays_found[k] + len(er.failed_eray_theta_brackets[k]) == n_br
        assert er.zs[k].shape == (er.num_eigenrays_found[k], 31) == er.ts[k].shape
        assert np.all(np.abs(er.zs[k][:, -1] + rd) < 1.0)            # within ztol of the receiver
        assert np.all(np.diff(er.launch_angles[k]) > 0)              # bracket order = launch-angle order
        assert er.received_angles[k].shape == er.launch_angles[k].shape == er.ray_id[k].shape
        # every eigenray's launch angle lies inside a fan bracket
        idx = np.searchsorted(fan.thetas, er.launch_angles[k])
        assert np.all((idx > 0) & (idx < len(fan.thetas)))
    assert er.num_eigenrays_found[1] >= 3


def test_environment_cache_follows_replaced_tables():
    z = np.arange(0, 6000, 2.0)
    r = np.linspace(0, 50e3, 20)
    env = _env(np.tile(munk_ssp(z), (20, 1)), z, r, np.full(20, 5000.0))
    a = shoot_ray(1000.0, 0.0, 4.0, 50e3, 11, env, debug=False, flatearth=False)
    env.sound_speed = DataArray(np.tile(munk_ssp(z, sofar_depth=1000.0), (20, 1)), dims=["range", "depth"],
                                coords={"range": r, "depth": z})
    b = shoot_ray(1000.0, 0.0, 4.0, 50e3, 11, env, debug=False, flatearth=False)
    assert abs(a.z[-1] - b.z[-1]) > 1.0
