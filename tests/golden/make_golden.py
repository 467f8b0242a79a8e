#!/usr/bin/env python3
"""Generate golden vectors by RUNNING the reference (pygenray) in the build container.

TEST INFRASTRUCTURE ONLY.  This script is the only place that imports
/root/reference; it is run by hand in the build container (the reference does
not exist on the GPU box) and only its *outputs* (small .npz files of inputs
and expected outputs) are committed next to it.

The reference needs `numba` and `xarray`, neither of which is installed here
(SURVEY.md section 8c).  As recorded there, the import works with in-memory
stand-ins: `numba.njit` -> identity decorator (the reference then runs as
plain, strictly-IEEE Python - no fastmath re-association) and an empty
`xarray` module (only used for a type annotation).  The hot path is driven
through the reference's own functions:

  * pygenray.launch_rays._shoot_ray_array / _interpolate_ray   (array level)
  * pygenray.launch_rays.shoot_ray / shoot_rays (<70 branch)    via a duck-typed
    environment object exposing exactly the attributes _unpack_envi reads
    (reference launch_rays.py:717-742)
  * pygenray.eigenrays._find_single_eigenray                    (same duck env)
  * pygenray.integration_processes.*                            (unit vectors)

Usage:  python tests/golden/make_golden.py            (writes tests/golden/*.npz)
"""
import os
import sys
import types
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SRC = "/root/reference/src"


def _install_standins():
    sys.dont_write_bytecode = True
    nb = types.ModuleType("numba")

    def njit(*a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return lambda f: f

    nb.njit = njit
    sys.modules["numba"] = nb
    xr = types.ModuleType("xarray")

    class DataArray:  # annotation only (environment.py:240)
        pass

    xr.DataArray = DataArray
    sys.modules["xarray"] = xr
    sys.path.insert(0, REF_SRC)


_install_standins()
import pygenray as pr  # noqa: E402
from pygenray import launch_rays as lr  # noqa: E402
from pygenray import eigenrays as er  # noqa: E402
from pygenray.environment import munk_ssp, eflat  # noqa: E402


# --------------------------------------------------------------------------
# duck-typed environment: just what _unpack_envi / EigenRays touch
# --------------------------------------------------------------------------
class _Coord:
    def __init__(self, v):
        self.values = np.asarray(v, dtype=float)


class _DA:
    """The handful of xarray.DataArray members the reference reads."""

    def __init__(self, values, **coords):
        self.values = np.asarray(values, dtype=float)
        self._coords = {k: np.asarray(v, dtype=float) for k, v in coords.items()}
        for k, v in self._coords.items():
            setattr(self, k, _Coord(v))

    def differentiate(self, coord):
        # xarray.DataArray.differentiate(coord, edge_order=1) == np.gradient
        assert coord == "depth"
        return _DA(np.gradient(self.values, self._coords["depth"], axis=1, edge_order=1),
                   **self._coords)


class DuckEnv:
    def __init__(self, c2d, r, z, bathy, bathy_r, lat=35.0, flat=False):
        self.sound_speed = _DA(c2d, range=r, depth=z)
        self.bathymetry = _DA(bathy, range=bathy_r)
        # environment.py:111-114 (computed from the UNtransformed bathymetry, Q10)
        slope = np.gradient(np.asarray(bathy, float), np.asarray(bathy_r, float))
        self.bottom_angle = np.degrees(np.arctan(slope))
        if flat:
            # environment.py:121-154 (same depth grid for every column)
            depf, _ = eflat(np.asarray(z, float), lat, np.asarray(c2d, float)[0])
            cf = np.array([eflat(np.asarray(z, float), lat, row)[1] for row in np.asarray(c2d, float)])
            self.sound_speed_fe = _DA(cf, range=r, depth=depf)
            bf, _ = eflat(np.asarray(bathy, float), lat)
            self.bathymetry_fe = _DA(bf, range=bathy_r)


def env_arrays(env, flatearth=False):
    return lr._unpack_envi(env, flatearth=flatearth)


def shoot_array_level(arrs, source_depth, source_range, receiver_range, theta_ode_deg, S,
                      rtol=1e-9, terminate_backwards=True):
    """Exactly what _shoot_single_ray_process does (launch_rays.py:544-576) minus shm."""
    cin, cpin, rin, zin, depths, depth_ranges, bottom_angles = arrs
    c = pr.bilinear_interp(source_range, source_depth, rin, zin, cin)
    N = len(theta_ode_deg)
    r = np.linspace(source_range, receiver_range, S)
    T = np.full((N, S), np.nan)
    Z = np.full((N, S), np.nan)
    P = np.full((N, S), np.nan)
    nb = np.zeros(N, np.int64)
    ns = np.zeros(N, np.int64)
    ok = np.zeros(N, np.int64)
    nsteps = np.zeros(N, np.int64)
    nfev = np.zeros(N, np.int64)
    nseg = np.zeros(N, np.int64)
    y0s = np.zeros((N, 3))
    for k, th in enumerate(theta_ode_deg):
        y0 = np.array([0, source_depth, np.sin(np.radians(th)) / c])
        y0s[k] = y0
        sols, full_ray, n_b, n_s = lr._shoot_ray_array(
            y0.copy(), source_depth, source_range, receiver_range, cin, cpin, rin, zin,
            depths, depth_ranges, bottom_angles, rtol, terminate_backwards, False)
        if full_ray is None:
            continue
        out = lr._interpolate_ray(sols, r)
        T[k], Z[k], P[k] = out[1], out[2], out[3]
        nb[k], ns[k], ok[k] = n_b, n_s, 1
        nsteps[k] = sum(len(s.t) - 1 for s in sols)
        nfev[k] = sum(s.nfev for s in sols)
        nseg[k] = len(sols)
    return dict(r=r, T=T, z=Z, p=P, n_bott=nb, n_surf=ns, ok=ok, n_steps=nsteps, nfev=nfev,
                n_seg=nseg, y0=y0s)


def pack_env(arrs, prefix="env_"):
    names = ["cin", "cpin", "rin", "zin", "depths", "depth_ranges", "bottom_angles"]
    return {prefix + n: np.asarray(a) for n, a in zip(names, arrs)}


def save(name, **kw):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **kw)
    print(f"  wrote {name}: {os.path.getsize(path) / 1024:.1f} KiB")


# --------------------------------------------------------------------------
def munk_env(r_max, nr, z, bathy_depth=5000.0, flat=False):
    r = np.linspace(0.0, r_max, nr)
    c2d = np.outer(np.ones(nr), munk_ssp(z))
    return DuckEnv(c2d, r, z, np.full(nr, bathy_depth), r, flat=flat)


def g1_fixture_case():
    """The reference's own regression case (tests/test_physics.py:310-386):
    nz=400, nr=30, 50 km, S=50, angles [-8..8] through shoot_rays' <70 branch."""
    env = munk_env(50e3, 30, np.linspace(0.0, 6000.0, 400))
    angles = [-8.0, -4.0, 0.0, 4.0, 8.0]
    rf = pr.shoot_rays(1300.0, 0.0, angles, 50e3, 50, env, n_processes=1, debug=False,
                       flatearth=False)
    arrs = env_arrays(env)
    # +-1 ulp self-noise of the reference on this coarse grid
    c0 = pr.bilinear_interp(0.0, 1300.0, arrs[2], arrs[3], arrs[0])
    noise_z = np.zeros(len(angles))
    noise_t = np.zeros(len(angles))
    for k, a in enumerate(rf.thetas):
        p0 = np.sin(np.radians(a)) / c0
        outs = []
        for p in (np.nextafter(p0, -1), p0, np.nextafter(p0, 1)):
            sols, fr, _, _ = lr._shoot_ray_array(np.array([0, 1300.0, p]), 1300.0, 0.0, 50e3,
                                                 *arrs, 1e-9, True, False)
            outs.append(sols[-1].y[:, -1].copy())
        outs = np.array(outs)
        noise_t[k] = np.ptp(outs[:, 0])
        noise_z[k] = np.ptp(outs[:, 1])
    save("g1_fixture_case.npz", angles_user=np.array(angles), thetas=rf.thetas, rs=rf.rs,
         ts=rf.ts, zs=rf.zs, ps=rf.ps, n_botts=rf.n_botts, n_surfs=rf.n_surfs,
         source_depth=1300.0, source_range=0.0, receiver_range=50e3, S=50,
         selfnoise_t=noise_t, selfnoise_z=noise_z, **pack_env(arrs))


def g2_munk_100km():
    """BASELINE config[0]-shaped: dz=1 m Munk, 64 angles, 100 km (S=101 to keep it small).
    Both sign conventions of Q1 are covered by passing ODE angles +-."""
    z = np.arange(0, 6000, 1.0)
    env = munk_env(100e3, 100, z)
    arrs = env_arrays(env)
    theta = np.linspace(-20, 20, 64)
    out = shoot_array_level(arrs, 1000.0, 0.0, 100e3, theta, 101)
    # the table is range independent: store one row + nr (rebuilt by the test)
    save("g2_munk_100km.npz", theta_ode=theta, source_depth=1000.0, source_range=0.0,
         receiver_range=100e3, c_row=arrs[0][0], cp_row=arrs[1][0], rin=arrs[2], zin=arrs[3],
         depths=arrs[4], depth_ranges=arrs[5], bottom_angles=arrs[6], **out)
    # also the public API path for 3 of them (reference shoot_ray sign convention, Q2/Q3)
    rays = [pr.shoot_ray(1000.0, 0.0, a, 100e3, 101, env, debug=False, flatearth=False)
            for a in (-12.5, 3.0, 17.0)]
    save("g2_shoot_ray_api.npz", user_angles=np.array([-12.5, 3.0, 17.0]),
         launch_angle=np.array([r.launch_angle for r in rays]),
         r=np.array([r.r for r in rays]), t=np.array([r.t for r in rays]),
         z=np.array([r.z for r in rays]), p=np.array([r.p for r in rays]),
         n_bottom=np.array([r.n_bottom for r in rays]),
         n_surface=np.array([r.n_surface for r in rays]))


def g3_munk_1000km():
    """BASELINE config[1]-shaped subset: 1000 km, 16 angles incl. +-16, +-20 (many bounces),
    S=101, with step counts and +-1 ulp self-noise of the reference end state."""
    z = np.arange(0, 6000, 1.0)
    env = munk_env(1000e3, 100, z)
    arrs = env_arrays(env)
    theta = np.array([-20.0, -16.0, -13.0, -11.0, -8.0, -5.0, -2.5, -0.3, 0.0, 1.7, 4.0, 7.5,
                      10.0, 12.7, 16.0, 19.3])
    out = shoot_array_level(arrs, 1000.0, 0.0, 1000e3, theta, 101)
    c0 = pr.bilinear_interp(0.0, 1000.0, arrs[2], arrs[3], arrs[0])
    noise = np.zeros((len(theta), 3))
    for k, th in enumerate(theta):
        p0 = np.sin(np.radians(th)) / c0
        ends = []
        for p in (np.nextafter(p0, -1), np.nextafter(p0, 1)):
            sols, fr, _, _ = lr._shoot_ray_array(np.array([0, 1000.0, p]), 1000.0, 0.0, 1000e3,
                                                 *arrs, 1e-9, True, False)
            ends.append(sols[-1].y[:, -1].copy() if fr is not None else np.full(3, np.nan))
        ends.append(np.array([out["T"][k, -1], out["z"][k, -1], out["p"][k, -1]]))
        noise[k] = np.ptp(np.array(ends), axis=0)
    save("g3_munk_1000km.npz", theta_ode=theta, source_depth=1000.0, source_range=0.0,
         receiver_range=1000e3, c_row=arrs[0][0], cp_row=arrs[1][0], rin=arrs[2], zin=arrs[3],
         depths=arrs[4], depth_ranges=arrs[5], bottom_angles=arrs[6], selfnoise_end=noise, **out)


def g4_range_dependent():
    """Range-dependent c(r,z) + sloping bottom of tests/test_physics.py:494-506, forward,
    backward (mirrored, launch_rays.py:684-714) and through shoot_ray; plus a BASELINE
    config[2]-shaped (dz=1 m, sofar depth drifting 2e-4*r, 1000 km) subset."""
    z = np.linspace(0.0, 6000.0, 400)
    r = np.linspace(0.0, 100e3, 80)
    c2d = np.array([munk_ssp(z, sofar_depth=1300 + 0.01 * ri) for ri in r])
    bathy = np.linspace(4500.0, 4900.0, len(r))
    env = DuckEnv(c2d, r, z, bathy, r)
    arrs = env_arrays(env)
    theta = np.array([-15.0, -9.0, -3.0, 2.0, 8.0, 14.0, 18.0])
    fwd = shoot_array_level(arrs, 200.0, 10e3, 90e3, theta, 81)
    cin_m, cpin_m, rin_m, depths_m, dr_m, ba_m = lr._mirror_envi_arrays(
        arrs[0], arrs[1], arrs[2], arrs[4], arrs[5], arrs[6])
    arrs_m = (cin_m, cpin_m, rin_m, arrs[3], depths_m, dr_m, ba_m)
    bwd = shoot_array_level(arrs_m, 200.0, -60e3, -10e3, theta, 80)
    rb = pr.shoot_ray(200.0, 60e3, -15.0, 10e3, 80, env, rtol=1e-9, flatearth=False, debug=False)
    save("g4_range_dependent.npz", theta_ode=theta,
         **pack_env(arrs), **{"fwd_" + k: v for k, v in fwd.items()},
         **{"bwd_" + k: v for k, v in bwd.items()},
         api_bwd_r=rb.r, api_bwd_t=rb.t, api_bwd_z=rb.z, api_bwd_p=rb.p,
         api_bwd_nb=rb.n_bottom, api_bwd_ns=rb.n_surface, api_bwd_launch_angle=rb.launch_angle)

    # config[2]-shaped: big table is rebuilt by the test from the formula below
    z1 = np.arange(0, 6000, 1.0)
    r1 = np.linspace(0.0, 1000e3, 101)
    c2 = np.array([munk_ssp(z1, sofar_depth=1300 + 2e-4 * ri) for ri in r1])
    env2 = DuckEnv(c2, r1, z1, np.full(101, 5000.0), r1)
    arrs2 = env_arrays(env2)
    theta2 = np.array([-19.0, -14.0, -9.0, -4.0, 0.5, 6.0, 11.0, 15.0])
    out2 = shoot_array_level(arrs2, 1000.0, 0.0, 1000e3, theta2, 101)
    save("g4_config2_subset.npz", theta_ode=theta2, source_depth=1000.0,
         sofar_slope=2e-4, nr=101, r_max=1000e3, cp_checksum=np.sum(arrs2[1]),
         c_checksum=np.sum(arrs2[0]), **out2)


def g5_analytic_envs():
    """Constant-c and linear-gradient environments of tests/test_physics.py:25-51 with
    surface+bottom bounces, steep (terminating) rays, and flat-earth tables."""
    # constant c: 30 km, bounces
    zc = np.linspace(0.0, 5000.0, 200)
    rc = np.linspace(0.0, 100e3, 20)
    envc = DuckEnv(np.full((20, 200), 1500.0), rc, zc, np.full(20, 4500.0), rc)
    arrc = env_arrays(envc)
    th = np.array([-15.0, -10.0, -5.0, 5.0, 10.0, 15.0, 40.0, 60.0, 80.0])
    outc = shoot_array_level(arrc, 200.0, 0.0, 30e3, th, 60)
    save("g5_const_c.npz", theta_ode=th, **pack_env(arrc), **outc)
    # near-vertical rays over a short range (tests/test_physics.py:394-455 use rtol=1e-6);
    # an initially vertical ray never fires vertical_ray (event starts at +1, Q6)
    thv = np.array([85.0, 89.0, -89.0])
    outv = shoot_array_level(arrc, 200.0, 0.0, 1.5e3, thv, 31, rtol=1e-6)
    save("g5_const_c_steep.npz", theta_ode=thv, rtol=1e-6, **pack_env(arrc), **outv)
    # linear gradient
    zl = np.linspace(0.0, 5000.0, 500)
    rl = np.linspace(0.0, 100e3, 50)
    envl = DuckEnv(np.outer(np.ones(50), 1500.0 + 0.05 * zl), rl, zl, np.full(50, 4500.0), rl)
    arrl = env_arrays(envl)
    thl = np.array([-20.0, -12.0, 3.0, 20.0])
    outl = shoot_array_level(arrl, 200.0, 0.0, 80e3, thl, 400)
    save("g5_linear_gradient.npz", theta_ode=thl, **pack_env(arrl),
         **{k: (v[:, ::4] if v.ndim == 2 and v.shape[1] == 400 else v) for k, v in outl.items()
            if k != "r"}, r=outl["r"][::4])
    # flat-earth tables (non-uniform depth grid) + default sloping bathymetry (Q11), 100 km
    zf = np.arange(0, 6000, 4.0)
    rf = np.linspace(0.0, 100e3, 100)
    envf = DuckEnv(np.outer(np.ones(100), munk_ssp(zf)), rf, zf, np.linspace(4500, 4900, 100), rf,
                   lat=35.0, flat=True)
    arrf = env_arrays(envf, flatearth=True)
    thf = np.array([-14.0, -6.0, 0.0, 5.0, 12.0])
    outf = shoot_array_level(arrf, 800.0, 0.0, 100e3, thf, 101)
    save("g5_flatearth.npz", theta_ode=thf, **pack_env(arrf), **outf)


def g6_eigenrays():
    """Eigenray search (the reference has no tests for it): fan end depths, brackets, and the
    per-bracket result of the reference's _find_single_eigenray (eigenrays.py:206-268).

    The search is only self-consistent for fans from shoot_rays' >=70-ray branch (Q1: there
    RayFan.thetas = user angle and the ODE angle is -user, the same as shoot_ray).  That
    branch runs a spawn pool (workers cannot see the stand-in modules), so the fan is
    assembled here exactly as launch_rays.py:140-186 does it, from the same per-ray call
    (_shoot_ray_array + _interpolate_ray + pr.Ray) the pool workers make."""
    z = np.arange(0, 6000, 1.0)
    env = munk_env(100e3, 100, z)
    arrs = env_arrays(env)
    angles = np.linspace(-12, 12, 80)
    la = -angles                                                   # launch_rays.py:67
    out = shoot_array_level(arrs, 1000.0, 0.0, 100e3, la, 21)
    rays = []
    for k in range(len(angles)):
        ray = pr.Ray(out["r"], np.stack([out["T"][k], out["z"][k], out["p"][k]]),
                     out["n_bott"][k], out["n_surf"][k], source_depth=1000.0)
        ray.launch_angle = -la[k]                                  # launch_rays.py:180
        rays.append(ray)
    fan = pr.RayFan(rays)
    rd = 1000.0
    depth_sign = np.sign(fan.zs[:, -1] + rd)                       # eigenrays.py:65-69
    starts = np.where(np.diff(depth_sign))[0]
    res = []
    for k, s in enumerate(starts):
        z1, z2 = fan.zs[s, -1], fan.zs[s + 1, -1]
        t1, t2 = fan.thetas[s], fan.thetas[s + 1]
        rft = t1 - (z1 + rd) * (t2 - t1) / (z2 - z1)               # eigenrays.py:118-120
        ray = er._find_single_eigenray((k, z1, z2, t1, t2, rft, rd, 1000.0, 0.0, 100e3, 21, env,
                                        1, 20, dict(debug=False, flatearth=False)))
        if ray is None:
            res.append([np.nan] * 6)
        else:
            res.append([ray.launch_angle, ray.t[-1], ray.z[-1], ray.p[-1], ray.n_bottom,
                        ray.n_surface])
    save("g6_eigenrays.npz", fan_angles_user=angles, fan_thetas=fan.thetas, fan_z_end=fan.zs[:, -1],
         fan_t_end=fan.ts[:, -1], receiver_depth=rd, bracket_starts=starts,
         eigen=np.array(res, dtype=float), source_depth=1000.0, receiver_range=100e3, S=21)


def g7_unit_vectors():
    """Unit vectors for a1-a8 (integration_processes.py): random points inside and outside
    the grid (Q4 extrapolation), |p c|>1 -> NaN angle (Q7), clamp (Q8)."""
    rng = np.random.default_rng(0)
    z = np.linspace(0.0, 6000.0, 61) ** 1.0
    z[1:-1] += rng.uniform(-20, 20, 59)  # non-uniform depth grid
    r = np.sort(rng.uniform(0, 100e3, 12))
    r[0], r[-1] = 0.0, 100e3
    cin = np.array([munk_ssp(z, sofar_depth=1300 + 0.002 * ri) for ri in r])
    cpin = np.gradient(cin, z, axis=1, edge_order=1)
    dr = np.linspace(0, 100e3, 7)
    depths = np.array([4500.0, 4700.0, 4300.0, 4800.0, 4900.0, 4650.0, 4500.0])
    M = 400
    xs = rng.uniform(-5e3, 105e3, M)
    zs = rng.uniform(-200, 6200, M)
    # exact-node queries (searchsorted side='left' semantics)
    xs[:12] = r
    zs[12:73] = z
    ps = rng.uniform(-7.2e-4, 7.2e-4, M)
    ys = np.stack([rng.uniform(0, 100, M), zs, ps], 1)
    bil = np.array([pr.bilinear_interp(x, zz, r, z, cin) for x, zz in zip(xs, zs)])
    lin = np.array([pr.linear_interp(x, dr, depths) for x in xs])
    der = np.array([pr.derivsrd(x, y, cin, cpin, r, z, depths, dr) for x, y in zip(xs, ys)])
    with np.errstate(invalid="ignore"):
        ang = np.array([pr.ray_angle(x, y, cin, r, z) for x, y in zip(xs, ys)])
        ev = np.array([[f(x, y, cin, cpin, r, z, depths, dr) for f in
                        (pr.surface_bounce, pr.bottom_bounce, pr.vertical_ray,
                         pr.ray_bounding_box_event)] for x, y in zip(xs, ys)])
    save("g7_unit_vectors.npz", cin=cin, cpin=cpin, rin=r, zin=z, depths=depths, depth_ranges=dr,
         xs=xs, ys=ys, bilinear=bil, linear=lin, derivs=der, angle=ang, events=ev)


def g8_timing():
    """Reference (un-jitted) CPU cost on config[0] for BASELINE bookkeeping."""
    z = np.arange(0, 6000, 1.0)
    env = munk_env(100e3, 100, z)
    arrs = env_arrays(env)
    theta = np.linspace(-20, 20, 64)
    t0 = time.time()
    out = shoot_array_level(arrs, 1000.0, 0.0, 100e3, theta, 1001)
    dt = time.time() - t0
    print(f"  config0 reference (un-jitted, 1 core): {dt:.2f} s, "
          f"{out['n_steps'].sum()} ray-steps -> {out['n_steps'].sum() / dt:.0f} ray-steps/s")


def g9_irregular_grids():
    """Grids the table look-up has to SEARCH: a randomly spaced range grid (cells of 50 m ... 3 km,
    so that loose-tolerance steps span several cells), a randomly spaced bathymetry with slopes and
    a stretched depth grid; range-dependent sound speed.  rtol 1e-9 and 1e-5."""
    rng = np.random.default_rng(9)
    z = 5500.0 * np.linspace(0, 1, 900) ** 1.25
    r = np.sort(np.concatenate([[0.0, 70e3], rng.uniform(0, 70e3, 120)]))
    c2d = np.array([munk_ssp(z, sofar_depth=1200 + 3e-3 * ri) for ri in r])
    br = np.sort(np.concatenate([[0.0, 70e3], rng.uniform(0, 70e3, 17)]))
    bathy = 4600.0 + 250.0 * np.sin(br / 11e3)
    env = DuckEnv(c2d, r, z, bathy, br)
    arrs = env_arrays(env)
    theta = np.array([-17.0, -11.0, -6.0, -1.5, 3.0, 7.5, 12.0, 16.0])
    out = {}
    for tag, rtol in (("t9_", 1e-9), ("t5_", 1e-5)):
        o = shoot_array_level(arrs, 600.0, 1e3, 69e3, theta, 61, rtol=rtol)
        out.update({tag + k: v for k, v in o.items()})
    save("g9_irregular_grids.npz", theta_ode=theta, **pack_env(arrs), **out)


def g10_eigenrays_1000km():
    """BASELINE configs[3] at reference speed: the config-1 environment to 1000 km, launch angles on
    the 1e6-angle grid linspace(-20, 20, 1_000_000).  The reference cannot shoot 1e6 rays (0.4 s
    each), so: the C oracle (test infrastructure, oracle/) scans every 500th grid angle for sign
    changes of z_end + receiver_depth and bisects on the grid index down to the adjacent pair; the
    REFERENCE then shoots windows of 48 consecutive grid angles around three of those brackets (a
    near-axial one, a refracted one, a surface/bottom-reflected one) and runs its own
    _find_single_eigenray (eigenrays.py:206-268) on each, with every trial angle recorded."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    import oracle
    z = np.arange(0, 6000, 1.0)
    env = munk_env(1000e3, 100, z)
    arrs = env_arrays(env)
    N = 1_000_000
    grid = np.linspace(-20, 20, N)
    rd, zs, x1 = 1000.0, 1000.0, 1000e3
    c0 = pr.bilinear_interp(0.0, zs, arrs[2], arrs[3], arrs[0])

    def zend(idx):
        th = -grid[np.asarray(idx)]                     # ODE angle = -user (>= 70-ray branch)
        y0 = np.stack([np.zeros(len(th)), np.full(len(th), zs), np.sin(np.radians(th)) / c0], 1)
        o = oracle.shoot_fan(*arrs, y0, 0.0, x1, 2)
        return np.where(o["status"] == 0, -o["z"][:, -1], np.nan)   # stored convention

    coarse = np.arange(0, N, 500)
    zc = zend(coarse)
    sg = np.sign(zc + rd)
    cand = np.where((np.diff(sg) != 0) & np.isfinite(zc[:-1]) & np.isfinite(zc[1:]))[0]
    print(f"  {len(cand)} sign changes on the 500-stride scan")
    picks = []
    for want in (0.3, 8.0, 15.5):                        # |angle| of the brackets to keep
        k = cand[np.argmin(np.abs(np.abs(grid[coarse[cand]]) - want))]
        lo, hi = int(coarse[k]), int(coarse[k + 1])
        slo = np.sign(zend([lo])[0] + rd)
        while hi - lo > 1:                               # bisect on the grid index
            mid = (lo + hi) // 2
            if np.sign(zend([mid])[0] + rd) == slo:
                lo = mid
            else:
                hi = mid
        picks.append(lo)
    print("  brackets start at grid indices", picks, "angles", grid[picks])
    W = 48
    out = dict(n_grid=N, receiver_depth=rd, source_depth=zs, receiver_range=x1, window=W)
    log = []
    orig = pr.shoot_ray

    def logged(source_depth, source_range, launch_angle, *a, **k):
        log.append(float(launch_angle))
        return orig(source_depth, source_range, launch_angle, *a, **k)

    for j, s0 in enumerate(picks):
        idx = np.arange(s0 - W // 2 + 1, s0 + W // 2 + 1)
        la = -grid[idx]
        t0 = time.time()
        o = shoot_array_level(arrs, zs, 0.0, x1, la, 2)
        print(f"  window {j}: {len(idx)} reference rays in {time.time() - t0:.0f} s")
        zs_end, ts_end = -o["z"][:, -1], o["T"][:, -1]
        depth_sign = np.sign(zs_end + rd)
        starts = np.where(np.diff(depth_sign))[0]
        assert len(starts) >= 1 and (s0 - idx[0]) in starts, (starts, s0 - idx[0])
        b = s0 - idx[0]
        z1, z2, t1, t2 = zs_end[b], zs_end[b + 1], grid[idx[b]], grid[idx[b + 1]]
        rft = t1 - (z1 + rd) * (t2 - t1) / (z2 - z1)
        del log[:]
        pr.shoot_ray = logged
        try:
            ray = er._find_single_eigenray((0, z1, z2, t1, t2, rft, rd, zs, 0.0, x1, 2, env, 1, 20,
                                            dict(debug=False, flatearth=False)))
        finally:
            pr.shoot_ray = orig
        res = [np.nan] * 6 if ray is None else [ray.launch_angle, ray.t[-1], ray.z[-1], ray.p[-1], ray.n_bottom, ray.n_surface]
        out.update({f"w{j}_idx": idx, f"w{j}_z_end": zs_end, f"w{j}_t_end": ts_end, f"w{j}_ok": o["ok"],
                    f"w{j}_n_bott": o["n_bott"], f"w{j}_n_surf": o["n_surf"], f"w{j}_bracket": b,
                    f"w{j}_starts": starts, f"w{j}_theta_seq": np.array(log), f"w{j}_eigen": np.array(res, float)})
        print(f"    bracket at window position {b}; trial angles {len(log)}; eigenray {res[:3]}")
        # the same root from a COARSE bracket (a fan 500 times coarser: 0.02 degrees between the ends), so
        # that the false-position iteration takes several trial rays: the whole theta sequence is recorded
        ic = np.array([s0 - 250, s0 + 251])
        oc = shoot_array_level(arrs, zs, 0.0, x1, -grid[ic], 2)
        zc1, zc2 = -oc["z"][0, -1], -oc["z"][1, -1]
        if np.sign(zc1 + rd) != np.sign(zc2 + rd):
            tc1, tc2 = grid[ic[0]], grid[ic[1]]
            rftc = tc1 - (zc1 + rd) * (tc2 - tc1) / (zc2 - zc1)
            del log[:]
            pr.shoot_ray = logged
            try:
                rayc = er._find_single_eigenray((0, zc1, zc2, tc1, tc2, rftc, rd, zs, 0.0, x1, 2, env, 1, 20,
                                                 dict(debug=False, flatearth=False)))
            finally:
                pr.shoot_ray = orig
            resc = [np.nan] * 6 if rayc is None else [rayc.launch_angle, rayc.t[-1], rayc.z[-1], rayc.p[-1], rayc.n_bottom, rayc.n_surface]
            out.update({f"w{j}_coarse_idx": ic, f"w{j}_coarse_z": np.array([zc1, zc2]), f"w{j}_coarse_theta_seq": np.array(log),
                        f"w{j}_coarse_eigen": np.array(resc, float)})
            print(f"    coarse bracket: trial angles {len(log)}; eigenray {resc[:3]}")
    save("g10_eigenrays_1000km.npz", **out)


# --------------------------------------------------------------------------
# round 4: wider reference-produced pins at the headline range (a fork pool: the stand-in modules are
# inherited, each ray is one task like the reference's own pool, REF/launch_rays.py:157-164)
# --------------------------------------------------------------------------
_POOL_ARRS = {}


def _one_ray_with_noise(task):
    """One reference ray (array level, as _shoot_single_ray_process does) + the end states of its six
    neighbours at +-1, 2, 3 ulp of p0 (the reference's own self-noise)."""
    key, source_depth, x0, x1, th, S = task
    arrs = _POOL_ARRS[key]
    one = shoot_array_level(arrs, source_depth, x0, x1, np.array([th]), S)
    p0 = one["y0"][0, 2]
    ends = [np.array([one["T"][0, -1], one["z"][0, -1], one["p"][0, -1]])]
    for ulps in (-3, -2, -1, 1, 2, 3):   # (SURVEY section 8c measured the self-noise with +-1 ... 3 ulp on p0)
        p = p0
        for _ in range(abs(ulps)):
            p = np.nextafter(p, np.sign(ulps) * np.inf)
        sols, fr, _, _ = lr._shoot_ray_array(np.array([0, source_depth, p]), source_depth, x0, x1,
                                             *arrs, 1e-9, True, False)
        ends.append(sols[-1].y[:, -1].copy() if fr is not None else np.full(3, np.nan))
    one["neighbour_ends"] = np.array(ends)[None, :, :]                 # [ray][0, -3, -2, -1, +1, +2, +3 ulp][T, z, p]
    one["selfnoise_end"] = np.ptp(np.array(ends), axis=0)[None, :]
    return one


def shoot_pool(key, arrs, source_depth, x0, x1, theta, S, procs=7):
    import multiprocessing as mp
    _POOL_ARRS[key] = arrs
    t0 = time.time()
    with mp.get_context("fork").Pool(procs) as pool:
        parts = pool.map(_one_ray_with_noise, [(key, source_depth, x0, x1, float(th), S) for th in theta], chunksize=1)
    out = {k: (parts[0][k] if k == "r" else np.concatenate([q[k] for q in parts])) for k in parts[0]}
    print(f"  {len(theta)} reference rays (+ 6 neighbours at 1 ... 3 ulp of p0 each) in {time.time() - t0:.0f} s; "
          f"{int(((out['n_bott'] + out['n_surf']) > 0).sum())} bouncing, {int((out['ok'] == 0).sum())} dropped")
    return out


def g11_munk_1000km_wide():
    """BASELINE configs[1] tables, 288 launch angles (256 over +-20 degrees + 32 steep ones) at 1000 km (the steep third of
    the fan bounces), S = 101, step counts and the reference's +-1-ulp self-noise per ray."""
    z = np.arange(0, 6000, 1.0)
    env = munk_env(1000e3, 100, z)
    arrs = env_arrays(env)
    # (+ 32 more steep angles, off the linspace grid: >= 100 of the 288 rays bounce, up to 64 times)
    theta = np.concatenate([np.linspace(-20.0, 20.0, 256), -np.linspace(13.37, 19.93, 16), np.linspace(13.61, 19.77, 16)])
    out = shoot_pool("g11", arrs, 1000.0, 0.0, 1000e3, theta, 101)
    save("g11_munk_1000km_288.npz", theta_ode=theta, source_depth=1000.0, source_range=0.0,
         receiver_range=1000e3, c_row=arrs[0][0], cp_row=arrs[1][0], rin=arrs[2], zin=arrs[3],
         depths=arrs[4], depth_ranges=arrs[5], bottom_angles=arrs[6], **out)


def g12_config2_wide():
    """BASELINE configs[2] (sofar axis drifting 2e-4 m/m, dz = 1 m, 1000 km): 128 launch angles over +-20 degrees.
    The 9.7 MB tables are rebuilt by the test from the formula (checksums stored)."""
    z1 = np.arange(0, 6000, 1.0)
    r1 = np.linspace(0.0, 1000e3, 101)
    c2 = np.array([munk_ssp(z1, sofar_depth=1300 + 2e-4 * ri) for ri in r1])
    env2 = DuckEnv(c2, r1, z1, np.full(101, 5000.0), r1)
    arrs2 = env_arrays(env2)
    theta2 = np.linspace(-20.0, 20.0, 128)
    out2 = shoot_pool("g12", arrs2, 1000.0, 0.0, 1000e3, theta2, 101)
    save("g12_config2_128.npz", theta_ode=theta2, source_depth=1000.0, sofar_slope=2e-4, nr=101, r_max=1000e3,
         cp_checksum=np.sum(arrs2[1]), c_checksum=np.sum(arrs2[0]), **out2)


def g13_default_environment():
    """The reference's DEFAULT environment, OceanEnvironment2D() (REF/environment.py:62-119): Munk on
    arange(0, 6000, 1), 100 range columns, the 4500 -> 4900 m slope (Q11), flat-earth transform at lat 35
    (non-uniform zin), bottom angle from the UNtransformed bathymetry (Q10).  64 angles at 100 km as the
    constructor builds it, and 32 angles at 1000 km with the same tables on a range axis stretched to 1000 km
    (the bench's flat-earth leg's range).  Only zin / one profile row / the bathymetry are stored."""
    z = np.arange(0, 6000, 1)
    for tag, rmax, theta in (("100km", 100e3, np.linspace(-18.0, 18.0, 64)),
                             ("1000km", 1000e3, np.sort(np.concatenate([np.linspace(-19.9, 19.5, 24), [-16.3, -14.1, -12.2, -10.4, 11.3, 13.2, 15.7, 18.1]])))):
        r = np.linspace(0, rmax, 100)
        env = DuckEnv(np.array([munk_ssp(z)] * 100), r, z, np.linspace(4500, 4900, 100), r, lat=35, flat=True)
        arrs = env_arrays(env, flatearth=True)
        assert all(np.array_equal(arrs[0][0], row) for row in arrs[0])
        out = shoot_pool("g13" + tag, arrs, 1000.0, 0.0, rmax, theta, 101)
        save(f"g13_default_env_{tag}.npz", theta_ode=theta, source_depth=1000.0, source_range=0.0, receiver_range=rmax,
             c_row=arrs[0][0], cp_row=arrs[1][0], rin=arrs[2], zin=arrs[3], depths=arrs[4], depth_ranges=arrs[5],
             bottom_angles=arrs[6], **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g6", "g7", "g8", "g9"]
    table = dict(g1=g1_fixture_case, g2=g2_munk_100km, g3=g3_munk_1000km, g4=g4_range_dependent,
                 g5=g5_analytic_envs, g6=g6_eigenrays, g7=g7_unit_vectors, g8=g8_timing,
                 g9=g9_irregular_grids, g10=g10_eigenrays_1000km,
                 g11=g11_munk_1000km_wide, g12=g12_config2_wide, g13=g13_default_environment)
    for w in which:
        print(w)
        table[w]()
