"""``shoot_rays`` / ``shoot_ray``: pygenray's fan and single-ray entry points on the HIP path.

Mirrors ``pygenray.launch_rays`` (REF = /root/reference/src/pygenray): same names, arguments,
defaults, sign conventions and error behaviour as REF/launch_rays.py:11-322; the per-ray
spawn pool + shared-memory transport (REF/launch_rays.py:133-198, multi_processing.py) is
replaced by one table upload per environment and one kernel launch per fan.
"""
import numpy as np

from . import _lib
from .environment import _unpack_envi, _mirror_envi_arrays, _check_monotone
from .host_physics import bilinear_interp
from .ray_objects import Ray, RayFan

_DROP_MSG = {
    1: "ray is vertical, terminating integration",
    2: "ray left bounding box, terminating integration",
    3: "ray bounced backwards, terminating integration",
    4: "Integration failed with message: Required step size is less than spacing between numbers.",
    5: "ray exceeded the step limit, terminating integration",
    6: "Error in ray integration: A value in x_new is outside the interpolation range.",
    7: "Error in ray integration: f(a) and f(b) must have different signs",
}


def _device_env(environment, flatearth, backwards, device=0):
    """EnvHandle (tables resident in HBM) for this environment, cached on it."""
    cache = getattr(environment, "_cache", None)
    if cache is None:
        cache = environment._cache = {}
    # keyed on the identity of the tables too: replacing env.sound_speed / env.bathymetry (or
    # re-running the flat-earth transform) uploads afresh; in-place edits of .values do not
    src = (getattr(environment, "sound_speed_fe", None) if flatearth else environment.sound_speed,
           getattr(environment, "bathymetry_fe", None) if flatearth else environment.bathymetry)
    key = ("dev", bool(flatearth), bool(backwards), int(device), id(src[0]), id(src[1]))
    if key not in cache:
        cin, cpin, rin, zin, depths, depth_ranges, bottom_angles = _unpack_envi(
            environment, flatearth=flatearth)
        _check_monotone(rin, zin, depth_ranges)
        if backwards:
            cin, cpin, rin, depths, depth_ranges, bottom_angles = _mirror_envi_arrays(
                cin, cpin, rin, depths, depth_ranges, bottom_angles)
        cache[key] = (_lib.EnvHandle(cin, cpin, rin, zin, depths, depth_ranges, bottom_angles,
                                     device=device), (cin, rin, zin))
    return cache[key]


def _shoot_ode_angles(source_depth, source_range, ode_angles_deg, receiver_range, num_range_save,
                      environment, rtol, terminate_backwards, flatearth, device=0, save=True,
                      max_steps=1_000_000, stored_sign=False, compact=False):
    """Batched _shoot_single_ray_process (REF/launch_rays.py:487-590): ODE-convention launch
    angles in, ODE-convention SoA out (in the mirrored frame for backwards shots)."""
    backwards = receiver_range < source_range
    env, (cin, rin, zin) = _device_env(environment, flatearth, backwards, device)
    if backwards:
        source_range_i, receiver_range_i = -source_range, -receiver_range
    else:
        source_range_i, receiver_range_i = source_range, receiver_range
    # REF/launch_rays.py:140-144
    c = bilinear_interp(source_range_i, source_depth, rin, zin, cin)
    ang = np.asarray(ode_angles_deg, dtype=float).reshape(-1)
    y0 = np.zeros((len(ang), 3))
    y0[:, 1] = source_depth
    y0[:, 2] = _initial_slowness(ang, c)
    if not (source_range_i < receiver_range_i):
        # the reference's bounce loop never runs and _interpolate_ray indexes an empty list
        raise IndexError("list index out of range")
    # trajectories come back sample-major ([S][N]: coalesced stores on the device, 4x less HBM
    # write traffic) and are handed on as transposed (N, S) views -- same indexing as pygenray
    out = env.shoot_fan(y0, source_range_i, receiver_range_i, num_range_save, rtol=rtol,
                        terminate_backwards=terminate_backwards, save=save, max_steps=max_steps,
                        sample_major=True, stored_sign=stored_sign, compact=compact)
    if save:
        out["T"], out["z"], out["p"] = out["T"].T, out["z"].T, out["p"].T
    if backwards:
        out["r"] = -out["r"]
    return out


def _initial_slowness(ode_angles_deg, c):
    """sin(radians(theta)) / c per ray (REF/launch_rays.py:140-144) with NumPy's own sine; a million-angle fan
    splits the (GIL-free) ufunc calls over a few threads -- the same routine on every element, the same bits."""
    ang = np.asarray(ode_angles_deg, dtype=float).reshape(-1)
    if len(ang) < 400_000:
        return np.sin(np.radians(ang)) / c
    from concurrent.futures import ThreadPoolExecutor
    out = np.empty(len(ang))
    nt = 8
    per = (len(ang) + nt - 1) // nt

    def part(k):
        sl = slice(k * per, (k + 1) * per)
        np.divide(np.sin(np.radians(ang[sl])), c, out=out[sl])
    with ThreadPoolExecutor(nt) as ex:
        list(ex.map(part, range(nt)))
    return out


def _launch_device_fan(source_depth, source_range, ode_angles_deg, receiver_range, num_range_save, environment, rtol,
                       terminate_backwards, flatearth, device=0, max_steps=1_000_000, stored_sign=True,
                       device_y0=False, spread=1):
    """The fan of _shoot_ode_angles launched device-resident (``_lib.FanHandle``): returns (handle, r) while the kernel
    runs.  ``device_y0``: the initial states are computed on the device from the angles (correctly rounded sine) instead of
    from NumPy's sin(radians(.)) / c on the host (the default, the reference's arithmetic: fans, eigenray trial rays and the
    eigenrays handed back all use it).  ``spread``: ray k of the caller is
    ray k * spread of the launch, the rays between are padding that is never integrated (status 8) -- unrelated rays
    (the eigenrays of different brackets) get a wave each instead of bouncing in each other's way."""
    backwards = receiver_range < source_range
    env, (cin, rin, zin) = _device_env(environment, flatearth, backwards, device)
    x0, x1 = (-source_range, -receiver_range) if backwards else (source_range, receiver_range)
    if not (x0 < x1):
        raise IndexError("list index out of range")
    c = bilinear_interp(x0, source_depth, rin, zin, cin)
    ang = np.asarray(ode_angles_deg, dtype=float).reshape(-1)
    kw = dict(rtol=rtol, terminate_backwards=terminate_backwards, max_steps=max_steps, stored_sign=stored_sign)
    if device_y0:
        if spread > 1:
            padded = np.full(len(ang) * int(spread), np.nan)
            padded[::int(spread)] = ang
            ang, kw = padded, dict(kw, skip_nan=True)
        h = _lib.FanHandle(env, x0, x1, num_range_save, ode_angles_deg=ang, source_depth=source_depth, c_source=c, **kw)
    else:
        # NumPy's own sin(radians(.)) / c, as the reference computes it; [0, z_s, p0] is assembled on the device
        p0 = _initial_slowness(ang, c)
        if spread > 1:
            padded = np.full(len(p0) * int(spread), np.nan)
            padded[::int(spread)] = p0
            p0, kw = padded, dict(kw, skip_nan=True)
        h = _lib.FanHandle(env, x0, x1, num_range_save, p0=p0, source_depth=source_depth, **kw)
    r = np.linspace(x0, x1, int(num_range_save))
    return h, (-r if backwards else r)


def _report_drops(status, debug):
    if debug:
        for s in status[status != 0]:
            print(_DROP_MSG.get(int(s), f"ray dropped (status {int(s)})"))


def shoot_rays(source_depth, source_range, launch_angles, receiver_range, num_range_save,
               environment, rtol=1e-9, terminate_backwards=True, n_processes=None, debug=True,
               flatearth=True, device=0, device_resident=None):
    """Integrate a fan of rays (REF/launch_rays.py:11-200) -> ``RayFan``.

    ``n_processes`` is accepted for compatibility and ignored (the fan is one GPU launch).
    Launch-angle sign: like the reference, a fan of fewer than 70 angles is integrated with
    ODE angle = +user angle and a larger fan with ODE angle = -user angle (Q1 in SURVEY.md:
    REF/launch_rays.py:67,94,251); dropped rays vanish from the fan (Q12).

    ``device_resident`` (not a reference argument): True -- the call returns when the kernel has finished and the
    per-ray arrays are on the host; the fan's ``ts`` / ``zs`` / ``ps`` stay in HBM and cross PCIe when they are first
    read (``RayFan.from_device``; same values, same shapes); False -- everything is copied before the call returns;
    None (default) -- device resident from 2 million samples per array on (a 1e5 x 1001 fan: 7 ms instead of 50)."""
    if type(launch_angles) is list:
        launch_angles = np.array(launch_angles)
    user = np.asarray(launch_angles, dtype=float)
    n = len(user)
    if n < 70:
        ode = stored = user      # shoot_ray flips the (already flipped) angle again and stores it (REF/launch_rays.py:251,318)
    else:
        ode = -user              # REF/launch_rays.py:67
        stored = user            # REF/launch_rays.py:180
    if device_resident is None:
        device_resident = n * int(num_range_save) >= 2_000_000
    if device_resident and n > 0:
        h, r = _launch_device_fan(source_depth, source_range, ode, receiver_range, num_range_save, environment, rtol,
                                  terminate_backwards, flatearth, device=device, stored_sign=True)
        if debug:
            _report_drops(h.status(), debug)
        h.wait()                                    # the kernel; then only the surviving rays' per-ray arrays cross PCIe
        rays = h.fetch_rays_compact(per_ray=stored)
        return RayFan.from_device(h, rays["per_ray"], r, rays["end"], rays["n_bott"], rays["n_surf"],
                                  np.full(h.M, source_depth))
    # stored convention z -> -z, p -> -p (REF/ray_objects.py:51-52) applied by the kernel's stores
    out = _shoot_ode_angles(source_depth, source_range, ode, receiver_range, num_range_save,
                            environment, rtol, terminate_backwards, flatearth, device=device,
                            stored_sign=True, compact=True)
    _report_drops(out["status"], debug)
    keep = out["status"] == 0
    S = len(out["r"])
    M = int(keep.sum())
    rs = np.tile(out["r"], (M, 1)) if M * S <= 20_000_000 else np.broadcast_to(out["r"], (M, S))
    T, Z, P = out["T"], out["z"], out["p"]  # (M, S) views: dropped rays were squeezed out on the device
    return RayFan.from_arrays(stored[keep], rs, T, Z, P, out["n_bott"][keep].astype(np.int64),
                              out["n_surf"][keep].astype(np.int64), np.full(M, source_depth))


def shoot_ray(source_depth, source_range, launch_angle, receiver_range, num_range_save, environment,
              rtol=1e-9, terminate_backwards=True, debug=True, flatearth=True, device=0):
    """Integrate one ray (REF/launch_rays.py:203-322) -> ``Ray`` or ``None`` if it was dropped.
    ``Ray.launch_angle`` is the negated user angle, as in the reference (Q2)."""
    launch_angle = -launch_angle
    out = _shoot_ode_angles(source_depth, source_range, [launch_angle], receiver_range,
                            num_range_save, environment, rtol, terminate_backwards, flatearth,
                            device=device)
    _report_drops(out["status"], debug)
    if out["status"][0] != 0:
        return None
    y = np.stack([out["T"][0], out["z"][0], out["p"][0]])
    return Ray(out["r"], y, int(out["n_bott"][0]), int(out["n_surf"][0]), launch_angle, source_depth)


__all__ = ["shoot_rays", "shoot_ray", "_unpack_envi"]
