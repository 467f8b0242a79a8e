"""Eigenray search: bracket detection + regula falsi, batched over all brackets on the GPU.

Mirrors ``pygenray.eigenrays`` (REF = /root/reference/src/pygenray, eigenrays.py:11-268):
brackets are sign changes of ``z_end + receiver_depth`` between neighbouring fan rays; each
bracket is refined by false position, re-shooting one trial ray per iteration, until
``|z_end + receiver_depth| < ztol`` or the iteration budget is spent.  The reference maps
brackets to a process pool and re-unpacks the environment for every trial ray; here every
iteration is ONE fan launch holding the trial rays of all still-active brackets.
"""
import os

import numpy as np

from .ray_objects import RayFan, EigenRays

# diagnostics of the last find_eigenrays call (bench.py reports them): fan launches and trial rays
LAST_SEARCH_STATS = {}


def _regula_falsi_batch(z1, z2, th1, th2, receiver_depth, source_depth, source_range,
                        receiver_range, num_range_save, environment, ztol, max_iter, kwargs):
    """_find_single_eigenray (REF/eigenrays.py:206-268) for all brackets at once.  The whole
    false-position loop runs on the device (pgr_eigen_refine_depths: per iteration one small kernel for the
    reference's loop body + the trial rays' initial states, one fan launch over the brackets still
    active); the eigenrays found are then shot once more with trajectories.  `receiver_depth`: one depth, or one per
    bracket (the brackets of all receiver depths of a call searched together).
    Returns (found mask, launch angles, r, T, Z, P (stored convention), n_bott, n_surf)."""
    from .launch_rays import _device_env
    from .host_physics import bilinear_interp
    rtol = kwargs.get("rtol", 1e-9)
    terminate_backwards = kwargs.get("terminate_backwards", True)
    flatearth = kwargs.get("flatearth", True)
    device = kwargs.get("device", 0)
    quiet = kwargs.get("quiet", False)   # (not a reference kwarg: suppresses the per-bracket failure message)
    nbk = len(z1)
    rd_k = np.broadcast_to(np.asarray(receiver_depth, dtype=float), (nbk,))
    S = int(num_range_save)
    backwards = receiver_range < source_range
    env, (cin, rin, zin) = _device_env(environment, flatearth, backwards, device)
    x0, x1 = (-source_range, -receiver_range) if backwards else (source_range, receiver_range)
    c0 = bilinear_interp(x0, source_depth, rin, zin, cin)            # REF/launch_rays.py:284
    # ONE arithmetic for every initial slowness, the reference's: NumPy's sin(radians(.)) / c (REF/launch_rays.py:284-285)
    # -- for the fan (launch_rays._initial_slowness), for the trial rays of the device loop (handed in as a callback: the
    # loop sends the active brackets' angles down once per iteration) and for the eigenrays re-shot below.  So
    # shoot_ray(theta) of an eigenray's angle starts from the same bits as the eigenray itself.
    from .launch_rays import _initial_slowness
    out = env.eigen_refine(th1, th2, z1, z2, rd_k, source_depth, x0, x1, c0, rtol=rtol,
                           terminate_backwards=terminate_backwards, ztol=ztol, max_iter=max_iter,
                           slowness=lambda ang: _initial_slowness(ang, c0))
    LAST_SEARCH_STATS.setdefault("reshot_differs", 0)
    LAST_SEARCH_STATS["launches"] = LAST_SEARCH_STATS.get("launches", 0) + out["launches"]
    LAST_SEARCH_STATS["trial_rays"] = LAST_SEARCH_STATS.get("trial_rays", 0) + int(out["n_trial"].sum())
    if not quiet:
        for q in np.where(out["state"] == 2)[0]:
            # REF/eigenrays.py:241-245
            print(f"Failed to find eigen ray for receiver depth {rd_k[q]} [m] and "
                  f"approximate launch angle {out['theta'][q]} [m] ray θ = 90°")
    found = out["state"] == 1
    th_found = np.where(found, out["theta"], 0.0)
    T = np.zeros((nbk, S)); Z = np.zeros((nbk, S)); P = np.zeros((nbk, S))
    nb = np.zeros(nbk, np.int64); ns = np.zeros(nbk, np.int64)
    r = np.linspace(source_range, receiver_range, S)
    if found.any():
        idx = np.where(found)[0]
        # the eigenrays themselves, with trajectories: shoot_ray(theta), ODE angle = -theta (REF/launch_rays.py:251), from
        # the SAME initial state as the trial ray the search accepted -- NumPy's sin(radians(.)) / c, as the callback
        # above computed it -- so the ray returned IS the accepted one
        from .launch_rays import _launch_device_fan
        # (one ray per wave while they fit one wave per SIMD: the eigenrays of different brackets bounce at different ranges)
        # -- and while the padding stays small: the launch owns 3 * N * S * 8 B of HBM for N = spread * len(idx) rays and
        # the compact fetch sizes its host buffers alike (1000 eigenrays x S = 1001 would be 1.5 GB of each for 24 MB of data)
        spread = 64 if (len(idx) <= 1024 and len(idx) * 64 * S * 24 <= 256 * 1024 * 1024) else 1
        h, r = _launch_device_fan(source_depth, source_range, -th_found[idx], receiver_range, S, environment, rtol,
                                  terminate_backwards, flatearth, device=device, stored_sign=True, spread=spread)
        LAST_SEARCH_STATS["launches"] += 1
        rays = {k: v[::spread] for k, v in h.fetch_rays().items()}
        if not np.all(rays["status"] == 0):
            raise RuntimeError("find_eigenrays: a re-shot eigenray differs from the trial ray the search accepted")
        smp = h.fetch_samples(("T", "z", "p"), compact=True)      # (the padding rays have status 8: squeezed out on the device)
        h.close()
        if not np.all(np.abs(-rays["end"][:, 1] + rd_k[idx]) < ztol):
            raise RuntimeError("find_eigenrays: a re-shot eigenray differs from the trial ray the search accepted")
        # (the ztol check above is the guard the user relies on.  That the re-shot ray -- trajectory kernel -- ends on the SAME
        # BITS as the accepted trial ray -- end-state kernel -- is a property of the build that tests assert
        # (PGR_EIGEN_STRICT=1, set for the whole test suite by tests/conftest.py); a one-ulp divergence between two kernel
        # instances must not fail a user's whole search: the affected brackets are recorded and a warning is raised instead)
        differs = -rays["end"][:, 1] != out["z_end"][idx]
        LAST_SEARCH_STATS["reshot_differs"] = LAST_SEARCH_STATS.get("reshot_differs", 0) + int(differs.sum())
        if differs.any():
            if os.environ.get("PGR_EIGEN_STRICT") == "1":     # (tests/conftest.py sets it for the whole suite)
                raise RuntimeError("find_eigenrays: a re-shot eigenray does not end where its trial ray did")
            import warnings
            warnings.warn(f"find_eigenrays: {int(differs.sum())} re-shot eigenray(s) end within ztol but not on the bits of "
                          "the trial ray the search accepted (two kernel instances diverged)", RuntimeWarning)
        T[idx], Z[idx], P[idx] = smp["T"].T, smp["z"].T, smp["p"].T
        nb[idx], ns[idx] = rays["n_bott"], rays["n_surf"]
    return found, th_found, r, T, Z, P, nb, ns


def _bracket(rays, receiver_depth):
    """Brackets of one receiver depth on the WHOLE fan, exactly as REF/eigenrays.py:65-79: sign changes of
    z_end + receiver_depth between neighbouring fan rays (dropped rays have vanished from the fan, Q12, so a
    bracket may span one)."""
    z_end = rays.zs_end if hasattr(rays, "zs_end") else rays.zs[:, -1]   # (a device-resident fan keeps its trajectories in HBM)
    depth_sign = np.sign(z_end + receiver_depth)
    starts = np.where(np.diff(depth_sign))[0]
    return starts, z_end[starts], z_end[starts + 1], rays.thetas[starts], rays.thetas[starts + 1]


def _find_eigenrays(rays, receiver_depths, source_depth, num_range_save, environment, refine, together=False):
    """The frame of find_eigenrays (REF/eigenrays.py:62-203) around `refine(z1s, z2s, th1s, th2s, receiver_depth)
    -> (found, th, r, T, Z, P, nb, ns)` over ALL brackets of one receiver depth: the single-GPU search refines them in
    one batch, the sharded one (distributed.find_eigenrays_sharded) deals them to the ranks and gathers.
    `together`: the brackets of ALL receiver depths go into ONE call of `refine`, which then gets an array of receiver
    depths, one per bracket (the device loop's iterations last as long as their slowest trial ray whatever the number
    of brackets: R receiver depths cost one search instead of R); the results are dealt back depth by depth."""
    erays_dict, num_eigenrays, num_found, failed = {}, {}, {}, {}
    brackets = [_bracket(rays, receiver_depth) for receiver_depth in receiver_depths]
    refined = {}
    if together and sum(len(b[0]) for b in brackets) > 0:
        cat = [np.concatenate([b[k] for b in brackets]) for k in range(1, 5)]      # z1s, z2s, th1s, th2s
        rd_k = np.concatenate([np.full(len(b[0]), float(rd)) for b, rd in zip(brackets, receiver_depths)])
        found, th, r, T, Z, P, nb, ns = refine(cat[0], cat[1], cat[2], cat[3], rd_k)
        o = 0
        for rd_idx, b in enumerate(brackets):
            sl = slice(o, o + len(b[0])); o += len(b[0])
            refined[rd_idx] = (found[sl], th[sl], r, T[sl], Z[sl], P[sl], nb[sl], ns[sl])
    for rd_idx, receiver_depth in enumerate(receiver_depths):
        starts, z1s, z2s, th1s, th2s = brackets[rd_idx]
        num_eigenrays[receiver_depth] = len(starts)
        failed[rd_idx] = []
        if len(starts) == 0:
            erays_dict[rd_idx] = RayFan.from_arrays(
                np.zeros(0), np.zeros((0, num_range_save)), np.zeros((0, num_range_save)),
                np.zeros((0, num_range_save)), np.zeros((0, num_range_save)), np.zeros(0, np.int64),
                np.zeros(0, np.int64), np.zeros(0))
            num_found[rd_idx] = 0
            continue
        found, th, r, T, Z, P, nb, ns = refined[rd_idx] if rd_idx in refined else refine(z1s, z2s, th1s, th2s, receiver_depth)
        for k in np.where(~found)[0]:
            failed[rd_idx].append((th1s[k], th2s[k]))
        M = int(found.sum())
        erays_dict[rd_idx] = RayFan.from_arrays(
            th[found], np.tile(r, (M, 1)), T[found], Z[found], P[found], nb[found], ns[found],
            np.full(M, source_depth))
        num_found[rd_idx] = M
    return EigenRays(receiver_depths, erays_dict, environment, num_eigenrays, num_found, failed)


def find_eigenrays(rays, receiver_depths, source_depth, source_range, receiver_range,
                   num_range_save, environment, ztol=1, max_iter=20, num_workers=None, **kwargs):
    """Find eigenrays from an initial ray fan by regula falsi (REF/eigenrays.py:11-203).

    ``num_workers`` is accepted and ignored.  ``kwargs`` are those of ``shoot_ray``
    (``rtol, terminate_backwards, debug, flatearth``).  Returns ``EigenRays``."""
    def refine(z1s, z2s, th1s, th2s, receiver_depth):
        return _regula_falsi_batch(z1s, z2s, th1s, th2s, receiver_depth, source_depth, source_range, receiver_range,
                                   num_range_save, environment, ztol, max_iter, kwargs)
    return _find_eigenrays(rays, receiver_depths, source_depth, num_range_save, environment, refine, together=True)


__all__ = ["find_eigenrays"]
