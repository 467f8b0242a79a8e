"""NumPy/SciPy restatement of pygenray's per-ray integrator -- TEST INFRASTRUCTURE ONLY.

This is the "pygenray NumPy/SciPy path" stand-in that can travel to the GPU box (the
reference cannot): the same ``scipy.integrate.solve_ivp(RK45, rtol, atol=1e-6 default,
dense_output=True, 4 terminal +-1 events)`` call pattern as REF/launch_rays.py:593-681,
the same bounce loop as REF/launch_rays.py:325-484 and the same nearest-index re-sampling
as REF/launch_rays.py:745-784, with a plain-NumPy right-hand side restating
REF/integration_processes.py:26-334.  (REF = /root/reference/src/pygenray.)

It is used (a) to pin the C oracle's RK45/brentq/dense-output restatement against the real
SciPy, and (b) as the SciPy-path CPU baseline in bench.py.  Pinned against the golden
vectors in tests/test_oracle_golden.py.
"""
import math

import numpy as np
import scipy.integrate
import scipy.interpolate


def _cell(grid, q):
    # REF/integration_processes.py:152-157: searchsorted(side='left') - 1, index clamped
    k = int(np.searchsorted(grid, q)) - 1
    return max(0, min(k, len(grid) - 2))


def bilinear(x, z, rin, zin, tab):
    # REF/integration_processes.py:101-174 (weights are NOT clamped: Q4)
    i = _cell(rin, x)
    j = _cell(zin, z)
    wx = (x - rin[i]) / (rin[i + 1] - rin[i])
    wz = (z - zin[j]) / (zin[j + 1] - zin[j])
    return ((1 - wx) * (1 - wz) * tab[i, j] + wx * (1 - wz) * tab[i + 1, j]
            + (1 - wx) * wz * tab[i, j + 1] + wx * wz * tab[i + 1, j + 1])


def linear(x, xin, yin):
    # REF/integration_processes.py:177-235
    i = _cell(xin, x)
    w = (x - xin[i]) / (xin[i + 1] - xin[i])
    return (1 - w) * yin[i] + w * yin[i + 1]


class Tables:
    """The 7-array environment contract of REF/multi_processing.py:37-45."""

    def __init__(self, cin, cpin, rin, zin, depths, depth_ranges, bottom_angles):
        f = lambda a: np.ascontiguousarray(a, dtype=np.float64)  # noqa: E731
        self.cin, self.cpin, self.rin, self.zin = f(cin), f(cpin), f(rin), f(zin)
        self.depths, self.depth_ranges, self.bottom_angles = f(depths), f(depth_ranges), f(bottom_angles)

    # ---- REF/integration_processes.py:26-98
    def rhs(self, x, y):
        c = bilinear(x, y[1], self.rin, self.zin, self.cin)
        cz = bilinear(x, y[1], self.rin, self.zin, self.cpin)
        a = 1.0 - (c * c) * (y[2] * y[2])
        if a <= 0.0:
            a = 1e-30
        s = 1 / math.sqrt(a)
        return np.array([s / c, c * y[2] * s, -s * cz / (c * c)])

    # ---- REF/integration_processes.py:306-334
    def angle(self, x, y):
        c = bilinear(x, y[1], self.rin, self.zin, self.cin)
        v = y[2] * c
        th = math.degrees(math.asin(v)) if abs(v) <= 1.0 else math.nan
        return th, c

    # ---- REF/integration_processes.py:238-303
    def ev_surface(self, x, y):
        th, _ = self.angle(x, y)
        return 1.0 if (y[1] < 0 and th < 0) else -1.0

    def ev_bottom(self, x, y):
        th, _ = self.angle(x, y)
        return 1.0 if (y[1] > linear(x, self.depth_ranges, self.depths) and th > 0) else -1.0

    def ev_vertical(self, x, y):
        th, _ = self.angle(x, y)
        return 1.0 if abs(th) > (90 - 1e-3) else -1.0

    def ev_bbox(self, x, y):
        t = 1e-6
        out = (y[1] > self.zin[-1] + t) or (y[1] < self.zin[0] - t) or (x < self.rin[0] - t) \
            or (x > self.rin[-1] + t)
        return 1.0 if out else -1.0

    def event_list(self):
        def mk(fn, direction):
            g = lambda x, y: fn(x, y)  # noqa: E731
            g.terminal = True
            g.direction = direction
            return g
        # REF/launch_rays.py:649-668: surface/bottom direction +1, the others default 0
        return [mk(self.ev_surface, 1), mk(self.ev_bottom, 1), mk(self.ev_vertical, 0),
                mk(self.ev_bbox, 0)]


def shoot_one(tb, y0, source_range, receiver_range, num_range_save, rtol=1e-9,
              terminate_backwards=True, **ivp_kwargs):
    """One ray. Returns (status, T[S], z[S], p[S], n_bott, n_surf, n_steps, nfev); status 0 = ok,
    else the reason the reference would have returned None (codes as in ray_oracle.c)."""
    S = int(num_range_save)
    grid = np.linspace(source_range, receiver_range, S)
    out = np.full((3, S), np.nan)
    beta = scipy.interpolate.interp1d(tb.depth_ranges, tb.bottom_angles, kind="cubic")
    x, y = float(source_range), np.array(y0, dtype=float)
    nb = ns = nsteps = nfev = 0
    evs = tb.event_list()
    last = None
    while x < receiver_range:
        sol = scipy.integrate.solve_ivp(tb.rhs, (x, receiver_range), y, events=evs, rtol=rtol,
                                        dense_output=True, **ivp_kwargs)
        nsteps += len(sol.t) - 1
        nfev += sol.nfev
        # nearest-index slice of the save grid (Q5)
        a = int(np.argmin(np.abs(grid - sol.t[0])))
        b = int(np.argmin(np.abs(grid - sol.t[-1])))
        if a != b:
            out[:, a:b] = sol.sol(grid[a:b])
        last = sol.y[:, -1].copy()
        if sol.status == 0:
            break
        if sol.status == -1:
            return 4, None, None, None, nb, ns, nsteps, nfev
        y = last.copy()
        if len(sol.t_events[0]) > 0:
            x = sol.t_events[0][0]
        elif len(sol.t_events[1]) > 0:
            x = sol.t_events[1][0]
        elif len(sol.t_events[2]) > 0:
            return 1, None, None, None, nb, ns, nsteps, nfev
        else:
            return 2, None, None, None, nb, ns, nsteps, nfev
        th, c = tb.angle(x, y)
        if len(sol.t_events[0]) == 1:
            th2 = -th
            ns += 1
        else:
            try:
                th2 = 2 * float(beta(x)) - th
            except ValueError:
                return 6, None, None, None, nb, ns, nsteps, nfev
            nb += 1
        if terminate_backwards and abs(th2) > 90:
            return 3, None, None, None, nb, ns, nsteps, nfev
        y[2] = math.sin(math.radians(th2)) / c
    out[:, -1] = last
    return 0, out[0], out[1], out[2], nb, ns, nsteps, nfev


def shoot_fan(cin, cpin, rin, zin, depths, depth_ranges, bottom_angles, y0, source_range,
              receiver_range, num_range_save, rtol=1e-9, terminate_backwards=True):
    tb = Tables(cin, cpin, rin, zin, depths, depth_ranges, bottom_angles)
    y0 = np.asarray(y0, dtype=float).reshape(-1, 3)
    N, S = len(y0), int(num_range_save)
    res = dict(r=np.linspace(source_range, receiver_range, S), T=np.full((N, S), np.nan),
               z=np.full((N, S), np.nan), p=np.full((N, S), np.nan), n_bott=np.zeros(N, np.int32),
               n_surf=np.zeros(N, np.int32), status=np.zeros(N, np.int32),
               n_steps=np.zeros(N, np.int64), nfev=np.zeros(N, np.int64))
    for k in range(N):
        st, T, Z, P, nb, ns, nst, nfe = shoot_one(tb, y0[k], source_range, receiver_range, S, rtol,
                                                  terminate_backwards)
        res["status"][k], res["n_bott"][k], res["n_surf"][k] = st, nb, ns
        res["n_steps"][k], res["nfev"][k] = nst, nfe
        if st == 0:
            res["T"][k], res["z"][k], res["p"][k] = T, Z, P
    return res


def _pool_worker(args):
    tabs, y0, x0, x1, S, rtol = args
    tb = Tables(*tabs)
    st, _, _, _, _, _, nst, _ = shoot_one(tb, y0, x0, x1, S, rtol)
    return nst
