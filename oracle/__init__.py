"""CPU oracle for the pygenray hot path -- TEST INFRASTRUCTURE ONLY.

Two restatements of the reference's per-ray integrator live here:

* ``ray_oracle.c``   plain C99 (fast; RK45, events, brentq and dense output restated)
* ``scipy_port.py``  NumPy + ``scipy.integrate.solve_ivp`` (the reference's own call pattern)

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package.  The product package (``pygenray_amd``) never does.
Parity pin: see ``tests/test_oracle_golden.py`` (reference fixture + vectors captured from
the reference itself, ``tests/golden/make_golden.py``).
"""
import ctypes
import os
import subprocess
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
# The arithmetic mode (libm / correctly rounded) is process-global state of the C library: every call that
# sets it, runs and restores it holds this lock, so that two threads using the oracle at once (or an exception
# between set and restore) cannot run a fan in the wrong mode -- the checker of every bit-parity test.
_MATH_LOCK = threading.RLock()

STATUS = {0: "ok", 1: "vertical", 2: "bbox", 3: "backward", 4: "step_too_small", 5: "max_steps",
          6: "bottom_angle_range", 7: "event_error"}


def build(force=False):
    """gcc build of ray_oracle.c.  ORACLE_SANITIZE=1 in the environment selects the AddressSanitizer +
    UBSan build (`make asan`; the process must have libasan preloaded, tests/test_sanitizers.py)."""
    san = os.environ.get("ORACLE_SANITIZE") == "1"
    target = "_asan/libray_oracle.so" if san else "libray_oracle.so"
    so = os.path.join(_HERE, target)
    src = os.path.join(_HERE, "ray_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", target], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(build())
        dp = ctypes.POINTER(ctypes.c_double)
        i64 = ctypes.c_int64
        L.orc_bilinear.restype = ctypes.c_double
        L.orc_bilinear.argtypes = [ctypes.c_double, ctypes.c_double, dp, i64, dp, i64, dp]
        L.orc_linear.restype = ctypes.c_double
        L.orc_linear.argtypes = [ctypes.c_double, dp, dp, i64]
        L.orc_brentq_step.restype = ctypes.c_double
        L.orc_brentq_step.argtypes = [ctypes.c_double] * 3 + [ctypes.POINTER(ctypes.c_int)]
        L.orc_num_threads.restype = ctypes.c_int
        for f in (L.orc_pow, L.orc_asin, L.orc_sin):
            f.restype = ctypes.c_double
        L.orc_pow.argtypes = [ctypes.c_double, ctypes.c_double]
        L.orc_asin.argtypes = [ctypes.c_double]
        L.orc_sin.argtypes = [ctypes.c_double]
        L.orc_set_math.argtypes = [ctypes.c_int]
        L.orc_get_math.restype = ctypes.c_int
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def bilinear(x, y, xg, yg, v):
    xg, yg, v = _c(xg), _c(yg), _c(v)
    return lib().orc_bilinear(float(x), float(y), _p(xg), len(xg), _p(yg), len(yg), _p(v))


def linear(x, xin, yin):
    xin, yin = _c(xin), _c(yin)
    return lib().orc_linear(float(x), _p(xin), _p(yin), len(xin))


def derivs(x, y, cin, cpin, rin, zin):
    cin, cpin, rin, zin, y = _c(cin), _c(cpin), _c(rin), _c(zin), _c(y)
    out = np.zeros(3)
    lib().orc_derivs(_p(cin), _p(cpin), _p(rin), _p(zin), ctypes.c_int64(len(rin)),
                     ctypes.c_int64(len(zin)), ctypes.c_double(x), _p(y), _p(out))
    return out


def ray_angle(x, y, cin, rin, zin):
    cin, rin, zin, y = _c(cin), _c(rin), _c(zin), _c(y)
    th, c = ctypes.c_double(), ctypes.c_double()
    lib().orc_ray_angle(_p(cin), _p(rin), _p(zin), ctypes.c_int64(len(rin)),
                        ctypes.c_int64(len(zin)), ctypes.c_double(x), _p(y), ctypes.byref(th),
                        ctypes.byref(c))
    return th.value, c.value


def events(x, y, cin, rin, zin, depths, depth_ranges):
    cin, rin, zin, y, depths, depth_ranges = map(_c, (cin, rin, zin, y, depths, depth_ranges))
    out = np.zeros(4)
    lib().orc_events(_p(cin), _p(rin), _p(zin), ctypes.c_int64(len(rin)), ctypes.c_int64(len(zin)),
                     _p(depths), _p(depth_ranges), ctypes.c_int64(len(depths)), ctypes.c_double(x),
                     _p(y), _p(out))
    return out


def bottom_angle_interp(depth_ranges, bottom_angles, xq):
    depth_ranges, bottom_angles, xq = _c(depth_ranges), _c(bottom_angles), _c(np.atleast_1d(xq))
    out = np.zeros(len(xq))
    rc = lib().orc_bottom_angle_interp(_p(depth_ranges), _p(bottom_angles),
                                       ctypes.c_int64(len(depth_ranges)), _p(xq),
                                       ctypes.c_int64(len(xq)), _p(out))
    if rc:
        raise ValueError("x and y arrays must have at least 4 entries")
    return out


def brentq_step(a, b, s):
    n = ctypes.c_int()
    r = lib().orc_brentq_step(a, b, s, ctypes.byref(n))
    return r, n.value


def num_threads():
    return lib().orc_num_threads()


def set_num_threads(n):
    """OpenMP threads used by shoot_fan (bench.py sizes this to the cores the process may use)."""
    lib().orc_set_num_threads(int(n))


MATH_LIBM, MATH_CR = 0, 1


def set_math(mode):
    """Which pow / arcsin / sin the integrator calls (see ray_oracle.c): MATH_LIBM = the platform
    libm, what NumPy/SciPy use here (the mode the golden vectors pin); MATH_CR = the same functions
    correctly rounded (libquadmath), the mode the HIP path must match bit for bit."""
    lib().orc_set_math(int(mode))


def get_math():
    return lib().orc_get_math()


def math_fn(name, a, math=MATH_CR):
    """The oracle's libm calls in the given mode, element-wise on an array (tests):
    name = "pow_m02" (a ** -0.2), "pow_p02" (a ** 0.2), "asin", "sin"."""
    L = lib()
    fn = {"pow_m02": 0, "pow_p02": 1, "asin": 2, "sin": 3}[name]
    a = _c(a)
    out = np.empty(a.shape)
    with _MATH_LOCK:
        old = L.orc_get_math()
        L.orc_set_math(int(math))
        try:
            L.orc_math_array(ctypes.c_int(fn), _p(a.reshape(-1)), _p(out.reshape(-1)), ctypes.c_int64(a.size))
        finally:
            L.orc_set_math(old)
    return out


def shoot_fan(cin, cpin, rin, zin, depths, depth_ranges, bottom_angles, y0, source_range,
              receiver_range, num_range_save, rtol=1e-9, atol=1e-6, terminate_backwards=True,
              max_steps=10_000_000, math=None):
    """Array-level fan: the argument list of the reference's _shoot_ray_array
    (launch_rays.py:325-340) batched over rays.  Returns a dict of ODE-convention arrays.
    ``math``: MATH_LIBM / MATH_CR for this call (default: the current mode, initially libm)."""
    if math is not None:
        with _MATH_LOCK:
            old = get_math()
            set_math(math)
            try:
                return shoot_fan(cin, cpin, rin, zin, depths, depth_ranges, bottom_angles, y0, source_range,
                                 receiver_range, num_range_save, rtol, atol, terminate_backwards, max_steps)
            finally:
                set_math(old)
    with _MATH_LOCK:   # (a call in the current mode must not overlap another thread's set / restore either)
        return _shoot_fan_locked(cin, cpin, rin, zin, depths, depth_ranges, bottom_angles, y0, source_range,
                                 receiver_range, num_range_save, rtol, atol, terminate_backwards, max_steps)


def _shoot_fan_locked(cin, cpin, rin, zin, depths, depth_ranges, bottom_angles, y0, source_range,
                      receiver_range, num_range_save, rtol, atol, terminate_backwards, max_steps):
    cin, cpin, rin, zin = _c(cin), _c(cpin), _c(rin), _c(zin)
    depths, depth_ranges, bottom_angles = _c(depths), _c(depth_ranges), _c(bottom_angles)
    y0 = _c(y0).reshape(-1, 3)
    N, S = len(y0), int(num_range_save)
    r = np.linspace(source_range, receiver_range, S)
    T = np.empty((N, S)); Z = np.empty((N, S)); P = np.empty((N, S)); XI = np.empty((N, S))
    nb = np.zeros(N, np.int32); ns = np.zeros(N, np.int32); st = np.zeros(N, np.int32)
    nsteps = np.zeros(N, np.int64); nfev = np.zeros(N, np.int64); nrej = np.zeros(N, np.int64)
    ip = ctypes.POINTER(ctypes.c_int32)
    lp = ctypes.POINTER(ctypes.c_int64)
    i64 = ctypes.c_int64
    rc = lib().orc_shoot_fan(
        _p(cin), _p(cpin), _p(rin), _p(zin), i64(len(rin)), i64(len(zin)), _p(depths),
        _p(depth_ranges), _p(bottom_angles), i64(len(depths)), _p(y0), i64(N),
        ctypes.c_double(source_range), ctypes.c_double(receiver_range), _p(r), i64(S),
        ctypes.c_double(rtol), ctypes.c_double(atol), ctypes.c_int(int(terminate_backwards)),
        i64(max_steps), _p(T), _p(Z), _p(P), _p(XI), nb.ctypes.data_as(ip), ns.ctypes.data_as(ip),
        st.ctypes.data_as(ip), nsteps.ctypes.data_as(lp), nfev.ctypes.data_as(lp),
        nrej.ctypes.data_as(lp))
    if rc:
        raise ValueError("bottom angle interpolant needs at least 4 bathymetry points")
    return dict(r=r, T=T, z=Z, p=P, xi=XI, n_bott=nb, n_surf=ns, status=st, n_steps=nsteps, nfev=nfev,
                n_rej=nrej)


def trace_ray(cin, cpin, rin, zin, depths, depth_ranges, bottom_angles, y0, source_range, receiver_range,
              rtol=1e-9, atol=1e-6, terminate_backwards=True, math=None, max_rows=200000):
    """Debugging aid: one row per step attempt of one ray -- columns t, h, y[3], f[3], error_norm,
    accepted, h_abs after the attempt, segment index."""
    cin, cpin, rin, zin = _c(cin), _c(cpin), _c(rin), _c(zin)
    depths, depth_ranges, bottom_angles = _c(depths), _c(depth_ranges), _c(bottom_angles)
    y0 = _c(y0).reshape(3)
    rows = np.zeros((max_rows, 12))
    i64 = ctypes.c_int64
    L = lib()
    L.orc_trace_ray.restype = i64
    _MATH_LOCK.acquire()
    old = get_math()
    if math is not None:
        set_math(math)
    try:
        n = L.orc_trace_ray(_p(cin), _p(cpin), _p(rin), _p(zin), i64(len(rin)), i64(len(zin)), _p(depths),
                            _p(depth_ranges), _p(bottom_angles), i64(len(depths)), _p(y0),
                            ctypes.c_double(source_range), ctypes.c_double(receiver_range),
                            ctypes.c_double(rtol), ctypes.c_double(atol), ctypes.c_int(int(terminate_backwards)),
                            _p(rows), i64(max_rows))
    finally:
        set_math(old)
        _MATH_LOCK.release()
    return rows[:max(int(n), 0)]
