#!/usr/bin/env python3
"""Hard cases for the correctly rounded powers of pygenray_amd/csrc/pgr_crmath.h (TEST INFRASTRUCTURE).

x ** -0.2 (the step controller, SCIPY/rk.py:156,162) and x ** 0.2 (select_initial_step, SCIPY/common.py:131), the
exponent being the DOUBLE 0.2: arguments whose exact power lies so close to a rounding boundary (the midpoint of two
neighbouring doubles) that the fast evaluation (good to 2^-74) cannot decide the rounding.  Search: random arguments
through the header's host twin (tests/crmath_host.c); those its rounding test sends to the second level (3e-6 of the
draws) are evaluated with mpmath at 400 bits, which gives the correctly rounded result and the distance of the exact
value from the boundary.  Written to tests/golden/g14_pow_hard_cases.npz:
  m02_x, m02_want, m02_log2_dist   every flagged argument of x ** -0.2 (log2 of the relative distance to the midpoint)
  p02_x, p02_want, p02_log2_dist   the same for x ** 0.2
tests/test_crmath.py (host twin) and tests/test_hip_parity.py (device) require bit-equality on all of them and at
least 24 cases closer than 2^-80.

usage: python scripts/gen/gen_pow_hard_cases.py [draws per function, default 6e9]"""
import ctypes, os, subprocess, sys, tempfile, time
import numpy as np
import mpmath as mp

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mp.mp.prec = 400
E02 = mp.mpf(0.2)          # the double 0.2, exactly


def build():
    so = os.path.join(tempfile.mkdtemp(), "libcrh.so")
    subprocess.check_call(["gcc", "-O2", "-std=gnu99", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-fopenmp",
                           "-o", so, os.path.join(ROOT, "tests", "crmath_host.c"), "-lm"])
    L = ctypes.CDLL(so)
    dp = ctypes.POINTER(ctypes.c_double)

    def ev(fn, a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        out = np.empty_like(a)
        L.crh_eval(ctypes.c_int(fn), a.ctypes.data_as(dp), out.ctypes.data_as(dp), ctypes.c_int64(a.size))
        return out
    return ev


def exact(x, sign):
    """(correctly rounded x ** (sign 0.2), log2 of the relative distance of the exact value from the nearest midpoint)"""
    v = mp.power(mp.mpf(float(x)), sign * E02)
    r = float(v)                                   # mpmath rounds to nearest even
    lo, hi = (np.nextafter(r, 0.0), r) if mp.mpf(r) > v else (r, np.nextafter(r, np.inf))
    mid = (mp.mpf(float(lo)) + mp.mpf(float(hi))) / 2
    d = abs(v - mid) / v
    return r, float(mp.log(d, 2)) if d > 0 else -np.inf


def search(ev, flag_fn, sign, draws, ranges, seed):
    rng = np.random.default_rng(seed)
    found = []
    t0 = time.time()
    batch = 8_000_000
    done = 0
    while done < draws:
        for lo, hi, logu in ranges:
            x = np.exp(rng.uniform(np.log(lo), np.log(hi), batch)) if logu else rng.uniform(lo, hi, batch)
            found.append(x[ev(flag_fn, x) != 0])
            done += batch
        if (done // batch) % 50 == 0:
            print(f"  {done:.2e} draws, {sum(len(f) for f in found)} flagged, {time.time() - t0:.0f} s", flush=True)
    xs = np.unique(np.concatenate(found))
    out = np.array([exact(x, sign) for x in xs])
    return xs, out[:, 0], out[:, 1]


if __name__ == "__main__":
    draws = float(sys.argv[1]) if len(sys.argv) > 1 else 6e9
    ev = build()
    # x ** -0.2: error norms -- everything the controller can see, and twice the weight on what it mostly sees
    mx, mw, md = search(ev, 5, -1, draws, [(1e-7, 1e4, True), (0.02, 1.2, False)], 14)
    print(f"x ** -0.2: {len(mx)} flagged, {int((md < -80).sum())} closer than 2^-80, closest 2^{md.min():.1f}")
    px, pw, pd = search(ev, 6, +1, draws / 4, [(1e-15, 1e13, True)], 15)
    print(f"x **  0.2: {len(px)} flagged, {int((pd < -80).sum())} closer than 2^-80, closest 2^{pd.min():.1f}")
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "g14_pow_hard_cases.npz"), m02_x=mx, m02_want=mw, m02_log2_dist=md,
                        p02_x=px, p02_want=pw, p02_log2_dist=pd)
    print("wrote tests/golden/g14_pow_hard_cases.npz")
