"""The correctly rounded pow / asin / sin of the HIP path (pygenray_amd/csrc/pgr_crmath.h), checked
WITHOUT a GPU: the header's host twin (tests/crmath_host.c, built here with gcc) against the oracle's
binary128 evaluation (libquadmath) rounded once.  The device build of the same text is checked
against the same reference in tests/test_hip_parity.py::test_arithmetic_building_blocks."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

import oracle

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def crh(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("crmath") / "libcrmath_host.so")
    # CRMATH_SANITIZE=1 (tests/test_sanitizers.py, in a python started with libasan preloaded): the same text under
    # AddressSanitizer + UBSan
    san = (["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer"]
           if os.environ.get("CRMATH_SANITIZE") == "1" else ["-O2"])
    subprocess.check_call(["gcc"] + san + ["-std=gnu99", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math",
                           "-fopenmp", "-Wno-unknown-pragmas", "-o", so, os.path.join(HERE, "crmath_host.c"), "-lm"])
    L = ctypes.CDLL(so)
    dp = ctypes.POINTER(ctypes.c_double)

    def ev(fn, a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        out = np.empty_like(a)
        L.crh_eval(ctypes.c_int(fn), a.ctypes.data_as(dp), out.ctypes.data_as(dp), ctypes.c_int64(a.size))
        return out
    return ev


N = int(os.environ.get("CRMATH_N", 2_000_000))


def test_pow_of_the_step_controller_is_correctly_rounded(crh):
    rng = np.random.default_rng(0)
    x = np.exp(rng.uniform(np.log(1e-7), np.log(1e4), N))          # error norms (SCIPY/rk.py:156,162)
    assert np.array_equal(crh(0, x), oracle.math_fn("pow_m02", x))
    # the error norms a step controller actually produces crowd around 0.1 ... 1
    x = rng.uniform(0.02, 1.2, N)
    assert np.array_equal(crh(0, x), oracle.math_fn("pow_m02", x))
    # what falls outside the fp32 range of the seed is decided by SciPy's min / max anyway
    assert not np.isfinite(crh(0, np.array([0.0])))[0] or crh(0, np.array([0.0]))[0] > 10
    # select_initial_step's (0.01 / max(d1, d2)) ** 0.2: d > 1e-15, so the argument stays below 1e13
    v = np.exp(rng.uniform(np.log(1e-15), np.log(1e13), N))
    assert np.array_equal(crh(1, v), oracle.math_fn("pow_p02", v))


def test_asin_and_sin_are_correctly_rounded(crh):
    rng = np.random.default_rng(1)
    v = np.concatenate([rng.uniform(-1, 1, N // 2), rng.uniform(-1, 1, N // 4) ** 5,            # p c of any ray
                        np.sign(rng.uniform(-1, 1, N // 4)) * (1 - 10 ** rng.uniform(-16, -0.3, N // 4)),  # near vertical
                        [0.0, -0.0, 1.0, -1.0, 0.75, -0.75, 0.5, 2.0 ** -30, 5e-324]])
    assert np.array_equal(crh(2, v), oracle.math_fn("asin", v))
    assert np.all(np.signbit(crh(2, np.array([-0.0, 0.0]))) == [True, False])
    assert np.all(np.isnan(crh(2, np.array([1.0000000000000002, -3.0, np.nan]))))             # Q7
    w = np.concatenate([rng.uniform(-6.5, 6.5, N // 2), rng.uniform(-1.6, 1.6, N // 4), rng.uniform(-1, 1, N // 4) ** 7,
                        np.pi / 2 * np.arange(-4, 5), np.pi / 4 * np.arange(-8, 9), [0.0, -0.0, 1e-300]])
    assert np.array_equal(crh(3, w), oracle.math_fn("sin", w))


def test_reflection_sine_short_form_equals_the_general_sine(crh):
    """The reflection law at the surface and on a flat sea floor, sin(radians(-degrees(arcsin v)))
    (REF/launch_rays.py:459-480), takes the sine from the arcsine's double-double value and a first-order term
    (pgr_cr_sin_near_minus_asin): the same double as the correctly rounded sine of that argument."""
    rng = np.random.default_rng(3)
    v = np.concatenate([rng.uniform(-1, 1, N // 2), rng.uniform(-1, 1, N // 4) ** 5,
                        np.sign(rng.uniform(-1, 1, N // 4)) * (1 - 10 ** rng.uniform(-16, -0.3, N // 4)),
                        0.999 + np.arange(-2000, 2001) * 2.0 ** -53, [0.0, 1.0, -1.0, 0.75, -0.75, 1e-300]])
    theta = oracle.math_fn("asin", v) * (180.0 / np.pi)
    x = (-theta) * (np.pi / 180.0)
    assert np.array_equal(crh(4, v), oracle.math_fn("sin", x))


def test_hard_neighbourhoods(crh):
    """Where correct rounding is hardest: the doubles next to arguments whose result is exactly representable
    (x = 32^n for x ** -0.2 and x ** 0.2; v = 0, +-1/2 (asin = pi/6 is not exact, but the branch structure changes
    near 0.75), +-1 for asin), next to the table nodes j/32 of the double-double sine / cosine and next to the
    multiples of pi/2 of the sine's argument reduction."""
    k = np.arange(-30_000, 30_001, dtype=np.float64)
    for n in (-4, -3, -2, -1, 0, 1, 2, 3):                         # 32^n (1 + k ulp): results next to 2^-n ...
        x = 32.0 ** n * (1.0 + k * 2.0 ** -52)
        assert np.array_equal(crh(0, x), oracle.math_fn("pow_m02", x)), n
        assert np.array_equal(crh(1, x), oracle.math_fn("pow_p02", x)), n
        # ... and EVERY double on the lower side, where they are twice as dense (the rounding test of the Ziv evaluation does
        # not look at the finer spacing below a result that is a power of two: pgr_cr_rounding_uncertain)
        x = 32.0 ** n * (1.0 - np.arange(0, 60_001) * 2.0 ** -53)
        assert np.array_equal(crh(0, x), oracle.math_fn("pow_m02", x)), n
        assert np.array_equal(crh(1, x), oracle.math_fn("pow_p02", x)), n
    for c in (0.0, 0.5, 0.75, 0.7499999, 0.9, 2.0 ** -27):
        v = np.concatenate([c + k * 2.0 ** -53 * max(c, 2.0 ** -60), -(c + k * 2.0 ** -53 * max(c, 2.0 ** -60))])
        v = v[np.abs(v) <= 1]
        assert np.array_equal(crh(2, v), oracle.math_fn("asin", v)), c
    v = np.concatenate([1.0 - np.arange(0, 60_001) * 2.0 ** -53, -(1.0 - np.arange(0, 60_001) * 2.0 ** -53)])
    assert np.array_equal(crh(2, v), oracle.math_fn("asin", v))
    for j in (1, 7, 16, 25, 28, 29, 50):                          # sine: table nodes j/32 (j <= 28 tabulated) ...
        w = j / 32.0 * (1.0 + k * 2.0 ** -52)
        assert np.array_equal(crh(3, np.concatenate([w, -w])), oracle.math_fn("sin", np.concatenate([w, -w]))), j
    for m in (1, 2, 3, 4):                                        # ... and the multiples of pi/2
        w = m * (np.pi / 2) * (1.0 + k * 2.0 ** -52)
        assert np.array_equal(crh(3, np.concatenate([w, -w])), oracle.math_fn("sin", np.concatenate([w, -w]))), m


def test_powers_rounding_test_and_second_level(crh):
    """The Ziv evaluation of the two powers (pgr_crmath.h): (i) the second level ALONE (logarithm in double precision,
    rounding-limited root) is correctly rounded on every random argument, not only on the rare ones that reach it;
    (ii) the rounding test sends a few in a million arguments there; (iii) EVERY argument it flagged in 6e9 (x ** -0.2) and
    1.5e9 (x ** 0.2) random draws (tests/golden/g14_pow_hard_cases.npz, expected values from mpmath at 400 bits,
    scripts/gen/gen_pow_hard_cases.py) comes out correctly rounded -- among them the cases whose exact value lies within
    2^-80 of a rounding boundary, which the fast evaluation (good to 2^-74) cannot decide; (iv) the fast evaluation alone
    (-DPGR_POW_NO_ZIV) does get some of the flagged ones wrong: the test is not vacuous."""
    from helpers import load
    rng = np.random.default_rng(4)
    x = np.exp(rng.uniform(np.log(1e-7), np.log(1e4), N))
    v = np.exp(rng.uniform(np.log(1e-15), np.log(1e13), N))
    assert np.array_equal(crh(7, x), oracle.math_fn("pow_m02", x)) and np.array_equal(crh(8, v), oracle.math_fn("pow_p02", v))
    assert np.abs(crh(9, v) - np.log2(v)).max() < 2.0 ** -45
    for fn, arg in ((5, x), (6, v)):
        frac = crh(fn, arg).mean()
        assert frac < 3e-5 and (frac > 2e-7 or N < 1_000_000), (fn, frac)     # (expected 3e-6; the sanitizer run draws 2e5)
    g = load("g14_pow_hard_cases.npz")
    for tag, fn, flag_fn, slow_fn in (("m02", 0, 5, 7), ("p02", 1, 6, 8)):
        xs, want, dist = g[tag + "_x"], g[tag + "_want"], g[tag + "_log2_dist"]
        assert len(xs) > 4000 and np.all(crh(flag_fn, xs) == 1.0)          # all of them take the second level
        assert (dist < -80).sum() >= (24 if tag == "m02" else 6), (tag, int((dist < -80).sum()))
        assert np.array_equal(crh(fn, xs), want), tag
        assert np.array_equal(crh(slow_fn, xs), want), tag
        assert np.array_equal(oracle.math_fn("pow_" + tag, xs), want), tag    # the oracle's binary128 agrees with mpmath


def test_fast_power_alone_misrounds_some_flagged_arguments(tmp_path):
    """(iv) above: the header built with -DPGR_POW_NO_ZIV (no rounding test) differs from the correctly rounded value on
    some of the flagged arguments -- every one of them within 2^-74 of a boundary."""
    from helpers import load
    so = str(tmp_path / "libcrmath_noziv.so")
    subprocess.check_call(["gcc", "-O2", "-std=gnu99", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-fopenmp",
                           "-DPGR_POW_NO_ZIV", "-o", so, os.path.join(HERE, "crmath_host.c"), "-lm"])
    L = ctypes.CDLL(so)
    dp = ctypes.POINTER(ctypes.c_double)
    g = load("g14_pow_hard_cases.npz")
    xs = np.ascontiguousarray(g["m02_x"])
    out = np.empty_like(xs)
    L.crh_eval(ctypes.c_int(0), xs.ctypes.data_as(dp), out.ctypes.data_as(dp), ctypes.c_int64(xs.size))
    wrong = out != g["m02_want"]
    assert 0 < wrong.sum() < 0.2 * len(xs)
    assert g["m02_log2_dist"][wrong].max() < -74 and np.all(np.abs(out[wrong] - g["m02_want"][wrong]) <= np.spacing(g["m02_want"][wrong]))


def test_the_platform_libm_is_faithful_but_not_correctly_rounded():
    """Why bit-identity needs the correctly rounded functions on BOTH sides: glibc's pow / asin /
    sin (what NumPy and SciPy call in this container, and the oracle's default MATH_LIBM mode)
    differ from the correctly rounded value in about one call in a thousand, by one ulp."""
    rng = np.random.default_rng(2)
    x = np.exp(rng.uniform(np.log(1e-6), np.log(2e3), 400_000))
    v = rng.uniform(-0.9, 0.9, 400_000)
    for name, arg in (("pow_m02", x), ("asin", v), ("sin", v * 1.7)):
        cr = oracle.math_fn(name, arg, math=oracle.MATH_CR)
        lm = oracle.math_fn(name, arg, math=oracle.MATH_LIBM)
        diff = cr != lm
        assert 1e-5 < diff.mean() < 1e-2, (name, diff.mean())
        assert np.all(np.abs(cr[diff] - lm[diff]) <= np.spacing(np.abs(cr[diff])))
    # ... which moves few rays, since a step's h = (t + h_abs) - t is quantised to ulp(t): the two
    # modes integrate most rays identically and the rest within the 1-ulp noise class
    import helpers
    arrs = helpers.munk_arrays(300e3)
    y0 = helpers.y0_for(oracle, arrs, 1000.0, 0.0, np.linspace(-20, 20, 96))
    a = oracle.shoot_fan(*arrs, y0, 0.0, 300e3, 31, math=oracle.MATH_LIBM)
    b = oracle.shoot_fan(*arrs, y0, 0.0, 300e3, 31, math=oracle.MATH_CR)
    assert np.array_equal(a["status"], b["status"]) and np.array_equal(a["n_bott"], b["n_bott"])
    assert np.mean(a["n_steps"] == b["n_steps"]) > 0.9
    d = np.abs(a["z"][:, -1] - b["z"][:, -1]) / 5000.0
    assert np.median(d) < 1e-9 and d.max() < 1e-5
