import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pygenray_amd import _lib
rng = np.random.default_rng(0)
M = 2_000_000
a = rng.uniform(-1e4, 1e4, M) * 10.0 ** rng.integers(-8, 8, M)
b = rng.uniform(0.1, 10, M) * 10.0 ** rng.integers(-12, 12, M)
b[:500000] = 10 ** rng.uniform(np.log10(5e-6), np.log10(1800), 500000)   # pow range
o = _lib.debug_math(a, b)
def ulps(x, ref): return np.abs(x - ref) / np.spacing(np.abs(ref))
print("div   max ulp", ulps(o[:, 0], a / b).max(), "mismatch frac", np.mean(o[:, 0] != a / b))
print("rcp   max ulp", ulps(o[:, 1], 1 / b).max(), "mismatch frac", np.mean(o[:, 1] != 1 / b))
print("rsqrt max ulp", ulps(o[:, 2], 1 / np.sqrt(b)).max())
print("sqrt  max ulp", ulps(o[:, 3], np.sqrt(b)).max(), "mismatch frac", np.mean(o[:, 3] != np.sqrt(b)))
pb = b[:500000]
ref = np.power(pb.astype(np.longdouble), np.longdouble(-0.2)).astype(np.float64)
print("pow   max ulp", ulps(o[:500000, 4], ref).max(), "max rel", (np.abs(o[:500000, 4] - ref) / ref).max())
print("minstep eq", np.array_equal(o[:, 5], 10 * np.abs(np.nextafter(a, np.inf) - a)))
