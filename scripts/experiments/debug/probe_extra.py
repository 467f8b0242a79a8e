import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import numpy as np
L = ctypes.CDLL(os.path.join(ROOT, "pygenray_amd/csrc/libpgr_probe.so"))
rng = np.random.default_rng(0)
M = 2_000_000
a = rng.uniform(-1e4, 1e4, M) * 10.0 ** rng.integers(-8, 8, M)
b = rng.uniform(0.1, 10, M) * 10.0 ** rng.integers(-12, 12, M)
out = np.empty((M, 12))
dp = ctypes.POINTER(ctypes.c_double)
rc = L.pgr_debug_math(a.ctypes.data_as(dp), b.ctypes.data_as(dp), ctypes.c_int64(M), out.ctypes.data_as(dp))
assert rc == 0
print("raw rcp max rel err", np.abs(out[:, 6] * b - 1).max())
print("div with 1 NR: mismatch frac", np.mean(out[:, 7] != a / b), "max ulp", (np.abs(out[:, 7] - a / b) / np.spacing(np.abs(a / b))).max())
print("raw rsq max rel err", np.abs(out[:, 8] * np.sqrt(b) - 1).max())
print("sqrt with 1 NR: mismatch frac", np.mean(out[:, 9] != np.sqrt(b)))
print("rcp 1NR+corr: mismatch frac", np.mean(out[:, 10] != 1 / b))
