"""Device build of the correctly rounded functions (csrc/pgr_crmath.h) against the oracle's
libquadmath values: mismatches out of N."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, oracle
from pygenray_amd import _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
rng = np.random.default_rng(5)
b = np.exp(rng.uniform(np.log(1e-7), np.log(1e4), N))
a = np.concatenate([rng.uniform(-1, 1, N // 2), rng.uniform(-1, 1, N // 4) ** 5,
                    np.sign(rng.uniform(-1, 1, N - N // 2 - N // 4)) * (1 - 10 ** rng.uniform(-16, -0.3, N - N // 2 - N // 4))])
o = _lib.debug_math(a, b)
for col, name, arg in ((4, "pow_m02", b), (6, "pow_p02", b), (7, "asin", a), (8, "sin", a)):
    ref = oracle.math_fn(name, arg, math=oracle.MATH_CR)
    lm = oracle.math_fn(name, arg, math=oracle.MATH_LIBM)
    print(f"{name}: device != correctly rounded: {int(np.sum(o[:, col] != ref))} of {N}   (glibc != correctly rounded: {int(np.sum(lm != ref))})")
a2 = rng.uniform(-6.5, 6.5, N)
o2 = _lib.debug_math(a2, b)
print("sin on [-6.5, 6.5]: device != correctly rounded:", int(np.sum(o2[:, 8] != oracle.math_fn("sin", a2))), "of", N)
