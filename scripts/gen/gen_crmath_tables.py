"""Generates the double-double constants of pygenray_amd/csrc/pgr_crmath.h with mpmath
(sin/cos(j/32) for j = 0..28, pi/2 in two and three pieces, series coefficients).
usage: python scripts/gen/gen_crmath_tables.py > /tmp/tables.inc   (pasted into the header)"""
import mpmath as mp

mp.mp.prec = 400


def dd(x):
    h = float(x)
    l = float(x - mp.mpf(h))
    return h, l


def hx(v):
    return float(v).hex()


print("// sin(j/32), cos(j/32), j = 0..28, as double-double {hi, lo}")
print("PGR_CR_TABLE(double, pgr_cr_sincos_tab, 29 * 4) = {")
for j in range(29):
    s = dd(mp.sin(mp.mpf(j) / 32))
    c = dd(mp.cos(mp.mpf(j) / 32))
    print(f"    {hx(s[0])}, {hx(s[1])}, {hx(c[0])}, {hx(c[1])},")
print("};")
pi2 = mp.pi / 2
h, l = dd(pi2)
print(f"#define PGR_CR_PIO2_H {hx(h)}\n#define PGR_CR_PIO2_L {hx(l)}")
# Cody-Waite: P1 = pi/2 rounded to 33 bits (k * P1 exact for |k| < 2^20), P2 next 53 bits, P3 the rest
import math
p1 = mp.mpf(math.ldexp(round(float(pi2) * 2 ** 32), -32))
p2 = mp.mpf(float(pi2 - p1))
p3 = mp.mpf(float(pi2 - p1 - p2))
print(f"#define PGR_CR_PIO2_1 {hx(p1)}\n#define PGR_CR_PIO2_2 {hx(p2)}\n#define PGR_CR_PIO2_3 {hx(p3)}")
print(f"#define PGR_CR_2OPI {hx(2 / mp.pi)}")
for name, val in (("S3", -mp.mpf(1) / 6), ("S5", mp.mpf(1) / 120), ("S7", -mp.mpf(1) / 5040),
                  ("C4", mp.mpf(1) / 24), ("C6", -mp.mpf(1) / 720)):
    h, l = dd(val)
    print(f"#define PGR_CR_{name}_H {hx(h)}\n#define PGR_CR_{name}_L {hx(l)}")
for name, val in (("S9", mp.mpf(1) / 362880), ("S11", -mp.mpf(1) / 39916800), ("C8", mp.mpf(1) / 40320),
                  ("C10", -mp.mpf(1) / 3628800), ("C12", mp.mpf(1) / 479001600)):
    print(f"#define PGR_CR_{name} {hx(val)}")
# x ** y with y = the double nearest to -+0.2: y = -+(1/5)(1 + kappa)
y = mp.mpf(0.2)
kappa = y * 5 - 1
print(f"// 0.2 (double) = (1/5)(1 + kappa), kappa = {mp.nstr(kappa, 20)}")
print(f"#define PGR_CR_POW_KLN2 {hx(-(kappa / 5) * mp.log(2))}   /* -(kappa/5) ln 2: x**-0.2 = x^(-1/5) (1 + this * log2 x) */")
