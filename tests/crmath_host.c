/* Host twin of pygenray_amd/csrc/pgr_crmath.h -- TEST INFRASTRUCTURE ONLY (tests/test_crmath.py).
 * The same source text the device compiles, built with gcc so that the correctly rounded
 * pow / asin / sin can be checked against libquadmath on millions of arguments without a GPU.
 * Nothing in the product loads this. */
#define PGR_CR_HOST 1
#include <stdint.h>
#include "../pygenray_amd/csrc/pgr_crmath.h"

/* fn 0: a ** -0.2, 1: a ** 0.2, 2: asin, 3: sin; 4: the reflection law's sin(radians(-degrees(asin a))) through
 * pgr_cr_sin_near_minus_asin (np.degrees(x) = x * (180/pi), np.radians(x) = x * (pi/180)) */
static double reflect_sin(double v)
{
    const struct pgr_dd A = pgr_cr_asin_dd(v);
    const double theta = A.h * (180.0 / M_PI);
    const double x = (-theta) * (M_PI / 180.0);
    return pgr_cr_sin_near_minus_asin(x, v, A);
}
/* fn 5 / 6: does the rounding test of a ** -0.2 / a ** 0.2 send the argument to the second level (1.0 / 0.0)?
 * fn 7 / 8: the second level alone (every argument through pgr_cr_pow_*_slow / the double-precision logarithm);
 * fn 9: pgr_cr_log2_slow */
static double ziv_flag_m02(double x)
{
    const float lf = pgr_cr_seed_log2f((float)x);
    double y = pgr_cr_newton5(x, (double)pgr_cr_seed_exp2f(-0.2f * lf), 0.2);
    const double ah = y * y, al = __builtin_fma(y, y, -ah);
    const double bh = ah * ah, bl = __builtin_fma(ah, al + al, __builtin_fma(ah, ah, -bh));
    const double ch = bh * y, cl = __builtin_fma(bl, y, __builtin_fma(bh, y, -ch));
    const double dh = ch * x, dl = __builtin_fma(cl, x, __builtin_fma(ch, x, -dh));
    const double rho = (1.0 - dh) - dl, r5 = rho * 0.2;
    const double u = __builtin_fma((double)lf, PGR_CR_POW_KLN2, __builtin_fma(-2.0 * r5, r5, r5));
    const double r = __builtin_fma(y, u, y);
    return (double)pgr_cr_rounding_uncertain(__builtin_fma(y, u, -(r - y)), r);
}
static double slow_m02(double x)
{
    const float lf = pgr_cr_seed_log2f((float)x);
    return pgr_cr_pow_m02_slow(x, pgr_cr_newton5(x, (double)pgr_cr_seed_exp2f(-0.2f * lf), 0.2), 0.2, PGR_CR_POW_KLN2);
}
static double p02_parts(double x, int slow, int flag)
{
    const float lf = pgr_cr_seed_log2f((float)x);
    double w = (double)pgr_cr_seed_exp2f(-0.2f * lf), approx, corr;
    w = pgr_cr_newton5(x, pgr_cr_newton5(x, w, 0.2), 0.2);
    const double r = pgr_cr_pow_p02_eval(x, w, slow ? pgr_cr_log2_slow(x) : (double)lf, &approx, &corr);
    return flag ? (double)pgr_cr_rounding_uncertain((approx - r) + corr, r) : r;
}
void crh_eval(int fn, const double *a, double *out, int64_t n)
{
#pragma omp parallel for
    for (int64_t k = 0; k < n; k++)
        out[k] = fn == 0 ? pgr_cr_pow_m02(a[k]) : fn == 1 ? pgr_cr_pow_p02(a[k]) : fn == 2 ? pgr_cr_asin(a[k]) : fn == 3 ? pgr_cr_sin(a[k])
               : fn == 4 ? reflect_sin(a[k]) : fn == 5 ? ziv_flag_m02(a[k]) : fn == 6 ? p02_parts(a[k], 0, 1) : fn == 7 ? slow_m02(a[k])
               : fn == 8 ? p02_parts(a[k], 1, 0) : pgr_cr_log2_slow(a[k]);
}
