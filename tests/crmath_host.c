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
void crh_eval(int fn, const double *a, double *out, int64_t n)
{
#pragma omp parallel for
    for (int64_t k = 0; k < n; k++)
        out[k] = fn == 0 ? pgr_cr_pow_m02(a[k]) : fn == 1 ? pgr_cr_pow_p02(a[k]) : fn == 2 ? pgr_cr_asin(a[k]) : fn == 3 ? pgr_cr_sin(a[k]) : reflect_sin(a[k]);
}
