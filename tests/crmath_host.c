/* Host twin of pygenray_amd/csrc/pgr_crmath.h -- TEST INFRASTRUCTURE ONLY (tests/test_crmath.py).
 * The same source text the device compiles, built with gcc so that the correctly rounded
 * pow / asin / sin can be checked against libquadmath on millions of arguments without a GPU.
 * Nothing in the product loads this. */
#define PGR_CR_HOST 1
#include <stdint.h>
#include "../pygenray_amd/csrc/pgr_crmath.h"

/* fn 0: a ** -0.2, 1: a ** 0.2, 2: asin, 3: sin */
void crh_eval(int fn, const double *a, double *out, int64_t n)
{
#pragma omp parallel for
    for (int64_t k = 0; k < n; k++)
        out[k] = fn == 0 ? pgr_cr_pow_m02(a[k]) : fn == 1 ? pgr_cr_pow_p02(a[k]) : fn == 2 ? pgr_cr_asin(a[k]) : pgr_cr_sin(a[k]);
}
