/*
 * ray_oracle.c -- CPU restatement (plain C99, fp64, strictly IEEE: build with
 * -ffp-contract=off, no fast-math) of pygenray's per-ray integrator.
 *
 * >>> TEST INFRASTRUCTURE ONLY. <<<
 * This file is the parity ORACLE for the HIP path.  Only tests/, the smoke
 * check in __graft_entry__.py and bench.py's `cpu_baseline` leg may load it.
 * Nothing under pygenray_amd/ links, loads or calls it; the product path
 * fails loudly when its HIP library is missing.
 *
 * Parity pin: checked in tests/test_oracle_*.py against
 *   - the reference's own committed fixture tests/fixtures/munk_regression.npz
 *     (copied as DATA to tests/golden/ref_munk_regression.npz), and
 *   - golden vectors produced by running the reference itself in the build
 *     container (tests/golden/make_golden.py -> tests/golden/g*.npz).
 *
 * Each function cites the reference lines it restates.  REF = /root/reference/
 * src/pygenray, SCIPY = scipy/integrate/_ivp (SciPy 1.15.3, the unpinned
 * third-party dependency in which the integrator arithmetic lives; call site
 * REF/launch_rays.py:670-679).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#include <quadmath.h>

/*
 * The three libm calls on the path -- error_norm ** -0.2 and (0.01 / d) ** 0.2 (SCIPY/rk.py:156,162,
 * common.py:131), arcsin in ray_angle (REF/integration_processes.py:333) and sin in the reflection
 * law (REF/launch_rays.py:480) -- are the only operations of the reference whose result is not
 * fixed by IEEE 754: NumPy hands them to the platform libm, which is faithful (< 1 ulp) but not
 * correctly rounded, so the reference itself is not bit-reproducible across libm builds.  Two modes:
 *   ORC_MATH_LIBM (default)  the platform libm, i.e. what NumPy/SciPy call in this container: the
 *                            mode the golden vectors pin (step counts equal SciPy's exactly);
 *   ORC_MATH_CR              the same three functions CORRECTLY ROUNDED (evaluated in binary128 by
 *                            libquadmath, then rounded once to binary64): the platform-independent
 *                            idealisation every libm approximates.  glibc 2.35 differs from it by
 *                            1 ulp in 0.0x - 0.x % of the calls (tests/test_oracle_golden.py measures
 *                            it).  The HIP path implements the same correctly rounded functions
 *                            with its own double-double algorithms, so HIP and oracle agree BIT FOR
 *                            BIT in this mode; tests/test_hip_parity.py checks exactly that.
 */
#define ORC_MATH_LIBM 0
#define ORC_MATH_CR 1
static int g_math = ORC_MATH_LIBM;
void orc_set_math(int mode) { g_math = (mode == ORC_MATH_CR) ? ORC_MATH_CR : ORC_MATH_LIBM; }
int orc_get_math(void) { return g_math; }
double orc_pow(double x, double y)
{
    return g_math == ORC_MATH_CR ? (double)powq((__float128)x, (__float128)y) : pow(x, y);
}
double orc_asin(double x)
{
    return g_math == ORC_MATH_CR ? (double)asinq((__float128)x) : asin(x);
}
double orc_sin(double x)
{
    return g_math == ORC_MATH_CR ? (double)sinq((__float128)x) : sin(x);
}
/* test hook: out[k] = f(a[k]) in the current mode; fn 0: a ** -0.2, 1: a ** 0.2, 2: asin, 3: sin */
void orc_math_array(int fn, const double *a, double *out, int64_t n)
{
#pragma omp parallel for
    for (int64_t k = 0; k < n; k++)
        out[k] = fn == 0 ? orc_pow(a[k], -0.2) : fn == 1 ? orc_pow(a[k], 0.2) : fn == 2 ? orc_asin(a[k]) : orc_sin(a[k]);
}

#define ORC_OK 0
#define ORC_VERTICAL 1      /* REF/launch_rays.py:443-448 */
#define ORC_BBOX 2          /* REF/launch_rays.py:451-456 */
#define ORC_BACKWARD 3      /* REF/launch_rays.py:474-477 */
#define ORC_STEP_TOO_SMALL 4 /* SCIPY/base.py:129, rk.py:132-133 -> REF/launch_rays.py:427-430 */
#define ORC_MAX_STEPS 5     /* guard only (no reference counterpart) */
#define ORC_BETA_RANGE 6    /* interp1d bounds_error ValueError, REF/launch_rays.py:397-399,469 */
#define ORC_EVENT_ERROR 7   /* brentq "f(a) and f(b) must have different signs" ValueError */

typedef struct {
    const double *cin, *cpin; /* [nr][nz] range-major, depth contiguous */
    const double *rin, *zin;
    int64_t nr, nz;
    const double *depths, *depth_ranges, *bottom_angles;
    int64_t nb;
    double *pp; /* not-a-knot cubic of bottom_angles: 4 coeffs per interval */
} orc_env;

/* ---- np.searchsorted(grid, x) (side='left'): REF/integration_processes.py:152-153 */
static int64_t searchsorted_left(const double *g, int64_t n, double x)
{
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        int64_t mid = lo + ((hi - lo) >> 1);
        if (g[mid] < x) lo = mid + 1; else hi = mid;
    }
    return lo;
}

/* ---- REF/integration_processes.py:101-174 (index clamped, weight NOT clamped: Q4) */
double orc_bilinear(double x, double y, const double *xg, int64_t nx, const double *yg, int64_t ny,
                    const double *v)
{
    int64_t i = searchsorted_left(xg, nx, x) - 1;
    int64_t j = searchsorted_left(yg, ny, y) - 1;
    if (i > nx - 2) i = nx - 2;
    if (i < 0) i = 0;
    if (j > ny - 2) j = ny - 2;
    if (j < 0) j = 0;
    double wx = (x - xg[i]) / (xg[i + 1] - xg[i]);
    double wy = (y - yg[j]) / (yg[j + 1] - yg[j]);
    double v00 = v[i * ny + j], v10 = v[(i + 1) * ny + j];
    double v01 = v[i * ny + j + 1], v11 = v[(i + 1) * ny + j + 1];
    return (1 - wx) * (1 - wy) * v00 + wx * (1 - wy) * v10 + (1 - wx) * wy * v01 + wx * wy * v11;
}

/* ---- REF/integration_processes.py:177-235 */
double orc_linear(double x, const double *xin, const double *yin, int64_t n)
{
    int64_t i = searchsorted_left(xin, n, x) - 1;
    if (i > n - 2) i = n - 2;
    if (i < 0) i = 0;
    double w = (x - xin[i]) / (xin[i + 1] - xin[i]);
    return (1 - w) * yin[i] + w * yin[i + 1];
}

/* ---- REF/integration_processes.py:26-98 */
static void derivsrd(const orc_env *e, double x, const double *y, double *dydx)
{
    double z = y[1], pz = y[2];
    double c = orc_bilinear(x, z, e->rin, e->nr, e->zin, e->nz, e->cin);
    double cp = orc_bilinear(x, z, e->rin, e->nr, e->zin, e->nz, e->cpin);
    double arg = 1.0 - (c * c) * (pz * pz);
    if (arg <= 0.0) arg = 1e-30; /* Q8 */
    double fact = 1 / sqrt(arg);
    dydx[0] = fact / c;
    dydx[1] = c * pz * fact;
    dydx[2] = -fact * cp / (c * c);
}

/* ---- REF/integration_processes.py:306-334; np.degrees(x) = x*(180/pi) */
static double ray_angle(const orc_env *e, double x, const double *y, double *c_out)
{
    double c = orc_bilinear(x, y[1], e->rin, e->nr, e->zin, e->nz, e->cin);
    if (c_out) *c_out = c;
    return orc_asin(y[2] * c) * (180.0 / M_PI);
}

/* ---- the four +-1 event functions, REF/integration_processes.py:238-303 (Q6, Q7) */
static double ev_surface(const orc_env *e, double x, const double *y)
{
    double th = ray_angle(e, x, y, 0);
    return ((y[1] < 0) && (th < 0)) ? 1.0 : -1.0;
}
static double ev_bottom(const orc_env *e, double x, const double *y)
{
    double bd = orc_linear(x, e->depth_ranges, e->depths, e->nb);
    double th = ray_angle(e, x, y, 0);
    return ((y[1] > bd) && (th > 0)) ? 1.0 : -1.0;
}
static double ev_vertical(const orc_env *e, double x, const double *y)
{
    double th = ray_angle(e, x, y, 0);
    return (fabs(th) > (90 - 1e-3)) ? 1.0 : -1.0;
}
static double ev_bbox(const orc_env *e, double x, const double *y)
{
    double z = y[1], tol = 1e-6;
    int b = (z > e->zin[e->nz - 1] + tol) | (z < e->zin[0] - tol) | (x < e->rin[0] - tol) |
            (x > e->rin[e->nr - 1] + tol);
    return b ? 1.0 : -1.0;
}
typedef double (*event_fn)(const orc_env *, double, const double *);
static const event_fn EVENTS[4] = {ev_surface, ev_bottom, ev_vertical, ev_bbox};
static const int EV_DIR[4] = {1, 1, 0, 0}; /* REF/launch_rays.py:649-661 */

/* ---- Dormand-Prince tableau, SCIPY/rk.py:377-404 */
static const double RK_C[6] = {0, 1.0 / 5, 3.0 / 10, 4.0 / 5, 8.0 / 9, 1};
static const double RK_A[6][5] = {
    {0, 0, 0, 0, 0},
    {1.0 / 5, 0, 0, 0, 0},
    {3.0 / 40, 9.0 / 40, 0, 0, 0},
    {44.0 / 45, -56.0 / 15, 32.0 / 9, 0, 0},
    {19372.0 / 6561, -25360.0 / 2187, 64448.0 / 6561, -212.0 / 729, 0},
    {9017.0 / 3168, -355.0 / 33, 46732.0 / 5247, 49.0 / 176, -5103.0 / 18656}};
static const double RK_B[6] = {35.0 / 384, 0, 500.0 / 1113, 125.0 / 192, -2187.0 / 6784, 11.0 / 84};
static const double RK_E[7] = {-71.0 / 57600, 0, 71.0 / 16695, -71.0 / 1920, 17253.0 / 339200,
                               -22.0 / 525, 1.0 / 40};
static const double RK_P[7][4] = {
    {1, -8048581381.0 / 2820520608, 8663915743.0 / 2820520608, -12715105075.0 / 11282082432},
    {0, 0, 0, 0},
    {0, 131558114200.0 / 32700410799, -68118460800.0 / 10900136933, 87487479700.0 / 32700410799},
    {0, -1754552775.0 / 470086768, 14199869525.0 / 1410260304, -10690763975.0 / 1880347072},
    {0, 127303824393.0 / 49829197408, -318862633887.0 / 49829197408,
     701980252875.0 / 199316789632},
    {0, -282668133.0 / 205662961, 2019193451.0 / 616988883, -1453857185.0 / 822651844},
    {0, 40617522.0 / 29380423, -110615467.0 / 29380423, 69997945.0 / 29380423}};

/* np.linalg.norm(x)/x.size**0.5 for 3 elements, SCIPY/common.py:63-65 */
static double rms3(const double *v)
{
    return sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]) / 1.7320508075688772;
}

/* one dense-output piece, SCIPY/rk.py:178-180,552-574 */
typedef struct {
    double t_old, t, h, y_old[3], Q[3][4];
} dense_t;

static void dense_eval(const dense_t *d, double t, double *y)
{
    double x = (t - d->t_old) / d->h;
    double p[4];
    p[0] = x;
    p[1] = p[0] * x;
    p[2] = p[1] * x;
    p[3] = p[2] * x; /* np.cumprod */
    for (int i = 0; i < 3; i++) {
        double s = 0.0;
        for (int j = 0; j < 4; j++) s += d->Q[i][j] * p[j];
        y[i] = d->h * s + d->y_old[i];
    }
}

typedef struct {
    const orc_env *e;
    event_fn f;
    const dense_t *d;
} evroot_ctx;
static double evroot_f(const evroot_ctx *c, double t)
{
    double y[3];
    dense_eval(c->d, t, y);
    return c->f(c->e, t, y);
}

/* scipy.optimize.brentq (Zeros/brentq.c) as called from SCIPY/ivp.py:51-76 with
 * xtol = rtol = 4*EPS, maxiter = 100.  On +-1 step functions it degenerates to
 * bisection (Q6).  Transcription validated against scipy in tests. */
static double brentq(const evroot_ctx *c, double xa, double xb, double xtol, double rtol, int iter,
                     int *err)
{
    double xpre = xa, xcur = xb, xblk = 0., fpre, fcur, fblk = 0., spre = 0., scur = 0., sbis;
    double delta, stry, dpre, dblk;
    fpre = evroot_f(c, xpre);
    fcur = evroot_f(c, xcur);
    if (fpre == 0) return xpre;
    if (fcur == 0) return xcur;
    if (signbit(fpre) == signbit(fcur)) { *err = 1; return xcur; }
    for (int i = 0; i < iter; i++) {
        if (fpre != 0 && fcur != 0 && (signbit(fpre) != signbit(fcur))) {
            xblk = xpre;
            fblk = fpre;
            spre = scur = xcur - xpre;
        }
        if (fabs(fblk) < fabs(fcur)) {
            xpre = xcur; xcur = xblk; xblk = xpre;
            fpre = fcur; fcur = fblk; fblk = fpre;
        }
        delta = (xtol + rtol * fabs(xcur)) / 2;
        sbis = (xblk - xcur) / 2;
        if (fcur == 0 || fabs(sbis) < delta) return xcur;
        if (fabs(spre) > delta && fabs(fcur) < fabs(fpre)) {
            if (xpre == xblk) {
                stry = -fcur * (xcur - xpre) / (fcur - fpre);
            } else {
                dpre = (fpre - fcur) / (xpre - xcur);
                dblk = (fblk - fcur) / (xblk - xcur);
                stry = -fcur * (fblk * dblk - fpre * dpre) / (dblk * dpre * (fblk - fpre));
            }
            double m1 = fabs(spre), m2 = 3 * fabs(sbis) - delta;
            if (2 * fabs(stry) < (m1 < m2 ? m1 : m2)) {
                spre = scur; scur = stry;
            } else {
                spre = sbis; scur = sbis;
            }
        } else {
            spre = sbis; scur = sbis;
        }
        xpre = xcur;
        fpre = fcur;
        if (fabs(scur) > delta) xcur += scur;
        else xcur += (sbis > 0 ? delta : -delta);
        fcur = evroot_f(c, xcur);
    }
    return xcur;
}

/* test hook: brentq on a step function at `s` (+1 right of s) */
typedef struct { double s; int n; } stepf_t;
double orc_brentq_step(double a, double b, double s, int *ncalls)
{
    /* same control flow as brentq() above with f(x) = x > s ? 1 : -1 */
    double xpre = a, xcur = b, xblk = 0., fpre, fcur, fblk = 0., spre = 0., scur = 0., sbis, delta;
    double xtol = 4 * DBL_EPSILON, rtol = 4 * DBL_EPSILON;
    int n = 0;
#define SF(x) (n++, ((x) > s ? 1.0 : -1.0))
    fpre = SF(xpre);
    fcur = SF(xcur);
    for (int i = 0; i < 100; i++) {
        if (signbit(fpre) != signbit(fcur)) { xblk = xpre; fblk = fpre; spre = scur = xcur - xpre; }
        if (fabs(fblk) < fabs(fcur)) {
            xpre = xcur; xcur = xblk; xblk = xpre;
            fpre = fcur; fcur = fblk; fblk = fpre;
        }
        delta = (xtol + rtol * fabs(xcur)) / 2;
        sbis = (xblk - xcur) / 2;
        if (fabs(sbis) < delta) break;
        spre = sbis; scur = sbis; /* |fcur| < |fpre| never holds for +-1 */
        xpre = xcur; fpre = fcur;
        if (fabs(scur) > delta) xcur += scur; else xcur += (sbis > 0 ? delta : -delta);
        fcur = SF(xcur);
    }
#undef SF
    (void)spre;
    *ncalls = n;
    return xcur;
}

/* ---- not-a-knot cubic interpolant of bottom_angles: interp1d(kind="cubic")
 * (REF/launch_rays.py:397-399) == make_interp_spline(k=3, bc_type=None).  The
 * not-a-knot cubic spline is unique, so it is built here in piecewise-polynomial
 * form (rows as in scipy.interpolate.CubicSpline) and agrees with the B-spline
 * form to rounding.  pp[4*i..] = {y_i, s_i, c2, c3} on [x_i, x_{i+1}]. */
static int build_notaknot(const double *x, const double *y, int64_t n, double *pp)
{
    if (n < 4) return -1; /* interp1d: "x and y arrays must have at least 4 entries" */
    double *dx = malloc(sizeof(double) * (size_t)n * 6);
    double *sl = dx + n, *lo = sl + n, *di = lo + n, *up = di + n, *b = up + n;
    for (int64_t i = 0; i < n - 1; i++) {
        dx[i] = x[i + 1] - x[i];
        sl[i] = (y[i + 1] - y[i]) / dx[i];
    }
    for (int64_t i = 1; i < n - 1; i++) {
        lo[i] = dx[i];
        di[i] = 2 * (dx[i - 1] + dx[i]);
        up[i] = dx[i - 1];
        b[i] = 3 * (dx[i] * sl[i - 1] + dx[i - 1] * sl[i]);
    }
    double d = x[2] - x[0];
    di[0] = dx[1]; up[0] = d; lo[0] = 0;
    b[0] = ((dx[0] + 2 * d) * dx[1] * sl[0] + dx[0] * dx[0] * sl[1]) / d;
    d = x[n - 1] - x[n - 3];
    di[n - 1] = dx[n - 3]; lo[n - 1] = d; up[n - 1] = 0;
    b[n - 1] = (dx[n - 2] * dx[n - 2] * sl[n - 3] + (2 * d + dx[n - 2]) * dx[n - 3] * sl[n - 2]) / d;
    /* Thomas */
    for (int64_t i = 1; i < n; i++) {
        double m = lo[i] / di[i - 1];
        di[i] -= m * up[i - 1];
        b[i] -= m * b[i - 1];
    }
    b[n - 1] /= di[n - 1];
    for (int64_t i = n - 2; i >= 0; i--) b[i] = (b[i] - up[i] * b[i + 1]) / di[i];
    for (int64_t i = 0; i < n - 1; i++) {
        pp[4 * i + 0] = y[i];
        pp[4 * i + 1] = b[i];
        pp[4 * i + 2] = (3 * sl[i] - 2 * b[i] - b[i + 1]) / dx[i];
        pp[4 * i + 3] = (b[i] + b[i + 1] - 2 * sl[i]) / (dx[i] * dx[i]);
    }
    free(dx);
    return 0;
}

static int beta_eval(const orc_env *e, double x, double *beta)
{
    const double *xr = e->depth_ranges;
    int64_t n = e->nb;
    if (!(x >= xr[0] && x <= xr[n - 1])) return -1; /* bounds_error=True */
    int64_t i = searchsorted_left(xr, n, x) - 1;
    if (i < 0) i = 0;
    if (i > n - 2) i = n - 2;
    double t = x - xr[i];
    const double *c = e->pp + 4 * i;
    *beta = c[0] + t * (c[1] + t * (c[2] + t * c[3]));
    return 0;
}

/* test hook */
int orc_bottom_angle_interp(const double *xr, const double *ba, int64_t n, const double *xq,
                            int64_t nq, double *out)
{
    orc_env e;
    memset(&e, 0, sizeof e);
    e.depth_ranges = xr; e.bottom_angles = ba; e.nb = n;
    e.pp = malloc(sizeof(double) * 4 * (size_t)n);
    if (build_notaknot(xr, ba, n, e.pp)) { free(e.pp); return -1; }
    for (int64_t k = 0; k < nq; k++)
        if (beta_eval(&e, xq[k], &out[k])) out[k] = NAN;
    free(e.pp);
    return 0;
}

/* growable list of dense pieces for one solve_ivp segment */
typedef struct {
    dense_t *d;
    double *ts; /* sol.t : n+1 entries */
    int64_t n, cap;
} seg_t;
static void seg_push(seg_t *s, const dense_t *d, double t)
{
    if (s->n + 1 >= s->cap) {
        s->cap = s->cap ? 2 * s->cap : 256;
        s->d = realloc(s->d, sizeof(dense_t) * (size_t)s->cap);
        s->ts = realloc(s->ts, sizeof(double) * (size_t)(s->cap + 1));
    }
    s->d[s->n] = *d;
    s->ts[s->n + 1] = t;
    s->n++;
}

/* np.argmin(np.abs(range_save - t)) : first minimum */
static int64_t nearest_idx(const double *r, int64_t S, double t)
{
    int64_t best = 0;
    double bd = fabs(r[0] - t);
    for (int64_t j = 1; j < S; j++) {
        double dd = fabs(r[j] - t);
        if (dd < bd) { bd = dd; best = j; }
    }
    return best;
}

/* OdeSolution.__call__ for an ascending segment, SCIPY/common.py:139-242, written into the
 * slice [idx1, idx2) as REF/launch_rays.py:765-772 does (Q5) */
static void seg_sample(const seg_t *s, const double *r, int64_t S, double *T, double *Z, double *P,
                       double *XI)
{
    int64_t idx1 = nearest_idx(r, S, s->ts[0]);
    int64_t idx2 = nearest_idx(r, S, s->ts[s->n]);
    if (idx1 == idx2) return;
    for (int64_t j = idx1; j < idx2; j++) {
        int64_t k = searchsorted_left(s->ts, s->n + 1, r[j]) - 1;
        if (k < 0) k = 0;
        if (k > s->n - 1) k = s->n - 1;
        double y[3];
        dense_eval(&s->d[k], r[j], y);
        T[j] = y[0]; Z[j] = y[1]; P[j] = y[2];
        /* diagnostic for the tests: normalised abscissa of the quartic that produced the
         * sample (outside [0,1] = extrapolated, Q5; error amplification ~ |xi|^4) */
        if (XI) XI[j] = (r[j] - s->d[k].t_old) / s->d[k].h;
    }
}

/* per-ray counters */
typedef struct {
    int64_t n_steps, nfev, n_rej, n_seg;
} orc_stats;

/* debugging aid (orc_trace_ray): one row per step ATTEMPT of the ray being traced --
 * t, h, y[3], f[3], error_norm, accepted, h_abs after the attempt, segment index */
#define ORC_TRACE_COLS 12
static double *g_trace = 0;
static int64_t g_trace_cap = 0, g_trace_n = 0;
static void trace_row(double t, double h, const double *y, const double *f, double err, int acc, double h_next,
                      int64_t seg)
{
    if (!g_trace || g_trace_n >= g_trace_cap) return;
    double *r = g_trace + ORC_TRACE_COLS * g_trace_n++;
    r[0] = t; r[1] = h; r[2] = y[0]; r[3] = y[1]; r[4] = y[2]; r[5] = f[0]; r[6] = f[1]; r[7] = f[2];
    r[8] = err; r[9] = acc; r[10] = h_next; r[11] = (double)seg;
}

/* ---- one ray: REF/launch_rays.py:325-484 (_shoot_ray_array) with SciPy's solve_ivp/RK45
 * (SCIPY/ivp.py:654-726, rk.py:84-176, common.py:68-134) inlined, followed by
 * REF/launch_rays.py:745-784 (_interpolate_ray). */
static int shoot_one(const orc_env *e, const double *y0_in, double source_range, double receiver_range,
                     double rtol, double atol, int terminate_backwards, const double *r, int64_t S,
                     double *T, double *Z, double *P, double *XI, int *n_bott, int *n_surf,
                     orc_stats *st, int64_t max_steps)
{
    const double SAFETY = 0.9, MIN_FACTOR = 0.2, MAX_FACTOR = 10;
    /* validate_tol, SCIPY/common.py:44-51: rtol = np.maximum(rtol, 100 * EPS) (with a warning) */
    if (rtol < 100 * DBL_EPSILON) rtol = 100 * DBL_EPSILON;
    double x_int = source_range;
    double y[3] = {y0_in[0], y0_in[1], y0_in[2]};
    int nb = 0, ns = 0, status = ORC_OK;
    seg_t seg = {0, 0, 0, 0};
    double last_y[3] = {y[0], y[1], y[2]};
    for (int64_t j = 0; j < S; j++) T[j] = Z[j] = P[j] = NAN;
    if (XI) for (int64_t j = 0; j < S; j++) XI[j] = 1.0;
    memset(st, 0, sizeof *st);

    while (x_int < receiver_range) {
        /* ---- solve_ivp(derivsrd, (x_int, receiver_range), y, RK45, events, dense) ---- */
        double t = x_int, t_bound = receiver_range;
        double f[3], K[7][3];
        derivsrd(e, t, y, f); st->nfev++;
        /* select_initial_step, SCIPY/common.py:68-134 (order = 4, direction = +1) */
        double h_abs;
        {
            double interval = fabs(t_bound - t);
            double sc[3], a[3], b[3];
            for (int i = 0; i < 3; i++) {
                sc[i] = atol + fabs(y[i]) * rtol;
                a[i] = y[i] / sc[i];
                b[i] = f[i] / sc[i];
            }
            double d0 = rms3(a), d1 = rms3(b), h0;
            if (d0 < 1e-5 || d1 < 1e-5) h0 = 1e-6; else h0 = 0.01 * d0 / d1;
            if (!(h0 < interval)) h0 = interval; /* min(h0, interval) */
            double y1[3], f1[3], dd[3];
            for (int i = 0; i < 3; i++) y1[i] = y[i] + h0 * 1.0 * f[i];
            derivsrd(e, t + h0 * 1.0, y1, f1); st->nfev++;
            for (int i = 0; i < 3; i++) dd[i] = (f1[i] - f[i]) / sc[i];
            double d2 = rms3(dd) / h0, h1;
            if (d1 <= 1e-15 && d2 <= 1e-15) {
                h1 = h0 * 1e-3; if (!(h1 > 1e-6)) h1 = 1e-6; /* max(1e-6, h0*1e-3) */
            } else {
                double m = (d2 > d1) ? d2 : d1; /* max(d1, d2) */
                h1 = orc_pow(0.01 / m, 1.0 / 5.0);
            }
            h_abs = 100 * h0;                      /* min(100*h0, h1, interval, max_step=inf) */
            if (h1 < h_abs) h_abs = h1;
            if (interval < h_abs) h_abs = interval;
        }
        double g[4], g_new[4];
        for (int k = 0; k < 4; k++) g[k] = EVENTS[k](e, t, y);
        seg.n = 0;
        if (!seg.ts) { seg.cap = 256; seg.d = malloc(sizeof(dense_t) * 256); seg.ts = malloc(sizeof(double) * 257); }
        seg.ts[0] = t;
        int seg_status = -2; /* None */
        int ev_hit = -1;
        double t_event = 0;

        while (seg_status == -2) {
            /* ---- RK45._step_impl, SCIPY/rk.py:111-176 ---- */
            double min_step = 10 * fabs(nextafter(t, INFINITY) - t);
            if (h_abs < min_step) h_abs = min_step; /* max_step = inf */
            int accepted = 0, rejected = 0;
            double h = 0, t_new = 0, y_new[3], f_new[3];
            while (!accepted) {
                if (h_abs < min_step) { status = ORC_STEP_TOO_SMALL; goto done; }
                h = h_abs * 1.0;
                t_new = t + h;
                if (1.0 * (t_new - t_bound) > 0) t_new = t_bound;
                h = t_new - t;
                h_abs = fabs(h);
                /* rk_step, SCIPY/rk.py:14-71 */
                for (int i = 0; i < 3; i++) K[0][i] = f[i];
                for (int s = 1; s < 6; s++) {
                    double ys[3];
                    for (int i = 0; i < 3; i++) {
                        double dot = 0.0;
                        for (int q = 0; q < s; q++) dot += K[q][i] * RK_A[s][q];
                        ys[i] = y[i] + dot * h;
                    }
                    derivsrd(e, t + RK_C[s] * h, ys, K[s]);
                }
                for (int i = 0; i < 3; i++) {
                    double dot = 0.0;
                    for (int q = 0; q < 6; q++) dot += K[q][i] * RK_B[q];
                    y_new[i] = y[i] + h * dot;
                }
                derivsrd(e, t + h, y_new, f_new);
                st->nfev += 6;
                for (int i = 0; i < 3; i++) K[6][i] = f_new[i];
                double en[3];
                for (int i = 0; i < 3; i++) {
                    double ay = fabs(y[i]), an = fabs(y_new[i]);
                    double scale = atol + ((ay > an) ? ay : an) * rtol; /* np.maximum (NaN-free) */
                    double dot = 0.0;
                    for (int q = 0; q < 7; q++) dot += K[q][i] * RK_E[q];
                    en[i] = (dot * h) / scale;
                }
                double error_norm = rms3(en);
                if (error_norm < 1) {
                    double factor;
                    if (error_norm == 0) factor = MAX_FACTOR;
                    else {
                        factor = SAFETY * orc_pow(error_norm, -0.2);
                        if (!(factor < MAX_FACTOR)) factor = MAX_FACTOR; /* min(MAX, .) */
                    }
                    if (rejected && !(factor < 1)) factor = 1; /* min(1, factor) */
                    h_abs *= factor;
                    accepted = 1;
                } else {
                    double fac = SAFETY * orc_pow(error_norm, -0.2);
                    if (!(fac > MIN_FACTOR)) fac = MIN_FACTOR; /* max(MIN, .) ; NaN -> MIN */
                    h_abs *= fac;
                    rejected = 1;
                    st->n_rej++;
                }
                if (g_trace) trace_row(t, h, y, f, error_norm, accepted, h_abs, st->n_seg);
            }
            /* dense output for the accepted step: Q = K.T @ P */
            dense_t d;
            d.t_old = t; d.t = t_new; d.h = h;
            for (int i = 0; i < 3; i++) {
                d.y_old[i] = y[i];
                for (int j = 0; j < 4; j++) {
                    double s = 0.0;
                    for (int q = 0; q < 7; q++) s += K[q][i] * RK_P[q][j];
                    d.Q[i][j] = s;
                }
            }
            double t_old = t;
            t = t_new;
            for (int i = 0; i < 3; i++) { y[i] = y_new[i]; f[i] = f_new[i]; }
            st->n_steps++;
            if (1.0 * (t - t_bound) >= 0) seg_status = 0; /* finished, SCIPY/base.py:197 */

            /* events, SCIPY/ivp.py:671-694 */
            for (int k = 0; k < 4; k++) g_new[k] = EVENTS[k](e, t, y);
            int active[4], na = 0;
            for (int k = 0; k < 4; k++) {
                int up = (g[k] <= 0) && (g_new[k] >= 0), down = (g[k] >= 0) && (g_new[k] <= 0);
                int m = (up && EV_DIR[k] > 0) || (down && EV_DIR[k] < 0) || ((up || down) && EV_DIR[k] == 0);
                if (m) active[na++] = k;
            }
            double t_append = t;
            if (na > 0) {
                /* handle_events: all four are terminal (max_events = 1): the earliest root wins,
                 * ties -> lowest event index (stable argsort) */
                double best = 0; int bk = -1;
                for (int q = 0; q < na; q++) {
                    evroot_ctx c = {e, EVENTS[active[q]], &d};
                    int berr = 0;
                    double root = brentq(&c, t_old, t, 4 * DBL_EPSILON, 4 * DBL_EPSILON, 100, &berr);
                    if (berr) { status = ORC_EVENT_ERROR; goto done; }
                    if (bk < 0 || root < best) { best = root; bk = active[q]; }
                }
                seg_status = 1;
                ev_hit = bk;
                t_event = best;
                t_append = best;
                double ye[3];
                dense_eval(&d, best, ye);
                for (int i = 0; i < 3; i++) y[i] = ye[i];
            }
            for (int k = 0; k < 4; k++) g[k] = g_new[k];
            /* ts/ys bookkeeping incl. the "donot_append" corner, SCIPY/ivp.py:696-705 */
            if (seg.n + 1 > 1 && seg.ts[seg.n] == t_append) {
                /* nothing appended, interpolant dropped */
            } else {
                seg_push(&seg, &d, t_append);
                for (int i = 0; i < 3; i++) last_y[i] = y[i];
            }
            if (st->n_steps > max_steps) { status = ORC_MAX_STEPS; goto done; }
        }
        /* back in _shoot_ray_array (REF/launch_rays.py:418-480) */
        st->n_seg++;
        seg_sample(&seg, r, S, T, Z, P, XI);
        if (seg_status == 0) break;
        /* y_intermediate = sol.y[:, -1] */
        for (int i = 0; i < 3; i++) y[i] = last_y[i];
        if (ev_hit == 0 || ev_hit == 1) x_int = t_event;
        else if (ev_hit == 2) { status = ORC_VERTICAL; goto done; }
        else { status = ORC_BBOX; goto done; }
        double c, theta = ray_angle(e, x_int, y, &c), theta_b;
        if (ev_hit == 0) { theta_b = -theta; ns++; }
        else {
            double beta;
            if (beta_eval(e, x_int, &beta)) { status = ORC_BETA_RANGE; goto done; }
            theta_b = 2 * beta - theta;
            nb++;
        }
        if (terminate_backwards && (fabs(theta_b) > 90)) { status = ORC_BACKWARD; goto done; }
        y[2] = orc_sin(theta_b * (M_PI / 180.0)) / c; /* np.radians(x) = x*(pi/180) */
    }
    /* _interpolate_ray: last column is the exact final state (REF/launch_rays.py:775-777) */
    T[S - 1] = last_y[0]; Z[S - 1] = last_y[1]; P[S - 1] = last_y[2];
done:
    free(seg.d);
    free(seg.ts);
    *n_bott = nb;
    *n_surf = ns;
    if (status != ORC_OK)
        for (int64_t j = 0; j < S; j++) T[j] = Z[j] = P[j] = NAN;
    return status;
}

/* ---- batched entry used by tests / bench cpu_baseline.  y0 is [N][3]; T/z/p are [N][S]
 * ray-major.  `r` must be np.linspace(source_range, receiver_range, S) computed by the
 * caller (REF/launch_rays.py:309,561).  Parallel over rays with OpenMP when built with it. */
int orc_shoot_fan(const double *cin, const double *cpin, const double *rin, const double *zin,
                  int64_t nr, int64_t nz, const double *depths, const double *depth_ranges,
                  const double *bottom_angles, int64_t nb, const double *y0, int64_t N,
                  double source_range, double receiver_range, const double *r, int64_t S,
                  double rtol, double atol, int terminate_backwards, int64_t max_steps,
                  double *T, double *Z, double *P, double *XI, int32_t *n_bott, int32_t *n_surf,
                  int32_t *status, int64_t *n_steps, int64_t *nfev, int64_t *n_rej)
{
    orc_env e = {cin, cpin, rin, zin, nr, nz, depths, depth_ranges, bottom_angles, nb, 0};
    e.pp = malloc(sizeof(double) * 4 * (size_t)(nb > 1 ? nb : 1));
    if (build_notaknot(depth_ranges, bottom_angles, nb, e.pp)) { free(e.pp); return -1; }
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t k = 0; k < N; k++) {
        orc_stats st;
        int b, s;
        status[k] = shoot_one(&e, y0 + 3 * k, source_range, receiver_range, rtol, atol,
                              terminate_backwards, r, S, T + k * S, Z + k * S, P + k * S,
                              XI ? XI + k * S : 0, &b, &s, &st, max_steps);
        n_bott[k] = b; n_surf[k] = s;
        n_steps[k] = st.n_steps; nfev[k] = st.nfev; n_rej[k] = st.n_rej;
    }
    free(e.pp);
    return 0;
}

/* ---- debugging aid: the step attempts of ONE ray (single-threaded; see trace_row) */
int64_t orc_trace_ray(const double *cin, const double *cpin, const double *rin, const double *zin,
                      int64_t nr, int64_t nz, const double *depths, const double *depth_ranges,
                      const double *bottom_angles, int64_t nb, const double *y0, double source_range,
                      double receiver_range, double rtol, double atol, int terminate_backwards,
                      double *rows, int64_t max_rows)
{
    orc_env e = {cin, cpin, rin, zin, nr, nz, depths, depth_ranges, bottom_angles, nb, 0};
    e.pp = malloc(sizeof(double) * 4 * (size_t)(nb > 1 ? nb : 1));
    if (build_notaknot(depth_ranges, bottom_angles, nb, e.pp)) { free(e.pp); return -1; }
    double r[2] = {source_range, receiver_range}, T[2], Z[2], P[2];
    int b, s2;
    orc_stats st;
    g_trace = rows; g_trace_cap = max_rows; g_trace_n = 0;
    shoot_one(&e, y0, source_range, receiver_range, rtol, atol, terminate_backwards, r, 2, T, Z, P, 0, &b, &s2,
              &st, 10000000);
    g_trace = 0;
    free(e.pp);
    return g_trace_n;
}

/* ---- unit-level hooks for the a1-a8 golden vectors */
void orc_derivs(const double *cin, const double *cpin, const double *rin, const double *zin, int64_t nr,
                int64_t nz, double x, const double *y, double *out)
{
    orc_env e = {cin, cpin, rin, zin, nr, nz, 0, 0, 0, 0, 0};
    derivsrd(&e, x, y, out);
}
void orc_ray_angle(const double *cin, const double *rin, const double *zin, int64_t nr, int64_t nz,
                   double x, const double *y, double *theta, double *c)
{
    orc_env e = {cin, 0, rin, zin, nr, nz, 0, 0, 0, 0, 0};
    *theta = ray_angle(&e, x, y, c);
}
void orc_events(const double *cin, const double *rin, const double *zin, int64_t nr, int64_t nz,
                const double *depths, const double *depth_ranges, int64_t nb, double x, const double *y,
                double *out4)
{
    orc_env e = {cin, 0, rin, zin, nr, nz, depths, depth_ranges, 0, nb, 0};
    for (int k = 0; k < 4; k++) out4[k] = EVENTS[k](&e, x, y);
}
void orc_set_num_threads(int n)
{
#ifdef _OPENMP
    extern void omp_set_num_threads(int);
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
int orc_num_threads(void)
{
#ifdef _OPENMP
    extern int omp_get_max_threads(void);
    return omp_get_max_threads();
#else
    return 1;
#endif
}
