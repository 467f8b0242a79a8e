// pgr_hip.hip -- MI355X (gfx950) ray-fan integrator behind the C ABI of include/pgr.h.
//
// One ray per lane (wave64).  Every lane runs SciPy's adaptive Dormand-Prince 5(4)
// controller on the ray equations y = [T, z, p] over range x with the reference's
// bilinear c / dc/dz tables, +-1 step-function events located by brentq-style
// bisection on the quartic dense output, reflection at surface / bottom and the
// reference's nearest-index re-sampling -- restating, on the device,
//   REF/integration_processes.py:26-334   (derivsrd, bilinear/linear interp, events)
//   REF/launch_rays.py:325-484, 593-681    (_shoot_ray_array, _shoot_ray_segment)
//   REF/launch_rays.py:745-784             (_interpolate_ray)
//   SCIPY/rk.py:14-180,377-404,552-574 ; SCIPY/common.py:63-134 ; SCIPY/ivp.py:28-156,654-726
// (REF = /root/reference/src/pygenray, SCIPY = scipy/integrate/_ivp of SciPy 1.15.3).
//
// Layout: the c and dc/dz tables are interleaved node-wise as double2 {c, cp} so one
// 16-byte access fetches both values of a node; for range-independent tables the single
// depth profile (nz x 16 B, 96 KB at nz = 6000) is staged into LDS once per workgroup, next
// to zin and its bucket table when the depth grid is not uniform, and the bathymetry.
// State (x, y, f, h, K1..K7) lives in VGPRs.  No MFMA: there is no contraction here.
//
// Arithmetic that feeds back into the integration is written in the reference's operation
// order and compiled with -ffp-contract=off; divide / sqrt are correctly rounded, the three libm
// calls of the reference (err ** -0.2, arcsin, sin) are evaluated CORRECTLY ROUNDED
// (pgr_crmath.h) and the event locator returns brentq's own root, so the result is bit-identical
// to the CPU oracle in its correctly-rounded-libm mode (oracle/ray_oracle.c, ORC_MATH_CR).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <float.h>
#include <string>
#include <vector>
#include <mutex>
#include <thread>
#include <initializer_list>

#include "../../include/pgr.h"
#include "pgr_crmath.h"
#ifdef PGR_LIBM_TRIG  // experiments: the device library's asin / sin / pow at bounces (NOT bit-identical)
#define pgr_cr_asin(x) asin(x)
#define pgr_cr_sin(x) sin(x)
#define pgr_cr_pow_p02(x) pow((x), 0.2)
#define PGR_ASIN_DD_T double
#define PGR_ASIN_DD(v) asin(v)
#define PGR_ASIN_DD_HI(a) (a)
#define PGR_SIN_REFLECT(x, v, a) sin(x)
#else
#define PGR_ASIN_DD_T struct pgr_dd
#define PGR_ASIN_DD(v) pgr_cr_asin_dd(v)
#define PGR_ASIN_DD_HI(a) ((a).h)
#define PGR_SIN_REFLECT(x, v, a) pgr_cr_sin_near_minus_asin((x), (v), (a))
#endif

// ------------------------------------------------------------------------------------
// device-side environment description
// ------------------------------------------------------------------------------------
struct EnvDev {
    const double2* tab;  // [nr][nz] {c, cp}
    const double* rin;   // [nr]
    const double* zin;   // [nz]
    const double* depths;        // [nb]
    const double* depth_ranges;  // [nb]
    const double* pp;            // [nb-1][4] not-a-knot cubic of bottom_angles
    int nr, nz, nb;
    int row_stride;  // nz, or 0 when the table is range independent (one stored row)
    int z_uniform, r_uniform, b_uniform;  // grid[j] == g0 + j*dg bitwise (host verified)
    int beta_zero;  // all bottom angles are 0 -> the cubic is identically 0
    int z_pow2;     // z_uniform, dz a power of two and every zin[j+1]-zin[j] == dz bitwise
    int z_simple;   // z_pow2 and zin[0] == 0: zin[j] == j*dz
    double b_zmin, b_xlo, b_xhi;  // min(depths) - 1 m and the bathymetry table's range span
    double z0, dz, inv_dz;
    double r0, dr, inv_dr;
    double b0, db, inv_db;
    double zlo_tol, zhi_tol, rlo_tol, rhi_tol;  // bbox bounds -+ 1e-6 (REF/integration_processes.py:295-302)
    double c_lo, c_hi;  // min and max of the sound-speed table (c_hi with a 1e-3 margin): |p| c_hi < 1 settles |p c| <= 1 without a look-up
    // bucketed depth search for non-uniform zin (e.g. the flat-earth transformed grid): zbucket[k]
    // = the cell index at the lower edge of uniform bin k of width zb_w <= 0.9 min(diff(zin)), so
    // the cell of any z in bin k is zbucket[k] or zbucket[k] + 1 (host verified)
    const unsigned short* zbucket;
    int z_bucket, zb_B;
    double zb_z0, zb_inv_w;
    // ... or, when zin is smooth enough (the flat-earth grid is), no table at all: a quadratic
    // g(u) = q0 + u (q1 + u q2), u = (z - zin[0]) / span, with |g(zin[j]) - j| <= 0.45 for every
    // node and g' > 0 (host verified), so the cell of z is floor(g - 0.5) or the next one
    int z_quad;
    double zq_c0, zq_c1, zq_c2, zq_inv_span;
};

struct FanArgs {
    const double* y0;      // [N][3]
    const double* r_save;  // [S]
    double* T;
    double* Z;
    double* P;             // may be null
    double* end_state;     // [N][3] may be null
    int32_t* n_bott;
    int32_t* n_surf;
    int32_t* status;
    int32_t* n_steps;
    int32_t* n_rej;
    int64_t N;
    int64_t stride_ray, stride_smp;  // element strides of T/Z/P
    int32_t S;
    double x0, x1, rtol, atol;
    double inv_dsave;  // (S-1)/(x1-x0) guess for nearest-sample index
    double save_step;  // linspace step when save_formula
    int save_formula;  // r_save[j] == j*save_step + x0 bitwise (host verified)
    int park_lanes, park_trips;  // service batching thresholds
    int bathy_lds_off;    // byte offset of the LDS copy of {depth_ranges[nb], depths[nb]}, or -1 (read from HBM)
    const int* wave_map;  // [gridDim.x * waves_per_block] global wave of each slot, -1 = empty; null = strided deal
    int64_t max_steps;
    uint32_t flags;
};

static_assert(alignof(FanArgs) == 8, "the fan kernel re-reads its FanArgs at kernel-argument offset 8");

#define RUNNING (-1)

// ------------------------------------------------------------------------------------
// arithmetic building blocks
//
// The adaptive controller makes the solution extremely sensitive to rounding: the embedded
// error estimate is a near-cancelling sum (~1e-8 relative rounding noise), err^-0.2 feeds it
// into every following step size, and a 1e-9 relative change of the step sequence moves a
// 1000 km ray by millimetres (1e-6 relative) -- measured by building this file with FMA
// contraction on (-DPGR_FMA).  To stay within 1e-8 of the CPU reference the default build
// therefore reproduces the reference's IEEE arithmetic operation by operation
// (-ffp-contract=off) and only replaces the *expansions* of divide and sqrt by cheaper ones
// that are still correctly rounded for the operand ranges that occur here:
//   * a/b: two Newton steps on v_rcp_f64 + one Markstein correction (8 VALU ops instead of the
//     ~14 of the generic expansion with v_div_scale / v_div_fmas / v_div_fixup).  Correctly
//     rounded unless the exact quotient is within ~2^-104 of a rounding boundary (0 mismatches
//     in 2e6 random operands, tests/test_hip_parity.py::test_arithmetic_building_blocks);
//   * sqrt: v_rsq_f64 + Newton + one residual correction (0 mismatches in 2e6);
//   * err ** -0.2, arcsin, sin: CORRECTLY ROUNDED (pgr_crmath.h) -- the reference calls the platform
//     libm for them, which is faithful but not correctly rounded, so the oracle's ORC_MATH_CR mode (the
//     same functions in binary128, rounded once) is what this file matches bit for bit;
//     10*ulp(t) by integer arithmetic (exact).
// -DPGR_STRICT uses the compiler's IEEE divide/sqrt; -DPGR_FMA additionally allows
// contraction and a 2-ulp rsqrt (fastest, NOT within 1e-8 of the reference: experiments only).
// ------------------------------------------------------------------------------------
#ifdef PGR_STRICT
#define PGR_FAST 0
#else
#define PGR_FAST 1
#endif

__device__ __forceinline__ double frcp(double b)
{
#if PGR_FAST
    double y = __builtin_amdgcn_rcp(b);
    double e = fma(-b, y, 1.0);
    y = fma(y, e, y);
    e = fma(-b, y, 1.0);
    y = fma(y, e, y);
    return y;
#else
    return 1.0 / b;
#endif
}
__device__ __forceinline__ double fdiv(double a, double b)
{
#if PGR_FAST
    // one Newton step is enough before the correction (v_rcp_f64 is good to 4.6e-8)
    double y = __builtin_amdgcn_rcp(b);
    double e = fma(-b, y, 1.0);
    y = fma(y, e, y);
    double q = a * y;
    double r = fma(-q, b, a);
    return fma(r, y, q);
#else
    return a / b;
#endif
}
// 1/b to ~2e-15 (v_rcp_f64 + one Newton step): a SEED for fdiv_y, whose correction step squares
// the seed's error -- exactly what fdiv() itself does
__device__ __forceinline__ double frcp_seed(double b)
{
    double y = __builtin_amdgcn_rcp(b);
    double e = fma(-b, y, 1.0);
    return fma(y, e, y);
}
// q = a / b given y ~ 1/b (shared reciprocal)
__device__ __forceinline__ double fdiv_y(double a, double b, double y)
{
    double q = a * y;
    double r = fma(-q, b, a);
    return fma(r, y, q);
}
// 1/sqrt(x), x > 0 and normal: raw Newton form (<= 2 ulp), building block of fsqrt
__device__ __forceinline__ double frsqrt_raw(double x)
{
#if PGR_FAST
    double y = __builtin_amdgcn_rsq(x);
    // two Newton steps: y <- y + y*(1 - x y^2)/2
    double e = fma(-x * y, y, 1.0);
    y = fma(y * 0.5, e, y);
    e = fma(-x * y, y, 1.0);
    y = fma(y * 0.5, e, y);
    return y;
#else
    return 1 / sqrt(x);
#endif
}
__device__ __forceinline__ double fsqrt(double x)
{
#if PGR_FAST
    double y = __builtin_amdgcn_rsq(x);  // good to 5.2e-8: one Newton step, then the residual
    double e = fma(-x * y, y, 1.0);
    y = fma(y * 0.5, e, y);
    double g = x * y;
    double d = fma(-g, g, x);
    double r = fma(d * 0.5, y, g);
    // no branch: sqrt(+-0) = +-0 and sqrt(inf) = inf by select (rsq gives inf / 0 there and the
    // refinement NaN); x < 0 and NaN come out NaN by themselves
    return (x == 0.0 || x == INFINITY) ? x : r;
#else
    return sqrt(x);
#endif
}
// the reference's `1 / np.sqrt(arg)`: RN(1 / RN(sqrt x)).  The refined rsq is an excellent seed
// for 1/s (s = RN(sqrt x)): one correction step lands on the correctly rounded reciprocal.
__device__ __forceinline__ double frsqrt(double x)
{
#ifdef PGR_FMA
    return frsqrt_raw(x);
#elif PGR_FAST
    double y = __builtin_amdgcn_rsq(x);
    double e = fma(-x * y, y, 1.0);
    y = fma(y * 0.5, e, y);
    double g = x * y;
    double d = fma(-g, g, x);
    double s = fma(d * 0.5, y, g);   // RN(sqrt x)
    double r = fma(-s, y, 1.0);      // y ~ 1/s to ~4e-15
    // one correction: y (1 + r) = 1/s to ~2e-29 relative, rounded once by the fma -- RN(1/s)
    // unless 1/s lies within ~2^-96 (relative) of a rounding boundary, the same class as fdiv().
    // ONE input class is that close: s = 1 - 2^-53 (all-ones significand -- the classic exception of
    // Newton-Raphson reciprocals), where 1/s = 1 + 2^-53 + 2^-106 sits 2^-106 above a tie and the fma
    // returns 1 instead of 1 + 2^-52.  It is reached by x = 1 - c^2 p^2 in {1 - 2^-53, 1 - 2^-52}, i.e.
    // a stage that lands within |p c| < 1.7e-8 of a turning point: ~1e-6 per step, 2 rays in 10 000 of
    // the headline fan (scripts/trace_diff.py found it).  The select below repairs it exactly; it costs
    // 2-3 instructions in each of the 7 right-hand sides of an attempt (+2 % on the critical path), so
    // the product leaves it out and DESIGN.md section 4 reports the 99.98 % it leaves -- build with
    // -DPGR_EXACT_RSQRT to see the last rays fall into place.
    double out = fma(y, r, y);
#ifdef PGR_EXACT_RSQRT
    out = (s == 0x1.fffffffffffffp-1) ? 0x1.0000000000001p+0 : out;
#endif
    return out;
#else
    return 1 / sqrt(x);
#endif
}
// err ** -0.2 for err in [1e-7, 1e4], correctly rounded (pgr_crmath.h; -DPGR_POW_2ULP: the 2-ulp
// Newton iteration of round 1, 18 instructions shorter -- experiments only, NOT bit-identical)
__device__ __forceinline__ double pow_m02(double x, const double fifth = 0.2, const double kln2 = PGR_CR_POW_KLN2)
{
#ifdef PGR_POW_2ULP
    float xf = (float)x;
    double y = (double)__builtin_amdgcn_exp2f(-0.2f * __builtin_amdgcn_logf(xf));
#pragma unroll
    for (int k = 0; k < 2; k++) {
        double y2 = y * y, y4 = y2 * y2, y5 = y4 * y;
        double e = fma(-x, y5, 1.0);
        y = fma(y * 0.2, e, y);
    }
    return y;
#else
    return pgr_cr_pow_m02_k(x, fifth, kln2);
#endif
}
// 10 * |nextafter(t, +inf) - t|, SCIPY/rk.py:119
__device__ __forceinline__ double min_step_of(double t)
{
#if PGR_FAST
    long long b = __double_as_longlong(t);
    double nx = (t == 0.0) ? 4.9406564584124654e-324 : __longlong_as_double(t > 0 ? b + 1 : b - 1);
    return 10 * fabs(nx - t);
#else
    return 10 * fabs(nextafter(t, INFINITY) - t);
#endif
}
// g0 + j*dg with NO contraction: must reproduce the table coordinate bit for bit
__device__ __forceinline__ double grid_at(double g0, double dg, int j)
{
#pragma clang fp contract(off)
    double m = (double)j * dg;
    return g0 + m;
}

// the lanes of the wave whose predicate holds, as a mask: HIP's __ballot / __any take an int and
// compare it with 0 again (v_cndmask + v_cmp per call); the builtin takes the condition mask as it is
__device__ __forceinline__ unsigned long long ballot64(bool b) { return __builtin_amdgcn_ballot_w64(b); }

// min(max(j, 0), hi) for hi >= 0 in one instruction
__device__ __forceinline__ int clamp_index(int j, int hi)
{
    int r;
    asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(j), "s"(hi));
    return r;
}

// ------------------------------------------------------------------------------------
// grid cell lookup: np.searchsorted(grid, q) - 1 clamped to [0, n-2]
// (REF/integration_processes.py:152-157).  side='left': grid[j] < q <= grid[j+1].
// ------------------------------------------------------------------------------------
__device__ __forceinline__ int cell_uniform(double q, double g0, double dg, double inv_dg, int n)
{
    double t = (q - g0) * inv_dg;
    t = fmin(fmax(t, -1.0), (double)n);  // NaN -> -1
    int j = (int)floor(t);
    double gj = grid_at(g0, dg, j), gj1 = grid_at(g0, dg, j + 1);
    j += (gj >= q) ? -1 : ((gj1 < q) ? 1 : 0);
    return min(max(j, 0), n - 2);
}

__device__ __forceinline__ int cell_search(double q, const double* __restrict__ g, int n)
{
    int lo = 0, hi = n;
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (g[mid] < q) lo = mid + 1; else hi = mid;
    }
    return min(max(lo - 1, 0), n - 2);
}

// ------------------------------------------------------------------------------------
// Dormand-Prince coefficients, SCIPY/rk.py:377-404
// ------------------------------------------------------------------------------------
#define A21 (1.0 / 5)
#define A31 (3.0 / 40)
#define A32 (9.0 / 40)
#define A41 (44.0 / 45)
#define A42 (-56.0 / 15)
#define A43 (32.0 / 9)
#define A51 (19372.0 / 6561)
#define A52 (-25360.0 / 2187)
#define A53 (64448.0 / 6561)
#define A54 (-212.0 / 729)
#define A61 (9017.0 / 3168)
#define A62 (-355.0 / 33)
#define A63 (46732.0 / 5247)
#define A64 (49.0 / 176)
#define A65 (-5103.0 / 18656)
#define B1 (35.0 / 384)
#define B3 (500.0 / 1113)
#define B4 (125.0 / 192)
#define B5 (-2187.0 / 6784)
#define B6 (11.0 / 84)
#define E1 (-71.0 / 57600)
#define E3 (71.0 / 16695)
#define E4 (-71.0 / 1920)
#define E5 (17253.0 / 339200)
#define E6 (-22.0 / 525)
#define E7 (1.0 / 40)
#define C2 (1.0 / 5)
#define C3 (3.0 / 10)
#define C4 (4.0 / 5)
#define C5 (8.0 / 9)

// ------------------------------------------------------------------------------------
// per-kernel context: where table nodes come from
// ------------------------------------------------------------------------------------
// ZS ("z simple"): zin[j] == j*dz bitwise with dz a power of two and zin[0] == 0 (e.g. the
// reference's default np.arange(0, 6000, 1)): the cell index is ceil(z/dz) - 1 and the weight an
// exact scaling -- no search, no fix-up, no division.
// ZM = 2 ("z bucketed"): any other increasing zin whose bucket table fits the LDS: one LDS read
// gives the candidate cell, the next three nodes of zin (and of the profile) are read together
// and a compare picks the cell -- two dependent LDS reads instead of a 13-step binary search
// through L2 (the reference's default flat-earth grid: 36 -> 11 ms per 1e5-ray fan).
// the next double above a finite x (the band arithmetic of the event locator)
__device__ __forceinline__ double next_up(double x)
{
    const long long b = __double_as_longlong(x);
    return __longlong_as_double((x == 0.0) ? 1LL : (b >= 0 ? b + 1 : b - 1));
}

template <bool LDS_TAB, int ZM>
struct Ctx {
    static constexpr bool ZS = (ZM == 1 || ZM == 4);  // ZM == 4: ZS with dz == 1.0 (np.arange(0, 6000, 1)): no scaling at all
    const EnvDev& e;
    const double2* lds;  // LDS copy of the (single) depth profile when LDS_TAB
    const double* bx;               // depth_ranges and depths: LDS copies when they fit, else HBM
    const double* bd;
    const double* lds_z;            // ZM == 2: LDS copy of zin
    const unsigned short* lds_zb;   // ZM == 2: LDS copy of zbucket
    const double h_zb_z0, h_zb_inv_w;
    const int h_zb_B;
    const double h_zq_c0, h_zq_c1, h_zq_c2, h_zq_inv_span;  // ZM == 3
    // per-lane caches: x only moves forward, so the range cell (and the bathymetry cell under
    // the ray) changes once every ~10 km; keep its edges and the reciprocal of its width
    mutable double r_lo, r_hi, r_yden, r_hi2;  // r_hi2: upper edge of the NEXT cell (uniform rin) or r_hi
    mutable int r_i;
    // wave-uniform copies of the fields the step loop touches (kept in SGPRs; the rest of the
    // descriptor is read from memory where it is needed)
    const double h_inv_dz, h_dz, h_r0, h_dr, h_inv_dr;
    const double h_zhi_tol, h_zlo_tol;  // events()
    // events(): a caller whose x never leaves [x0, x1] (the fan kernel: the rays march from a.x0 to a.x1)
    // declares that span; when it lies inside the table's range box and inside the bathymetry table,
    // the range tests of the bounding-box event cannot fire and "z above the shallowest bathymetry
    // node" alone rules the bottom event out: two thresholds on the step's common path instead of six
    mutable int x_guard = 0;
    mutable double zmin_eff = -INFINITY;
    const double h_b0, h_db, h_inv_db;  // bathy()
    const int h_nb, h_b_uniform;
    const double2* const h_tab;  // HBM table variant
    const int h_row_stride, h_z_uniform, h_z_pow2;
    const double h_z0;
    const double* const h_zin;
    const double* const h_rin;
    const int h_nz, h_nr, h_r_uniform;
    // fp64 literals of the step attempt.  An fp64 literal cannot be an inline operand (two s_mov_b32
    // per use, and every instruction of a lone wave costs an issue slot); the fan kernel pins these
    // in VGPRs (PGR_PIN below) where it has registers to spare, everywhere else they fold back
    // into literals.
    mutable double k_c2 = C2, k_c3 = C3, k_c4 = C4, k_c5 = C5, k_tiny = 1e-30, k_vert = 0.9999999998;
    __device__ __forceinline__ Ctx(const EnvDev& e_, const double2* l, const double* lz = nullptr,
                                   const unsigned short* lzb = nullptr, const double* lbx = nullptr)
        : e(e_), lds(l), bx(lbx ? lbx : e_.depth_ranges), bd(lbx ? lbx + e_.nb : e_.depths), lds_z(lz),
          lds_zb(lzb), h_zb_z0(e_.zb_z0), h_zb_inv_w(e_.zb_inv_w), h_zb_B(e_.zb_B),
          h_zq_c0(e_.zq_c0), h_zq_c1(e_.zq_c1), h_zq_c2(e_.zq_c2), h_zq_inv_span(e_.zq_inv_span),
          h_inv_dz(e_.inv_dz), h_dz(e_.dz), h_r0(e_.r0), h_dr(e_.dr),
          h_zhi_tol(e_.zhi_tol), h_zlo_tol(e_.zlo_tol), h_b0(e_.b0), h_db(e_.db),
          h_inv_db(e_.inv_db), h_nb(e_.nb), h_b_uniform(e_.b_uniform), h_tab(e_.tab),
          h_row_stride(e_.row_stride), h_z_uniform(e_.z_uniform), h_z_pow2(e_.z_pow2), h_z0(e_.z0),
          h_zin(e_.zin),
          h_inv_dr(e_.inv_dr), h_rin(e_.rin), h_nz(e_.nz), h_nr(e_.nr), h_r_uniform(e_.r_uniform)
    {
        r_lo = 1.0; r_hi = 0.0; r_yden = 1.0; r_hi2 = 0.0; r_i = 0;  // empty interval: first use refills
    }
    __device__ __forceinline__ void declare_span(double x0, double x1) const
    {
        x_guard = (x0 <= x1) && (x0 >= e.rlo_tol) && (x1 <= e.rhi_tol) && (x0 >= e.b_xlo) && (x1 <= e.b_xhi);
        zmin_eff = x_guard ? e.b_zmin : -INFINITY;
    }

    __device__ __forceinline__ int cell_z(double z, double& zj, double& zj1) const
    {
        int j;
        if (h_z_uniform) {
            j = cell_uniform(z, h_z0, h_dz, h_inv_dz, h_nz);
            zj = grid_at(h_z0, h_dz, j);
            zj1 = grid_at(h_z0, h_dz, j + 1);
        } else {
            j = cell_search(z, h_zin, h_nz);
            zj = h_zin[j];
            zj1 = h_zin[j + 1];
        }
        return j;
    }
    __device__ __forceinline__ int cell_r(double x, double& ri, double& ri1) const
    {
        int i;
        if (h_r_uniform) {
            i = cell_uniform(x, h_r0, h_dr, h_inv_dr, h_nr);
            ri = grid_at(h_r0, h_dr, i);
            ri1 = grid_at(h_r0, h_dr, i + 1);
        } else {
            i = cell_search(x, h_rin, h_nr);
            ri = h_rin[i];
            ri1 = h_rin[i + 1];
        }
        return i;
    }
    __device__ __forceinline__ void refill(double x) const
    {
        double ri, ri1;
        r_i = cell_r(x, ri, ri1);
        r_lo = ri; r_hi = ri1;
        r_yden = frcp(ri1 - ri);
        r_hi2 = (h_r_uniform && r_i + 2 <= h_nr - 1) ? grid_at(h_r0, h_dr, r_i + 2) : ri1;
    }
    // wx = (x - rin[i]) / (rin[i+1] - rin[i]) through the cached cell
    __device__ __forceinline__ double weight_r(double x, int& i) const
    {
        if (!(x > r_lo && x <= r_hi)) refill(x);
        i = r_i;
#if PGR_FAST
        return fdiv_y(x - r_lo, r_hi - r_lo, r_yden);
#else
        return (x - r_lo) / (r_hi - r_lo);
#endif
    }

    // bilinear c and dc/dz at (x, z): REF/integration_processes.py:101-174, both tables at once
    __device__ __forceinline__ void lookup(double x, double z, double& c, double& cp) const
    {
        int i;
        double wx = weight_r(x, i);
        lookup_w(wx, i, z, c, cp);
    }
    // the same with the range weight and range cell already known (step_weights)
    // the four corner nodes {c, cp} of the cell of (range cell i, z) and the depth weight: the
    // memory half of a look-up, issued as early as the stage's z is known ...
    struct Fetch {
        double2 v00, v01, v10, v11;
        double wy;
    };
    __device__ __forceinline__ Fetch fetch(int i, double z) const
    {
        Fetch f;
        int j;
        if (ZM == 2 || ZM == 3) {
            // candidate cell j0 (zin[j0] < z <= zin[j0 + 2]) from the bin table, or from the
            // quadratic index estimate of a smooth grid; the three nodes from j0 on are fetched
            // together, then z > zin[j0 + 1] picks the upper cell
            int j0;
            if (ZM == 2) {
                const double t = (z - h_zb_z0) * h_zb_inv_w;
                const int k = min(max((int)floor(t), 0), h_zb_B - 1);  // NaN -> 0, like cell_search
                j0 = lds_zb[k];
            } else {
                const double u = (z - h_zb_z0) * h_zq_inv_span;
                const double g = h_zq_c0 + u * (h_zq_c1 + u * h_zq_c2);
                j0 = min(max((int)floor(g - 0.5), 0), h_nz - 2);      // NaN -> 0
            }
            const int j2 = min(j0 + 2, h_nz - 1);
            const double za = lds_z[j0], zb = lds_z[j0 + 1], zc = lds_z[j2];
            const bool up = (z > zb) & (j0 + 1 <= h_nz - 2);
            j = j0 + (up ? 1 : 0);
            const double zj = up ? zb : za, zj1 = up ? zc : zb;
            f.wy = fdiv(z - zj, zj1 - zj);
            if (LDS_TAB) {
                const double2 t0 = lds[j0], t1 = lds[j0 + 1], t2 = lds[j2];
                f.v00 = up ? t1 : t0;
                f.v01 = up ? t2 : t1;
                f.v10 = f.v00; f.v11 = f.v01;
                return f;
            }
        } else if (ZS) {
            const double t = (ZM == 4) ? z : z * h_inv_dz;  // exact
            // searchsorted(side='left') puts a z that IS a node into the cell above it (weight 1);
            // the cell below it (weight 0) blends to the same bits: the node's two products, each
            // rounded once, plus exact zeros.  So the truncating conversion serves (trunc = floor
            // for t >= 0, clamped to 0 below; v_cvt_i32_f64 saturates and maps NaN to 0), and the
            // clamp is one v_med3_i32: 4 instructions per look-up instead of 7.
            j = clamp_index((int)t, h_nz - 2);
            // (z - zin[j]) / dz, an exact scaling (by 1 when ZM == 4: the same bits without the multiplies)
            f.wy = (ZM == 4) ? (z - (double)j) : (z - (double)j * h_dz) * h_inv_dz;
        } else {
            double zj, zj1;
            j = cell_z(z, zj, zj1);
            // every cell exactly dz wide and dz a power of two: the division is an exact scaling
            f.wy = h_z_pow2 ? (z - zj) * h_inv_dz : fdiv(z - zj, zj1 - zj);
        }
        if (LDS_TAB) {
            f.v00 = lds[j];
            f.v01 = lds[j + 1];
            f.v10 = f.v00;  // range independent: rows are bitwise identical
            f.v11 = f.v01;
        } else {
            // (the table pointer comes out of the descriptor, i.e. out of memory, so the compiler takes it for a
            // generic pointer: flat_load + a wait on both counters; it IS global memory)
            typedef double __attribute__((ext_vector_type(2))) d2v;
            typedef const d2v __attribute__((address_space(1))) * GlobalTab;
            const GlobalTab row = (GlobalTab)h_tab + (size_t)i * h_row_stride + j;
            const d2v t00 = row[0], t01 = row[1], t10 = row[h_row_stride], t11 = row[h_row_stride + 1];
            f.v00 = make_double2(t00.x, t00.y);
            f.v01 = make_double2(t01.x, t01.y);
            f.v10 = make_double2(t10.x, t10.y);
            f.v11 = make_double2(t11.x, t11.y);
        }
        return f;
    }
    // ... and the arithmetic half: the reference's four-corner blend
    __device__ __forceinline__ void blend(const Fetch& f, double wx, double& c, double& cp) const
    {
        const double wy = f.wy;
        double a = (1 - wx) * (1 - wy), b = wx * (1 - wy), cc = (1 - wx) * wy, d = wx * wy;
        c = a * f.v00.x + b * f.v10.x + cc * f.v01.x + d * f.v11.x;
        cp = a * f.v00.y + b * f.v10.y + cc * f.v01.y + d * f.v11.y;
    }
    __device__ __forceinline__ void lookup_w(double wx, int i, double z, double& c, double& cp) const
    {
        const Fetch f = fetch(i, z);
        blend(f, wx, c, cp);
    }

    // bathymetry under the ray: linear_interp, REF/integration_processes.py:177-235
    __device__ __forceinline__ double bathy(double x) const
    {
        int i;
        return bathy(x, i);
    }
    __device__ __forceinline__ double bathy(double x, int& i) const
    {
        double xi, xi1;
        if (h_b_uniform) {
            i = cell_uniform(x, h_b0, h_db, h_inv_db, h_nb);
            xi = grid_at(h_b0, h_db, i);
            xi1 = grid_at(h_b0, h_db, i + 1);
        } else {
            i = cell_search(x, bx, h_nb);
            xi = bx[i];
            xi1 = bx[i + 1];
        }
        double w = fdiv(x - xi, xi1 - xi);
        return (1 - w) * bd[i] + w * bd[i + 1];
    }

    // derivsrd, REF/integration_processes.py:26-98 (clamp: Q8)
    __device__ __forceinline__ void rhs(double x, double z, double pz, double& d0, double& d1,
                                        double& d2, double& c) const
    {
        int i;
        double wx = weight_r(x, i);
        rhs_w(wx, i, z, pz, d0, d1, d2, c);
    }
    __device__ __forceinline__ void rhs_w(double wx, int i, double z, double pz, double& d0, double& d1,
                                          double& d2, double& c) const
    {
        rhs_f(fetch(i, z), wx, pz, d0, d1, d2, c);
    }
    __device__ __forceinline__ void rhs_f(const Fetch& ft, double wx, double pz, double& d0, double& d1,
                                          double& d2, double& c) const
    {
        double cp;
        blend(ft, wx, c, cp);
        double arg = 1.0 - (c * c) * (pz * pz);
#if PGR_FAST
        // `if arg <= 0: arg = 1e-30` as one v_max_f64: 1 - x is 0, negative or >= 2^-53, never in
        // (0, 1e-30).  (A NaN arg -- c or pz NaN -- becomes 1e-30 here; d1 and d2 are NaN through
        // their own factors all the same, and the error norm with them.)
        arg = fmax(arg, k_tiny);
#else
        if (arg <= 0.0) arg = 1e-30;
#endif
#if PGR_FAST
        double fact = frsqrt(arg);
        double rc = frcp_seed(c);  // seeds both quotients below (1/c and, squared, 1/c^2)
        d0 = fdiv_y(fact, c, rc);
        d1 = c * pz * fact;
        d2 = fdiv_y(-fact * cp, c * c, rc * rc);
#else
        double fact = 1 / sqrt(arg);
        d0 = fact / c;
        d1 = c * pz * fact;
        d2 = -fact * cp / (c * c);
#endif
    }

    // Range weights (and cells) of the five stage abscissae x_s = t + C_s h of ONE step attempt,
    // C = (1/5, 3/10, 4/5, 8/9, 1), bitwise what weight_r(x_s) returns.  x only moves forward and a
    // step is short against a range cell, so nearly always every x_s lies in the cached cell: five
    // multiplies by the cached reciprocal, no test per stage (a skipped refill block is a taken
    // branch: ~80 cycles, six per trip).  Steps that straddle the cell's upper edge take ONE block
    // per attempt: uniform rin -> the next cell is known in closed form and each x_s selects its
    // cell; anything else (non-uniform rin, a step wider than two cells, a cache that an event
    // search left elsewhere) goes stage by stage through weight_r.
    __device__ __forceinline__ void step_weights(double t, double h, double (&w)[5], int (&ic)[5]) const
    {
        const double xs[5] = {t + k_c2 * h, t + k_c3 * h, t + k_c4 * h, t + k_c5 * h, t + 1.0 * h};
        if (__builtin_expect((t >= r_lo) & (xs[4] <= r_hi), 1)) {
            const double den = r_hi - r_lo;
#pragma unroll
            for (int s = 0; s < 5; s++) {
#if PGR_FAST
                w[s] = fdiv_y(xs[s] - r_lo, den, r_yden);
#else
                w[s] = (xs[s] - r_lo) / den;
#endif
                ic[s] = r_i;
            }
        } else {
            // the committed t has left the cached cell: step the cache to the next cell
            if (h_r_uniform && (t > r_hi) && (t <= r_hi2) && (r_hi2 > r_hi)) {
                r_lo = r_hi; r_hi = r_hi2; r_i++;
                r_yden = frcp(r_hi - r_lo);
                r_hi2 = (r_i + 2 <= h_nr - 1) ? grid_at(h_r0, h_dr, r_i + 2) : r_hi;
            }
            if (h_r_uniform && (t >= r_lo) && (t <= r_hi) && (xs[4] <= r_hi2)) {
                const double a_den = r_hi - r_lo, b_den = r_hi2 - r_hi;
                const double b_yden = frcp(b_den);
#pragma unroll
                for (int s = 0; s < 5; s++) {
                    const bool in_b = xs[s] > r_hi;
                    const double lo = in_b ? r_hi : r_lo, den = in_b ? b_den : a_den, yd = in_b ? b_yden : r_yden;
#if PGR_FAST
                    w[s] = fdiv_y(xs[s] - lo, den, yd);
#else
                    w[s] = (xs[s] - lo) / den;
#endif
                    ic[s] = r_i + (in_b ? 1 : 0);
                }
            } else {
#pragma unroll
                for (int s = 0; s < 5; s++) w[s] = weight_r(xs[s], ic[s]);
            }
        }
    }

    // the four +-1 events (REF/integration_processes.py:238-303) as a bit mask, bit k = event
    // k is +1.  theta = degrees(arcsin(p c)): theta < 0 <=> -1 <= pc < 0 (NaN when |pc| > 1, Q7).
    __device__ __forceinline__ unsigned events(double x, double z, double pz, double c) const
    {
        double pc = pz * c;
        unsigned g = ((z < 0) & (pc < 0) & (pc >= -1.0)) ? 1u : 0u;
        // bottom: z > bathy(x) is impossible while z is above the shallowest bathymetry node
        // (minus a margin for the interpolation's rounding) and x is inside the bathymetry table;
        // vertical: only |pc| within 2e-10 of 1 can reach 90 - 1e-3 degrees.  Both tests sit in ONE
        // rarely entered block: every skipped block is a taken branch on the step's critical path.
        // (without a declared span: the full test)
        bool above = (z < zmin_eff);
        if (__builtin_expect(!x_guard, 0)) above = (z < e.b_zmin) & (x >= e.b_xlo) & (x <= e.b_xhi);
        const bool near_bottom = (pc > 0) & (pc <= 1.0) & !above;
        const bool near_vertical = (fabs(pc) > k_vert) & (fabs(pc) <= 1.0);
        if (near_bottom | near_vertical) {
            if (near_bottom) {
                if (z > bathy(x)) g |= 2u;
            }
            if (near_vertical) {
                double th = pgr_cr_asin(pc) * (180.0 / M_PI);
                if (fabs(th) > (90 - 1e-3)) g |= 4u;
            }
        }
        bool outside = (z > h_zhi_tol) | (z < h_zlo_tol);
        if (__builtin_expect(!x_guard, 0)) outside |= (x < e.rlo_tol) | (x > e.rhi_tol);
        if (outside) g |= 8u;
        return g;
    }
};

__device__ __forceinline__ double rms3(double a, double b, double c, double sqrt3 = 1.7320508075688772,
                                       double inv_sqrt3 = 0.57735026918962584)
{
    // np.linalg.norm(x) / x.size ** 0.5, SCIPY/common.py:63-65
#if PGR_FAST
    // x / 3**0.5 with the (correctly rounded) reciprocal of the constant as Markstein seed
    return fdiv_y(fsqrt(a * a + b * b + c * c), sqrt3, inv_sqrt3);
#else
    return sqrt(a * a + b * b + c * c) / 1.7320508075688772;
#endif
}

// quartic dense output of one accepted step: Q = K.T @ P (SCIPY/rk.py:178-180, 393-404)
struct Dense {
    double h;
    double q[3][4];
    // (t_old, y_old) are the lane's still-uncommitted (t, y): passed in, not duplicated
    __device__ __forceinline__ void eval(double t_old, double y0, double y1, double y2, double t,
                                         double& o0, double& o1, double& o2) const
    {
        // SCIPY/rk.py:560-574: x = (t - t_old)/h ; p = cumprod ; y = h * (Q @ p) + y_old
        double x = fdiv(t - t_old, h);
        double p1 = x, p2 = p1 * x, p3 = p2 * x, p4 = p3 * x;
        o0 = h * (q[0][0] * p1 + q[0][1] * p2 + q[0][2] * p3 + q[0][3] * p4) + y0;
        o1 = h * (q[1][0] * p1 + q[1][1] * p2 + q[1][2] * p3 + q[1][3] * p4) + y1;
        o2 = h * (q[2][0] * p1 + q[2][1] * p2 + q[2][2] * p3 + q[2][3] * p4) + y2;
    }
};

#define PQ(k1, k3, k4, k5, k6, k7, j)                                                   \
    ((k1) * P1##j + (k3) * P3##j + (k4) * P4##j + (k5) * P5##j + (k6) * P6##j + (k7) * P7##j)
// RK45.P rows (row 2 is all zero), columns 0..3
#define P10 1.0
#define P11 (-8048581381.0 / 2820520608)
#define P12 (8663915743.0 / 2820520608)
#define P13 (-12715105075.0 / 11282082432)
#define P30 0.0
#define P31 (131558114200.0 / 32700410799)
#define P32 (-68118460800.0 / 10900136933)
#define P33 (87487479700.0 / 32700410799)
#define P40 0.0
#define P41 (-1754552775.0 / 470086768)
#define P42 (14199869525.0 / 1410260304)
#define P43 (-10690763975.0 / 1880347072)
#define P50 0.0
#define P51 (127303824393.0 / 49829197408)
#define P52 (-318862633887.0 / 49829197408)
#define P53 (701980252875.0 / 199316789632)
#define P60 0.0
#define P61 (-282668133.0 / 205662961)
#define P62 (2019193451.0 / 616988883)
#define P63 (-1453857185.0 / 822651844)
#define P70 0.0
#define P71 (40617522.0 / 29380423)
#define P72 (-110615467.0 / 29380423)
#define P73 (69997945.0 / 29380423)

// Q = K.T @ P of the step just taken (SCIPY/rk.py:552-556); column 0 of P is e_1, so
// K.T @ P[:, 0] = K1 exactly (the other terms are +0.0)
#define PGR_FORM_Q()                                                                    \
    do {                                                                                \
        D.h = h;                                                                        \
        D.q[0][0] = f0;                                                                 \
        D.q[0][1] = PQ(f0, k30, k40, k50, k60, k70, 1);                                 \
        D.q[0][2] = PQ(f0, k30, k40, k50, k60, k70, 2);                                 \
        D.q[0][3] = PQ(f0, k30, k40, k50, k60, k70, 3);                                 \
        D.q[1][0] = f1;                                                                 \
        D.q[1][1] = PQ(f1, k31, k41, k51, k61, k71, 1);                                 \
        D.q[1][2] = PQ(f1, k31, k41, k51, k61, k71, 2);                                 \
        D.q[1][3] = PQ(f1, k31, k41, k51, k61, k71, 3);                                 \
        D.q[2][0] = f2;                                                                 \
        D.q[2][1] = PQ(f2, k32, k42, k52, k62, k72, 1);                                 \
        D.q[2][2] = PQ(f2, k32, k42, k52, k62, k72, 2);                                 \
        D.q[2][3] = PQ(f2, k32, k42, k52, k62, k72, 3);                                 \
    } while (0)

// rk_step, SCIPY/rk.py:14-71 (K1 = f by FSAL): the six new stages of ONE attempt from (T_, y, f)
// with step H_ -- defines k2*..k7* (k20 = dT/dx, k21 = dz/dx, k22 = dp/dx of stage 2, ...), y_new =
// (n0, n1, n2) and c_new = c at (T_ + H_, y_new).  Used by the step attempt and, with the same
// (t, y, f, h), by the service phase of a lane that parked on this step: IEEE arithmetic in a
// fixed order, so the replay reproduces every bit and nothing has to be kept while parked.
#define PGR_SB() __builtin_amdgcn_sched_barrier(0)
// -DPGR_TIMING (experiments only): s_memtime stamps along one step attempt; the time between stamp
// k-1 and stamp k accumulates in tacc[k] and comes back in n_rej[] of lanes 0..23 (scripts/phase_times.py)
#ifdef PGR_TIMING
#define PGR_STAMP(k)                                                                                 \
    do {                                                                                             \
        unsigned long long _t;                                                                       \
        PGR_SB();                                                                                    \
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t) : : "memory"); \
        tacc[k] += (unsigned)_t - tprev;                                                             \
        tprev = (unsigned)_t;                                                                        \
        PGR_SB();                                                                                    \
    } while (0)
#else
#define PGR_STAMP(k) do { } while (0)
#endif
#define PGR_RK_STAGES(T_, H_)                                                                        \
    double k20, k21, k22, k30, k31, k32, k40, k41, k42, k50, k51, k52, k60, k61, k62, k70, k71, k72, \
        cs;                                                                                          \
    double wr[5];                                                                                    \
    int ir[5];                                                                                       \
    PGR_STAMP(1);                                                                                    \
    C.step_weights(T_, H_, wr, ir);                                                                  \
    PGR_STAMP(2);                                                                                    \
    /* Every sum over stages -- sum_j A[s][j] K_j, K.T @ B, K.T @ E -- is accumulated term by term   \
       as each K_j arrives: the same additions in the same order as SciPy's dot products.  The       \
       terms that the NEXT stage does not need sit between the issue of that stage's table read      \
       (fetch) and its first use (rhs_f), fenced by scheduling barriers: ~90 cycles of read latency  \
       per stage that a single in-order wave would otherwise idle through. */                       \
    const double zs2 = y1 + (f1 * vA21) * (H_), ps2 = y2 + (f2 * vA21) * (H_);                         \
    PGR_STAMP(3);                                                                                    \
    const auto ft2 = C.fetch(ir[0], zs2);                                                            \
    PGR_SB();                                                                                        \
    double a31 = f1 * vA31, a32 = f2 * vA31, a41 = f1 * vA41, a42 = f2 * vA41, a51 = f1 * vA51,           \
           a52 = f2 * vA51, a61 = f1 * vA61, a62 = f2 * vA61;                                           \
    double bs0 = f0 * vB1, bs1 = f1 * vB1, bs2 = f2 * vB1, es0 = f0 * vE1, es1 = f1 * vE1, es2 = f2 * vE1; \
    PGR_SB();                                                                                        \
    C.rhs_f(ft2, wr[0], ps2, k20, k21, k22, cs);                                                     \
    PGR_STAMP(4);                                                                                    \
    a31 = a31 + k21 * vA32; a32 = a32 + k22 * vA32;                                                    \
    const double zs3 = y1 + a31 * (H_), ps3 = y2 + a32 * (H_);                                       \
    PGR_STAMP(5);                                                                                    \
    const auto ft3 = C.fetch(ir[1], zs3);                                                            \
    PGR_SB();                                                                                        \
    a41 = a41 + k21 * vA42; a42 = a42 + k22 * vA42;                                                    \
    a51 = a51 + k21 * vA52; a52 = a52 + k22 * vA52;                                                    \
    a61 = a61 + k21 * vA62; a62 = a62 + k22 * vA62;                                                    \
    PGR_SB();                                                                                        \
    C.rhs_f(ft3, wr[1], ps3, k30, k31, k32, cs);                                                     \
    PGR_STAMP(6);                                                                                    \
    a41 = a41 + k31 * vA43; a42 = a42 + k32 * vA43;                                                    \
    const double zs4 = y1 + a41 * (H_), ps4 = y2 + a42 * (H_);                                       \
    PGR_STAMP(7);                                                                                    \
    const auto ft4 = C.fetch(ir[2], zs4);                                                            \
    PGR_SB();                                                                                        \
    a51 = a51 + k31 * vA53; a52 = a52 + k32 * vA53;                                                    \
    a61 = a61 + k31 * vA63; a62 = a62 + k32 * vA63;                                                    \
    bs0 = bs0 + k30 * vB3; bs1 = bs1 + k31 * vB3; bs2 = bs2 + k32 * vB3;                                \
    es0 = es0 + k30 * vE3; es1 = es1 + k31 * vE3; es2 = es2 + k32 * vE3;                                \
    PGR_SB();                                                                                        \
    C.rhs_f(ft4, wr[2], ps4, k40, k41, k42, cs);                                                     \
    PGR_STAMP(8);                                                                                    \
    a51 = a51 + k41 * vA54; a52 = a52 + k42 * vA54;                                                    \
    const double zs5 = y1 + a51 * (H_), ps5 = y2 + a52 * (H_);                                       \
    PGR_STAMP(9);                                                                                    \
    const auto ft5 = C.fetch(ir[3], zs5);                                                            \
    PGR_SB();                                                                                        \
    a61 = a61 + k41 * vA64; a62 = a62 + k42 * vA64;                                                    \
    bs0 = bs0 + k40 * vB4; bs1 = bs1 + k41 * vB4; bs2 = bs2 + k42 * vB4;                                \
    es0 = es0 + k40 * vE4; es1 = es1 + k41 * vE4; es2 = es2 + k42 * vE4;                                \
    PGR_SB();                                                                                        \
    C.rhs_f(ft5, wr[3], ps5, k50, k51, k52, cs);                                                     \
    PGR_STAMP(10);                                                                                   \
    a61 = a61 + k51 * vA65; a62 = a62 + k52 * vA65;                                                    \
    const double zs6 = y1 + a61 * (H_), ps6 = y2 + a62 * (H_);                                       \
    PGR_STAMP(11);                                                                                   \
    const auto ft6 = C.fetch(ir[4], zs6);                                                            \
    PGR_SB();                                                                                        \
    bs0 = bs0 + k50 * vB5; bs1 = bs1 + k51 * vB5; bs2 = bs2 + k52 * vB5;                                \
    es0 = es0 + k50 * vE5; es1 = es1 + k51 * vE5; es2 = es2 + k52 * vE5;                                \
    PGR_SB();                                                                                        \
    C.rhs_f(ft6, wr[4], ps6, k60, k61, k62, cs);                                                     \
    PGR_STAMP(12);                                                                                   \
    /* y_new = y + h * (K[:-1].T @ B)   (B[1] = 0) */                                                \
    bs1 = bs1 + k61 * vB6; bs2 = bs2 + k62 * vB6;                                                      \
    const double n1 = y1 + (H_) * bs1, n2 = y2 + (H_) * bs2;                                         \
    /* f_new at t + h: the stage-6 abscissa */                                                       \
    PGR_STAMP(13);                                                                                   \
    const auto ft7 = C.fetch(ir[4], n1);                                                             \
    PGR_SB();                                                                                        \
    bs0 = bs0 + k60 * vB6;                                                                            \
    const double n0 = y0 + (H_) * bs0;                                                               \
    es0 = es0 + k60 * vE6; es1 = es1 + k61 * vE6; es2 = es2 + k62 * vE6;                                \
    PGR_SB();                                                                                        \
    double c_new;                                                                                    \
    C.rhs_f(ft7, wr[4], n2, k70, k71, k72, c_new);                                                   \
    PGR_STAMP(14);                                                                                   \
    /* K.T @ E complete (E[1] = 0), SCIPY/rk.py:106-110 */                                           \
    es0 = es0 + k70 * vE7; es1 = es1 + k71 * vE7; es2 = es2 + k72 * vE7

// the save grid np.linspace(x0, x1, S): either recomputed per index exactly as NumPy does
// (arange(S) * step + start, last point forced to x1 -- verified bitwise on the host) or loaded
struct SaveGrid {
    const double* r;
    double x0, x1, step;
    int S, formula;
    __device__ __forceinline__ double at(int j) const
    {
        if (formula) return (j >= S - 1) ? x1 : grid_at(x0, step, j);
        return r[j];
    }
    // np.argmin(np.abs(range_save - t)) (first minimum), REF/launch_rays.py:766-767
    // (inv_step = (S - 1) / (x1 - x0): a guess, the search around it decides)
    __device__ __forceinline__ int nearest(double t, double inv_step) const
    {
        double g = (t - x0) * inv_step;
        g = fmin(fmax(g, 0.0), (double)(S - 1));
        int j = (int)rint(g);
        int best = max(j - 1, 0);
        double bd = fabs(at(best) - t);
        for (int k = best + 1; k <= min(j + 1, S - 1); k++) {
            double d = fabs(at(k) - t);
            if (d < bd) { bd = d; best = k; }
        }
        return best;
    }
};

// ------------------------------------------------------------------------------------
// the fan kernel
//
// Lock-step structure.  Every lane owns one ray.  One trip of the main loop is one RK45
// step ATTEMPT for every lane that is "stepping" (accept/reject is a per-lane select, so the
// 6 right-hand-side evaluations run convergently).  Everything that happens only at a bounce
// -- locating the event on the dense output, re-sampling up to it, the reflection law and the
// restart of the integrator (2 more RHS evaluations, pow, asin/sin) -- costs about two step
// attempts and would run with one or two live lanes per trip if it were done on the spot.
// Instead a lane that accepted a step with an active event PARKS: it keeps (t, y, f) of the
// step's start, the step's end and the fired events, and stops stepping until the wave runs a
// SERVICE phase for all parked lanes together (when `park_lanes` lanes wait, or the oldest has
// waited `park_trips` trips, or nobody else can step); the service replays the step's stages to
// get the dense output back.  Per-ray arithmetic is unchanged by when the service runs.
//
// Control flow inside a trip is kept free of skipped blocks (a taken skip-branch costs a lone
// wave ~80 cycles): selects where both sides are cheap, ONE block per kind of rare work, and
// the memory half of each table look-up issued early with independent work behind it.
// ------------------------------------------------------------------------------------
// SAVE: 0 = end state only (no sample code); 1 = trajectories on a grid that IS np.linspace (recomputed
// per index, no loads) with the default sample evaluation; 2 = any grid / PGR_EXACT_SAMPLES
template <bool LDS_TAB, int ZM, int SAVE>
__global__ void __launch_bounds__(512)
pgr_fan_kernel(const EnvDev* __restrict__ env_p, FanArgs a)
{
    // the environment descriptor lives in device memory: its ~50 dwords would otherwise occupy
    // half the wave's SGPRs as kernel arguments and push the Runge-Kutta tableau (60 fp64
    // literals) into constant re-materialisation + SGPR spills inside the step loop
    const EnvDev& env = *env_p;
    extern __shared__ double2 lds_tab[];
    // LDS layout: [{c, cp}[nz] when LDS_TAB][zin[nz] when ZM >= 2, zbucket[zb_B] when ZM == 2][bathymetry]
    double* const lds_after_tab = (double*)(lds_tab + (LDS_TAB ? env.nz : 0));
    double* const lds_z = lds_after_tab;
    unsigned short* const lds_zb = (unsigned short*)(lds_z + env.nz);
    if (LDS_TAB) {
        // stage the single depth profile {c, cp}[nz] into LDS (coalesced 16 B per lane)
        for (int j = threadIdx.x; j < env.nz; j += blockDim.x) lds_tab[j] = env.tab[j];
    }
    if (ZM == 2 || ZM == 3) {
        for (int j = threadIdx.x; j < env.nz; j += blockDim.x) lds_z[j] = env.zin[j];
        if (ZM == 2) for (int j = threadIdx.x; j < env.zb_B; j += blockDim.x) lds_zb[j] = env.zbucket[j];
    }
    // the bathymetry under a deep ray is looked up every step: {depth_ranges, depths} in LDS too
    double* const lds_bx = (a.bathy_lds_off >= 0) ? (double*)((char*)lds_tab + a.bathy_lds_off) : nullptr;
    if (lds_bx) {
        for (int j = threadIdx.x; j < env.nb; j += blockDim.x) {
            lds_bx[j] = env.depth_ranges[j];
            lds_bx[env.nb + j] = env.depths[j];
        }
    }
    __syncthreads();
    const Ctx<LDS_TAB, ZM> C(env, lds_tab, lds_z, lds_zb, lds_bx);
    C.declare_span(a.x0, a.x1);
    // waves are dealt to workgroups round-robin (wave w of block b = global wave w*grid + b):
    // neighbouring launch angles cost alike, so a strided deal balances the CUs
    int64_t gwave = (int64_t)(threadIdx.x >> 6) * gridDim.x + blockIdx.x;
    if (a.wave_map) {
        // cost-aware scheduling (pgr_wave_place): slot -> wave (-1 = slot left empty) and the
        // wave's issue priority in bits 28..29: the costlier wave of a SIMD's pair runs at its own
        // pace, the cheaper one fills the issue slots it leaves
        int m = a.wave_map[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)];
        gwave = (m < 0) ? -1 : (m & 0x0fffffff);
        int prio = __builtin_amdgcn_readfirstlane((m < 0) ? 0 : ((m >> 28) & 3));
        if (prio == 3) __builtin_amdgcn_s_setprio(3);
        else if (prio == 2) __builtin_amdgcn_s_setprio(2);
        else if (prio == 1) __builtin_amdgcn_s_setprio(1);
    }
    const int64_t ray = gwave * 64 + (threadIdx.x & 63);
    const bool valid = (gwave >= 0) && (ray < a.N);
    double SAFETY = 0.9, MIN_FACTOR = 0.2, MAX_FACTOR = 10,
           SQRT3 = 1.7320508075688772, INV_SQRT3 = 0.57735026918962584,
           POW_FIFTH = 0.2, POW_KLN2 = PGR_CR_POW_KLN2;
    const double rtol = a.rtol, atol = a.atol, t_bound = a.x1;
    const int S = a.S;
    constexpr bool save = (SAVE != 0);  // trajectories wanted (a.T != nullptr)
    const bool exact_samples = (SAVE == 2) && (a.flags & PGR_EXACT_SAMPLES) != 0;
    // (max_steps <= 2^30, host checked; the counters are ints)
    const int attempt_limit = (int)((4 * a.max_steps + 4096 < 0x7fffffff) ? 4 * a.max_steps + 4096 : 0x7fffffff);
    // PGR_STORED_SIGN: trajectories leave as pygenray stores them, z -> -z and p -> -p
    // (REF/ray_objects.py:51-52): a sign-bit xor, exact, and two host passes over 1.6 GB less
    const unsigned long long sgn = (a.flags & PGR_STORED_SIGN) ? 0x8000000000000000ULL : 0ULL;
#define SGN(v) __longlong_as_double(__double_as_longlong(v) ^ (long long)sgn)
    SaveGrid G;
    G.r = a.r_save; G.x0 = a.x0; G.x1 = a.x1; G.step = a.save_step;
    G.S = S; G.formula = (SAVE == 1) ? 1 : a.save_formula;

    double t = a.x0, y0 = 0, y1 = 0, y2 = 0;
    if (valid) {
        y0 = a.y0[3 * ray + 0];
        y1 = a.y0[3 * ray + 1];
        y2 = a.y0[3 * ray + 2];
    }
    double f0 = 0, f1 = 0, f2 = 0, h_abs = 0;
    unsigned g = 0;
    int status = valid ? RUNNING : PGR_RAY_OK;
    if (valid && (a.flags & PGR_SKIP_NAN_Y0) && (y2 != y2)) status = PGR_RAY_SKIPPED;  // a parked eigenray bracket
    bool need_init = true, rejected = false, parked = false;
    int nb = 0, ns = 0, n_steps = 0, n_rej = 0;
    int jnext = 0;
    double rnext = 0;
    // a parked lane keeps only the end of its step and which events fired; the service phase
    // replays the step (PGR_RK_STAGES) to get its dense output back
    double pk_tnew = 0;
    unsigned pk_active = 0;
    int waited = 0;
    int trips = 0, services = 0, fallbacks = 0;  // diagnostics (PGR_DEBUG_TRIPS)
    const int max_steps32 = (int)(a.max_steps < 0x7fffffff ? a.max_steps : 0x7fffffff);  // n_steps is an int
#ifdef PGR_TIMING
    unsigned tacc[24];
    for (int k = 0; k < 24; k++) tacc[k] = 0;
    unsigned tprev = (unsigned)clock64();
#endif
    const int64_t out_off = ray * a.stride_ray;
#define Tp (a.T + out_off)
#define Zp (a.Z + out_off)
#define Pp (a.P + out_off)

    // The 26 tableau coefficients of the stage sums live in VGPRs for the whole kernel (the kernel
    // needs ~155 of its 256 VGPRs otherwise): an fp64 literal cannot be an inline operand, so each use
    // cost two s_mov_b32 -- 61 SALU instructions per attempt that a lone wave cannot overlap.  The
    // empty asm hides the value from constant propagation.
#define PGR_VCONST(n) double v##n = n; asm volatile("" : "+v"(v##n))
    PGR_VCONST(A21);
    PGR_VCONST(A31);
    PGR_VCONST(A32);
    PGR_VCONST(A41);
    PGR_VCONST(A42);
    PGR_VCONST(A43);
    PGR_VCONST(A51);
    PGR_VCONST(A52);
    PGR_VCONST(A53);
    PGR_VCONST(A54);
    PGR_VCONST(A61);
    PGR_VCONST(A62);
    PGR_VCONST(A63);
    PGR_VCONST(A64);
    PGR_VCONST(A65);
    PGR_VCONST(B1);
    PGR_VCONST(B3);
    PGR_VCONST(B4);
    PGR_VCONST(B5);
    PGR_VCONST(B6);
    PGR_VCONST(E1);
    PGR_VCONST(E3);
    PGR_VCONST(E4);
    PGR_VCONST(E5);
    PGR_VCONST(E6);
    PGR_VCONST(E7);
#undef PGR_VCONST
    // ... and, in the kernels that save trajectories, the 18 coefficients of the stage-major sample form
#ifndef PGR_PIN_P   // (not where the depth search already fills the register file: those instances spill otherwise)
#define PGR_PIN_P (SAVE != 0 && ZM != 0 && ZM != 3)
#endif
#define PGR_VCONST_IF(c, n) double v##n = n; if (c) asm volatile("" : "+v"(v##n))
    PGR_VCONST_IF(PGR_PIN_P, P11);
    PGR_VCONST_IF(PGR_PIN_P, P12);
    PGR_VCONST_IF(PGR_PIN_P, P13);
    PGR_VCONST_IF(PGR_PIN_P, P31);
    PGR_VCONST_IF(PGR_PIN_P, P32);
    PGR_VCONST_IF(PGR_PIN_P, P33);
    PGR_VCONST_IF(PGR_PIN_P, P41);
    PGR_VCONST_IF(PGR_PIN_P, P42);
    PGR_VCONST_IF(PGR_PIN_P, P43);
    PGR_VCONST_IF(PGR_PIN_P, P51);
    PGR_VCONST_IF(PGR_PIN_P, P52);
    PGR_VCONST_IF(PGR_PIN_P, P53);
    PGR_VCONST_IF(PGR_PIN_P, P61);
    PGR_VCONST_IF(PGR_PIN_P, P62);
    PGR_VCONST_IF(PGR_PIN_P, P63);
    PGR_VCONST_IF(PGR_PIN_P, P71);
    PGR_VCONST_IF(PGR_PIN_P, P72);
    PGR_VCONST_IF(PGR_PIN_P, P73);
#undef PGR_VCONST_IF
    // ... and the remaining fp64 literals of a step attempt (stage abscissae, controller and norm
    // constants, the RHS clamp): 26 s_mov_b32 per attempt otherwise
#ifndef PGR_PIN_LITERALS
#define PGR_PIN_LITERALS (SAVE == 0 || ZM == 3)
#endif
    if (PGR_PIN_LITERALS) {
#define PGR_PIN(x) asm volatile("" : "+v"(x))
        PGR_PIN(C.k_c2); PGR_PIN(C.k_c3); PGR_PIN(C.k_c4); PGR_PIN(C.k_c5); PGR_PIN(C.k_tiny); PGR_PIN(C.k_vert);
        PGR_PIN(SAFETY); PGR_PIN(MIN_FACTOR); PGR_PIN(MAX_FACTOR);
        PGR_PIN(SQRT3); PGR_PIN(INV_SQRT3);
        PGR_PIN(POW_FIFTH); PGR_PIN(POW_KLN2);
#undef PGR_PIN
    }
    // One trip = one step attempt of every stepping lane, THEN the gate that decides whether the
    // wave services its parked lanes.  (The first trip only runs the gate: every lane starts with
    // need_init.)  The step comes first so that the common path -- nobody parked -- is the loop's
    // fall-through: one skipped block and the back-edge are its only taken branches.
    do {
        bool run, pend;
        unsigned long long pm;
        // inner loop: step attempts while no lane is waiting for service (the common case: its
        // only taken branch is its own back-edge)
        do {
        trips++;
        PGR_STAMP(0);
        if (status == RUNNING && !parked && !need_init) {
            // ---- one attempt of RK45._step_impl, SCIPY/rk.py:111-176 ----
            double min_step = min_step_of(t);
            if (!rejected && h_abs < min_step) h_abs = min_step;  // clamp only on entry
            bool too_small = h_abs < min_step;
            double h = h_abs;
            double t_new = t + h;
            if ((t_new - t_bound) > 0) t_new = t_bound;
            h = t_new - t;
            h_abs = fabs(h);

            PGR_RK_STAGES(t, h);
            // error estimate, SCIPY/rk.py:106-110,146-147  (E[1] = 0)
            double sc0 = atol + fmax(fabs(y0), fabs(n0)) * rtol;
            double sc1 = atol + fmax(fabs(y1), fabs(n1)) * rtol;
            double sc2 = atol + fmax(fabs(y2), fabs(n2)) * rtol;
            double er0 = fdiv(es0 * h, sc0);
            double er1 = fdiv(es1 * h, sc1);
            double er2 = fdiv(es2 * h, sc2);
            double error_norm = rms3(er0, er1, er2, SQRT3, INV_SQRT3);
            PGR_STAMP(15);

            // ---- accept / reject and the next step size, SCIPY/rk.py:148-165, without branches: a
            // taken skip-branch costs a lone in-order wave ~80 cycles (scripts/probes/branch_probe2),
            // as much as 20 fp64 operations, and the slowest wave's latency is the fan's run time.
            // ONE err ** -0.2 serves both outcomes, and SciPy's min / max do the rest -- also where the
            // power is not a number: err = 0 or below the fp32 range of its seed (the true power is
            // huge: MAX_FACTOR, SCIPY/rk.py:153-154), err = inf or NaN (MIN_FACTOR, as np.max / Python's
            // max(MIN_FACTOR, nan) give)
            const bool accepted = !too_small && (error_norm < 1);
            const bool reject = !too_small && !accepted;
            const double pw = SAFETY * pow_m02(error_norm, POW_FIFTH, POW_KLN2);
            double fac_acc = (pw < MAX_FACTOR) ? pw : MAX_FACTOR;
            fac_acc = (rejected && !(fac_acc < 1)) ? 1.0 : fac_acc;
            const double fac_rej = (pw > MIN_FACTOR) ? pw : MIN_FACTOR;
            h_abs = too_small ? h_abs : h_abs * (accepted ? fac_acc : fac_rej);
            rejected = too_small ? rejected : reject;
            n_rej += reject ? 1 : 0;
            const bool over = (n_rej + n_steps) > attempt_limit;
            status = too_small ? PGR_RAY_STEP_TOO_SMALL : ((reject & over) ? PGR_RAY_MAX_STEPS : status);
            PGR_STAMP(16);

            if (accepted) {
                n_steps++;
                // events at the new point, SCIPY/ivp.py:671-675 (c at (t_new, y_new) is the FSAL lookup)
                unsigned g_new = C.events(t_new, n1, n2, c_new);
                PGR_STAMP(17);
                // find_active_events, SCIPY/ivp.py:133-156: values are +-1, so "up" = -1 -> +1,
                // "down" = +1 -> -1; surface/bottom need "up", vertical/bbox take either
                unsigned up = (~g) & g_new, down = g & (~g_new);
                unsigned active = (up & 3u) | ((up | down) & 12u);
                g = g_new;
                bool want_samples = save && (jnext < S - 1) && (rnext <= t_new);
                // samples behind the step (rnext < t: the extrapolated ones a segment's first step
                // owns, Q5, |xi| up to 1e5) amplify rounding by xi^4 and keep SciPy's order
                const bool scipy_order = exact_samples || (rnext < t);
                Dense D;
                if (active) {
                    // park: the step is located, truncated and bounced in the next service phase, which
                    // replays this attempt's stages from (t, y, f) and pk_tnew -- nothing else is kept
                    parked = true;
                    pk_active = active;
                    pk_tnew = t_new;
                } else {
                    // ---- _interpolate_ray, streamed (REF/launch_rays.py:763-772, Q5): samples of the
                    // segment slice [idx1, idx2) that this step's quartic owns ----
                    if (want_samples) {
                        if (scipy_order) {
                            PGR_FORM_Q();
                            while (jnext < S - 1 && rnext <= t_new) {
                                double o0, o1, o2;
                                D.eval(t, y0, y1, y2, rnext, o0, o1, o2);
                                Tp[(int64_t)jnext * a.stride_smp] = o0;
                                Zp[(int64_t)jnext * a.stride_smp] = SGN(o1);
                                Pp[(int64_t)jnext * a.stride_smp] = SGN(o2);
                                jnext++;
                                rnext = G.at(jnext);
                            }
                        } else {
                            // The same quartic summed stage-major, y_old + h * sum_j K_j b_j(xi) with
                            // b_j(xi) = sum_k P[j][k] xi^(k+1), in FMAs: no Q = K.T @ P to form (a third
                            // of the work) and a few ulp from SciPy's summation order.  Output samples
                            // never feed back into the integration, so this cannot move a ray.
                            const double inv_h = frcp_seed(h);  // 2e-15 is plenty for xi (no feedback)
#define PGR_KSUM(k1, k3, k4, k5, k6, k7)                                                          \
    __builtin_fma(k1, b1, __builtin_fma(k3, b3, __builtin_fma(k4, b4, __builtin_fma(k5, b5,      \
                  __builtin_fma(k6, b6, (k7) * b7)))))
// trajectory stores: with the table in HBM/L2 they are streaming (non-temporal) stores, so that
// 2.4 GB of samples per fan do not evict the table rows from L2 (range-dependent fan with
// trajectories 8.5 -> 7.7 ms); with the table in LDS plain stores are faster (5.9 vs 6.3 ms)
#define PGR_SSTORE(v, p)                                                                          \
    do {                                                                                          \
        if (LDS_TAB) *(p) = (v); else __builtin_nontemporal_store((v), (p));                      \
    } while (0)
#define PGR_SAMPLE_LOOP(NEXT)                                                                     \
    while (jnext < S - 1 && rnext <= t_new) {                                                     \
        const double xi = (rnext - t) * inv_h, x2 = xi * xi;                                      \
        const double b1 = xi * __builtin_fma(xi, __builtin_fma(xi, __builtin_fma(xi, vP13, vP12), vP11), P10); \
        const double b3 = x2 * __builtin_fma(xi, __builtin_fma(xi, vP33, vP32), vP31);               \
        const double b4 = x2 * __builtin_fma(xi, __builtin_fma(xi, vP43, vP42), vP41);               \
        const double b5 = x2 * __builtin_fma(xi, __builtin_fma(xi, vP53, vP52), vP51);               \
        const double b6 = x2 * __builtin_fma(xi, __builtin_fma(xi, vP63, vP62), vP61);               \
        const double b7 = x2 * __builtin_fma(xi, __builtin_fma(xi, vP73, vP72), vP71);               \
        PGR_SSTORE(__builtin_fma(h, PGR_KSUM(f0, k30, k40, k50, k60, k70), y0), &Tp[(int64_t)jnext * a.stride_smp]); \
        PGR_SSTORE(SGN(__builtin_fma(h, PGR_KSUM(f1, k31, k41, k51, k61, k71), y1)), &Zp[(int64_t)jnext * a.stride_smp]); \
        PGR_SSTORE(SGN(__builtin_fma(h, PGR_KSUM(f2, k32, k42, k52, k62, k72), y2)), &Pp[(int64_t)jnext * a.stride_smp]); \
        jnext++;                                                                                  \
        rnext = NEXT;                                                                             \
    }
                            // two copies so that the linspace one holds no load: a load in the loop
                            // makes every iteration wait (vmcnt) for the stores of the one before
                            // (every use of rnext is guarded by jnext < S - 1, so the formula copy needs no
                            // select for the forced last grid point)
                            if (SAVE == 1 || G.formula) { PGR_SAMPLE_LOOP(grid_at(G.x0, G.step, jnext)) }
                            else { PGR_SAMPLE_LOOP(G.r[jnext]) }
#undef PGR_SAMPLE_LOOP
#undef PGR_SSTORE
#undef PGR_KSUM
                        }
                    }
                    t = t_new; y0 = n0; y1 = n1; y2 = n2;
                    f0 = k70; f1 = k71; f2 = k72;
                    // (selects, not a skipped block: a taken branch costs more than these four instructions)
                    status = ((t - t_bound) >= 0) ? PGR_RAY_OK  // SCIPY/base.py:197
                                                  : ((n_steps > max_steps32) ? PGR_RAY_MAX_STEPS : status);
                }
                PGR_STAMP(18);
            }
        }
        PGR_STAMP(19);
        run = (status == RUNNING);
        pend = run && (parked || need_init);
        pm = ballot64(pend);
        } while (pm == 0 && ballot64(run) != 0);
        // keep ONE exit of the trip loop: without this the compiler threads "left with pm != 0" straight
        // to the gate and gives the loop two exits, whose unification costs the common path two more
        // taken branches per trip
        asm volatile("" : "+s"(pm));
        if (pm) {
            waited++;
            const bool nobody_steps = ballot64(run && !pend) == 0;
            if (__popcll(pm) >= a.park_lanes || waited > a.park_trips || nobody_steps) {
                waited = 0;
                services++;
                // =========================== SERVICE phase ===========================
                // what only the service needs of the environment descriptor and of the kernel arguments is read
                // HERE, through pointers the compiler cannot trace back (the empty asm): hoisted to the prologue
                // these values sit in SGPRs across the step loop and push loop values out to VGPR lanes
                const EnvDev* es_p = env_p;
                asm volatile("" : "+s"(es_p));
                const EnvDev& es = *es_p;
                const char __attribute__((address_space(4))) * ks_p =
                    (const char __attribute__((address_space(4))) *)__builtin_amdgcn_kernarg_segment_ptr();
                asm volatile("" : "+s"(ks_p));
                const FanArgs __attribute__((address_space(4))) & as = *(const FanArgs __attribute__((address_space(4))) *)(ks_p + 8);
                const double svc_c_lo = es.c_lo, svc_c_hi = es.c_hi;
#ifdef PGR_DBG_REPLAY
                const unsigned long long dbg_s0 = __builtin_amdgcn_s_memtime();
                unsigned long long dbg_s1 = dbg_s0, dbg_s4 = dbg_s0, dbg_s5 = dbg_s0, dbg_r0 = dbg_s0, dbg_r3 = dbg_s0;
                unsigned long long dbg_n1 = dbg_s0, dbg_n2 = dbg_s0, dbg_n3 = dbg_s0, dbg_b1 = dbg_s0, dbg_b2 = dbg_s0;
#endif
                if (pend && parked) {
                    parked = false;
                    const unsigned active = pk_active;
                    // replay the parked attempt: same t, y, f and h = t_new - t as when it ran
                    const double t_new = pk_tnew, h = t_new - t;
                    PGR_RK_STAGES(t, h);
                    (void)n0; (void)c_new; (void)es0; (void)es1; (void)es2;
                    Dense D;
                    PGR_FORM_Q();
#ifdef PGR_DBG_REPLAY
                    dbg_s1 = __builtin_amdgcn_s_memtime();
#endif
                    int ev = -1;
                    double best = 0;
                    // (a step that crosses the surface nearly always also crosses the bounding box's
                    // z = zin[0] - 1e-6 just after it: both events are active, the surface flips first)
                    const bool with_bbox = (active == 9u);
                    const unsigned act = with_bbox ? 1u : active;
                    if ((a.flags & PGR_EXACT_BISECTION) == 0 && (act == 1u || act == 2u)) {
                        // ---- fast event location (default) ----
                        // SciPy's brentq on the +-1 event degenerates to ~42 bisection steps, each a
                        // dense-output + table evaluation.  A bisection's iterates depend only on where
                        // the function flips, and the flip of a surface/bottom event is the zero of the
                        // continuous F(x) = z(x) [- bathy(x)] on the step's quartic.  So: safeguarded
                        // Newton on F; a BAND around its root, wide enough to hold every point where the
                        // rounding noise of the true event's evaluation could decide its value (4 E / |F'|,
                        // E bounding that noise, and at least an ulp of x); the TRUE event at the band's
                        // two edges (must be: not fired / fired); then brentq's own iterates are REPLAYED
                        // (scipy/optimize/Zeros/brentq.c with xtol = rtol = 4 EPS, SCIPY/ivp.py:51-76) with
                        // the function decided by position outside the band and evaluated for real inside
                        // it: the root returned is the one SciPy returns, at 2 + (1..3) event evaluations
                        // instead of 2 x 42.  Anything unexpected falls through to the exact bisection.
                        const bool bottom = (act == 2u);
                        const double q0 = D.q[1][0], q1 = D.q[1][1], q2 = D.q[1][2], q3 = D.q[1][3];
                        double bs = 0, be = 0;
                        int cell_s = 0, cell_e = 0;
                        if (bottom) { bs = C.bathy(t, cell_s); be = C.bathy(t_new, cell_e); }
                        double zb = y1 + h * (q0 + q1 + q2 + q3);
                        double Fa = bottom ? (y1 - bs) : y1;  // F at s = 0: not yet crossed
                        double Fb = bottom ? (zb - be) : zb;  // F at s = 1: crossed
                        // surface: F falls through 0 (z < 0 fires); bottom: F rises (z > bathy fires)
                        bool pre = bottom ? (Fa <= 0 && Fb > 0) : (Fa >= 0 && Fb < 0);
                        double xa = t, xb = t_new;
                        bool live = false;
                        const bool any_bottom = ballot64(bottom) != 0;  // (wave-uniform: skips the bathymetry look-up of a surface-only service)
                        // the sea floor under a step that stays inside one bathymetry cell is its chord: Newton (which
                        // only has to land inside the noise band, checked at its edges below) takes that instead of
                        // a look-up per iterate
                        const bool chord = ballot64(bottom && cell_s != cell_e) == 0;
                        if (pre) {
                            const double bslope = bottom ? (be - bs) : 0.0;  // per unit s
                            double lo = 0.0, hi = 1.0;
                            double sN = fdiv(Fa, Fa - Fb);  // secant start
                            double dFs = 0;
                            // safeguarded Newton: three steps in a row (quadratic convergence from the secant
                            // start: ~1e-3, 1e-6, 1e-12 of the step; the third one moves by less than the
                            // tolerance), more only for a lane that still moves
                            for (int it = 0; it < 16; it++) {
                                double zs = y1 + h * (sN * (q0 + sN * (q1 + sN * (q2 + sN * q3))));
                                double dz = h * (q0 + sN * (2 * q1 + sN * (3 * q2 + sN * 4 * q3)));
                                double F = zs, dF = dz;
                                if (any_bottom) {
                                    double bq;
                                    if (chord) bq = bs + sN * bslope;
                                    else bq = C.bathy(t + sN * h);
                                    F = bottom ? zs - bq : zs;
                                    dF = bottom ? dz - bslope : dz;
                                }
                                dFs = dF;
                                bool crossed = bottom ? (F > 0) : (F < 0);
                                if (crossed) hi = sN; else lo = sN;
                                double sn = sN - F * frcp_seed(dF);
                                // (closed bracket: when F evaluates to exactly 0 the Newton step is
                                // zero, sn == lo, and that is convergence, not an escape)
                                if (!(sn >= lo && sn <= hi)) sn = 0.5 * (lo + hi);
                                double ds = fabs(sn - sN);
                                sN = sn;
                                if (it >= 2 && ballot64(ds * h >= 1e-12 * (1.0 + fabs(t))) == 0) break;
                            }
#ifdef PGR_DBG_REPLAY
                            dbg_n1 = __builtin_amdgcn_s_memtime();
#endif
                            const double xs = t + sN * h;
                            // E: rounding noise of F as the event evaluates it.  z(x) = h (Q p) + y_old: half an ulp
                            // of the result for the last add and ~4 roundings of terms <= |h| sum|Q|; the sea floor
                            // (1 - w) d_i + w d_(i+1): ~3 half-ulps of the depth.  Each is below EPS x (the sum of
                            // the magnitudes); E takes twice that, and the band twice the distance 2 E / |F'| over
                            // which noise of that size could decide the sign -- but never less than the doubles
                            // next to the root.
                            const double E = 2 * DBL_EPSILON * (fabs(y1) + fabs(h) * (fabs(q0) + fabs(q1) + fabs(q2) + fabs(q3)) +
                                                                (bottom ? fabs(bs) + fabs(be) : 0.0));
                            const double nu = 4 * E * fabs(h) / fabs(dFs);
                            xa = fmax(nextafter(xs - nu, -INFINITY), t);
                            xb = fmin(nextafter(xs + nu, INFINITY), t_new);
                            // the TRUE event at x (surface: REF/integration_processes.py:238-250, bottom: :253-266)
                            // on the step's quartic, SciPy's evaluation order (Dense::eval), without the generic
                            // event code's branches: z and p only, c from the table, the predicate
#define PGR_TRUE_EVENT(X_, FIRED_, BBOX_)                                                                        \
    do {                                                                                                         \
        const double xx_ = fdiv((X_) - t, D.h);                                                                  \
        const double e1_ = xx_, e2_ = e1_ * xx_, e3_ = e2_ * xx_, e4_ = e3_ * xx_;                               \
        const double z_ = D.h * (D.q[1][0] * e1_ + D.q[1][1] * e2_ + D.q[1][2] * e3_ + D.q[1][3] * e4_) + y1;   \
        const double pz_ = D.h * (D.q[2][0] * e1_ + D.q[2][1] * e2_ + D.q[2][2] * e3_ + D.q[2][3] * e4_) + y2;  \
        /* theta = degrees(arcsin(p c)) only enters through its sign and through |p c| <= 1 (NaN otherwise,   \
           Q7): with 0 < c <= c_hi (the table's maximum, a margin for the extrapolated sliver above the        \
           surface included) |p| c_hi < 1 settles both from p alone -- no table look-up */                    \
        double pc_ = pz_;                                                                                        \
        if (ballot64(!((svc_c_lo > 0) & (fabs(pz_) * svc_c_hi < 1.0))) != 0) {                                    \
            double c_, cp_;                                                                                      \
            C.lookup((X_), z_, c_, cp_);                                                                         \
            pc_ = pz_ * c_;                                                                                      \
        }                                                                                                        \
        const double bd_ = any_bottom ? C.bathy(X_) : 0.0;                                                       \
        FIRED_ = bottom ? ((pc_ > 0) & (pc_ <= 1.0) & (z_ > bd_)) : ((z_ < 0) & (pc_ < 0) & (pc_ >= -1.0));       \
        BBOX_ = (z_ > C.h_zhi_tol) | (z_ < C.h_zlo_tol) | ((X_) < es.rlo_tol) | ((X_) > es.rhi_tol);             \
    } while (0)
                            bool ga, gb, bbox_a, bbox_b;
#ifdef PGR_DBG_REPLAY
                            asm volatile("" : "+v"(xa), "+v"(xb));
                            dbg_n2 = __builtin_amdgcn_s_memtime();
#endif
                            PGR_TRUE_EVENT(xa, ga, bbox_a);
                            PGR_TRUE_EVENT(xb, gb, bbox_b);
#ifdef PGR_DBG_REPLAY
                            dbg_n3 = __builtin_amdgcn_s_memtime();
#endif
                            (void)bbox_a;
                            // with the bounding-box event also active its flip must lie beyond xb, so
                            // that the surface root is the earlier one (SCIPY/ivp.py:100-131)
                            live = (xa < xb) && !ga && gb && !(with_bbox && bbox_b);
#ifndef PGR_NO_BAND_TABLE
                            // The doubles strictly inside the band: one or two once the noise band is narrower than
                            // an ulp of x (beyond ~100 km).  Evaluate the true event there as well, and if it flips
                            // once the band shrinks to the flip itself, (last double not fired, first double fired):
                            // every iterate of the replay is then decided by its position, no lane waits inside the
                            // band for phase 2 and phase 2 evaluates nothing.  Wider or non-monotone bands stay as
                            // they are and are evaluated iterate by iterate.
                            {
                                const double x1 = next_up(xa), x2 = next_up(x1), x3 = next_up(x2);
                                const int m = !live ? 3 : (x1 >= xb) ? 0 : (x2 >= xb) ? 1 : (x3 >= xb) ? 2 : 3;
                                bool g1 = true, g2 = true, bbox_q;
                                if (ballot64(m == 1 || m == 2) != 0) { PGR_TRUE_EVENT(x1, g1, bbox_q); }
                                if (ballot64(m == 2) != 0) { PGR_TRUE_EVENT(x2, g2, bbox_q); }
                                (void)bbox_q;
                                g1 = (m == 1 || m == 2) ? g1 : true;
                                g2 = (m == 2) ? g2 : true;
                                if (m <= 2 && (g2 || !g1)) {
                                    const double nxa = g1 ? xa : (g2 ? x1 : x2);
                                    xb = g1 ? x1 : (g2 ? x2 : xb);
                                    xa = nxa;
                                }
                            }
#endif
                        }
#ifdef PGR_NO_REPLAY  // experiments: round 1's "a root within brentq's tolerance" (NOT bit-identical)
                        if (live) { best = xb; ev = bottom ? 1 : 0; }
#else
                        // ---- the replay.  brentq's state: cur = the latest iterate, blk = the other end of
                        // the bracket, fcur = the event at cur; it starts from cur = t_new (fired), blk = t.
                        const double xtol = 4 * DBL_EPSILON, brtol = 4 * DBL_EPSILON;
                        double cur = t_new, blk = t;
                        bool fcur = true;
                        int n1dbg = 0;
#ifdef PGR_DBG_REPLAY
                        const unsigned long long dbg_t0 = __builtin_amdgcn_s_memtime();
#endif
                        {
                            // phase 1: the halvings that can neither end the search nor take brentq's minimum
                            // step (|blk - cur| / 2 stays above 4 delta), kept as (not fired end, fired end): 9
                            // instructions each, no tolerance arithmetic.  cur + (blk - cur) / 2 and lo + (hi -
                            // lo) / 2 are the same double when hi - lo is exact (ends within a factor of two of
                            // each other).  An iterate that falls inside the band moves neither end, so the lane
                            // stays where it is (the next round computes the same iterate again) and phase 2
                            // picks it up there.  The wave runs the count every lane can take.
                            const double dmax = (xtol + brtol * fmax(fabs(t), fabs(t_new))) / 2;
                            const bool sterbenz = (t > 0) ? (t_new <= 2 * t) : ((t_new < 0) && (t >= 2 * t_new));
                            int n1 = (live && sterbenz) ? ilogb(h) - ilogb(dmax) - 3 : (live ? 0 : 90);
                            n1 = min(max(n1, 0), 90);
                            {   // the smallest n1 among the lanes in this block (ballots see the active lanes only)
                                int m = 0;
#pragma unroll
                                for (int bit = 64; bit > 0; bit >>= 1)
                                    if (ballot64(n1 < m + bit) == 0) m += bit;
                                n1 = m;
                            }
                            double plo = t, phi = t_new;
                            bool lastc = true;
#pragma unroll 4
                            for (int k = 0; k < n1; k++) {
                                const double nw = __builtin_fma(phi - plo, 0.5, plo);  // (phi - plo) / 2 is exact: one rounding either way
                                const bool ge = (nw >= xb), le = (nw <= xa);
                                phi = ge ? nw : phi;
                                plo = le ? nw : plo;
                                lastc = ge | (lastc & !le);
                            }
                            if (n1 > 0) { cur = lastc ? phi : plo; blk = lastc ? plo : phi; fcur = lastc; n1dbg = n1; }
                        }
                        // phase 2: brentq's loop as it stands (scipy/optimize/Zeros/brentq.c) for the last few
                        // iterations, all lanes in lock step; the true event is evaluated (for the whole wave,
                        // behind a uniform branch) whenever some lane's iterate lies inside the band, and
                        // decides for those lanes.  A lane whose search has ended (|sbis| < delta) stands still.
#ifdef PGR_DBG_REPLAY
                        const unsigned long long dbg_t1 = __builtin_amdgcn_s_memtime();
                        unsigned long long dbg_ev = 0;
#endif
                        int dbg_it = 0;
                        double fcv = fcur ? 1.0 : 0.0;  // the event at cur, as a number (a carried bool costs more)
                        bool done = !live;
                        for (int it = 0; it < 200; it++) {
                            dbg_it++;
                            const double dlt = (xtol + brtol * fabs(cur)) / 2;
                            const double sbis = (blk - cur) / 2;
                            done = !live | (fabs(sbis) < dlt);
                            if (ballot64(!done) == 0) break;
                            const double nw = (fabs(sbis) > dlt) ? cur + sbis : cur + (sbis > 0 ? dlt : -dlt);
                            const bool inside = !done & (nw > xa) & (nw < xb);
                            double fnv = (nw >= xb) ? 1.0 : 0.0;
                            if (ballot64(inside) != 0) {
#ifdef PGR_DBG_REPLAY
                                const unsigned long long dbg_t2 = __builtin_amdgcn_s_memtime();
#endif
                                bool fired, bbox_q;
                                PGR_TRUE_EVENT(nw, fired, bbox_q);
                                (void)bbox_q;
                                fnv = inside ? (fired ? 1.0 : 0.0) : fnv;
#ifdef PGR_DBG_REPLAY
                                dbg_ev += __builtin_amdgcn_s_memtime() - dbg_t2;
#endif
                            }
                            blk = (!done & (fnv != fcv)) ? cur : blk;
                            cur = done ? cur : nw;
                            fcv = done ? fcv : fnv;
                        }
#ifdef PGR_DBG_REPLAY
                        (void)dbg_it; (void)n1dbg;
                        {
                            const unsigned long long dbg_t3 = __builtin_amdgcn_s_memtime();
                            const int ln = threadIdx.x & 63;
                            dbg_r0 = dbg_t0; dbg_r3 = dbg_t3;
                            fallbacks += (ln == 2) ? (int)(dbg_t1 - dbg_t0) : (ln == 3) ? (int)(dbg_t3 - dbg_t1) : (ln == 4) ? (int)dbg_ev : (ln == 5) ? dbg_it : 0;
                        }
#endif
                        if (live && done) { best = cur; ev = bottom ? 1 : 0; }
#endif
                    }
                    if (ev < 0) {
                        fallbacks++;
                        // handle_events + solve_event_equation, SCIPY/ivp.py:51-131: brentq(xtol =
                        // rtol = 4 EPS) on a +-1 step function == bisection (Q6).  All events are
                        // terminal: the earliest root wins, ties go to the lowest event index.
                        const double xtol = 4 * DBL_EPSILON, brtol = 4 * DBL_EPSILON;
                        for (int k = 0; k < 4; k++) {
                            if (!(active & (1u << k))) continue;
                            double xpre = t, xcur = t_new, xblk = 0;
                            double ez0, ez1, ez2, ec, ecp;
                            D.eval(t, y0, y1, y2, xpre, ez0, ez1, ez2);
                            C.lookup(xpre, ez1, ec, ecp);
                            bool fpre = (C.events(xpre, ez1, ez2, ec) >> k) & 1u;
                            D.eval(t, y0, y1, y2, xcur, ez0, ez1, ez2);
                            C.lookup(xcur, ez1, ec, ecp);
                            bool fcur = (C.events(xcur, ez1, ez2, ec) >> k) & 1u;
                            if (fpre == fcur) { status = PGR_RAY_EVENT_ERROR; break; }
                            for (int it = 0; it < 100; it++) {
                                if (fpre != fcur) xblk = xpre;
                                double delta = (xtol + brtol * fabs(xcur)) / 2;
                                double sbis = (xblk - xcur) / 2;
                                if (fabs(sbis) < delta) break;
                                xpre = xcur;
                                fpre = fcur;
                                if (fabs(sbis) > delta) xcur += sbis;
                                else xcur += (sbis > 0 ? delta : -delta);
                                D.eval(t, y0, y1, y2, xcur, ez0, ez1, ez2);
                                C.lookup(xcur, ez1, ec, ecp);
                                fcur = (C.events(xcur, ez1, ez2, ec) >> k) & 1u;
                            }
                            if (ev < 0 || xcur < best) { best = xcur; ev = k; }
                        }
                    }
#ifdef PGR_DBG_REPLAY
                    dbg_s4 = __builtin_amdgcn_s_memtime();
#endif
                    if (status == RUNNING) {
                        const double t_end = best;
                        // samples of this (truncated) step, REF/launch_rays.py:763-772 (Q5)
                        if (save) {
                            while (jnext < S - 1 && rnext <= t_end) {
                                double o0, o1, o2;
                                D.eval(t, y0, y1, y2, rnext, o0, o1, o2);
                                Tp[(int64_t)jnext * a.stride_smp] = o0;
                                Zp[(int64_t)jnext * a.stride_smp] = SGN(o1);
                                Pp[(int64_t)jnext * a.stride_smp] = SGN(o2);
                                jnext++;
                                rnext = G.at(jnext);
                            }
                        }
                        // terminal event: t = root, y = sol(root) (SCIPY/ivp.py:689-692), then the
                        // bounce logic of REF/launch_rays.py:432-480
                        double r0, r1, r2;
                        D.eval(t, y0, y1, y2, t_end, r0, r1, r2);
                        t = t_end; y0 = r0; y1 = r1; y2 = r2;
#ifdef PGR_DBG_REPLAY
                        asm volatile("" : "+v"(y0), "+v"(y1), "+v"(y2));
                        dbg_b1 = __builtin_amdgcn_s_memtime();
#endif
                        if (ev == 2) status = PGR_RAY_VERTICAL;
                        else if (ev == 3) status = PGR_RAY_BBOX;
                        else {
                            double c, cp;
                            C.lookup(t, y1, c, cp);
                            const double pc_b = y2 * c;
                            const PGR_ASIN_DD_T A_b = PGR_ASIN_DD(pc_b);
                            double theta = PGR_ASIN_DD_HI(A_b) * (180.0 / M_PI);  // ray_angle
#ifdef PGR_DBG_REPLAY
                            asm volatile("" : "+v"(theta));
                            dbg_b2 = __builtin_amdgcn_s_memtime();
#endif
                            double theta_b;
                            if (ev == 0) {
                                theta_b = -theta;
                                ns++;
                            } else {
                                // beta = interp1d(depth_ranges, bottom_angles, 'cubic')(x)
                                const double* xr = es.depth_ranges;
                                if (!(t >= xr[0] && t <= xr[es.nb - 1])) {
                                    status = PGR_RAY_BETA_RANGE;
                                    theta_b = 0;
                                } else {
                                    double beta = 0.0;
                                    if (!es.beta_zero) {
                                        int i;
                                        double xi;
                                        if (es.b_uniform) {
                                            i = cell_uniform(t, es.b0, es.db, es.inv_db, es.nb);
                                            xi = grid_at(es.b0, es.db, i);
                                        } else {
                                            i = cell_search(t, xr, es.nb);
                                            xi = xr[i];
                                        }
                                        double u = t - xi;
                                        const double* q = es.pp + 4 * i;
                                        beta = q[0] + u * (q[1] + u * (q[2] + u * q[3]));
                                    }
                                    theta_b = 2 * beta - theta;
                                    nb++;
                                }
                            }
                            if (status == RUNNING) {
                                if ((as.flags & PGR_TERMINATE_BACKWARDS) && (fabs(theta_b) > 90))
                                    status = PGR_RAY_BACKWARD;
                                else {
                                    // (theta_b = -theta at the surface and on a flat floor: the sine of minus an arcsine, cheaply)
                                    y2 = fdiv(PGR_SIN_REFLECT(theta_b * (M_PI / 180.0), pc_b, A_b), c);
                                    need_init = true;
                                    if (!(t < t_bound)) status = PGR_RAY_OK;
                                    else if (n_steps > max_steps32) status = PGR_RAY_MAX_STEPS;
                                }
                            }
                        }
                    }
                }
#ifdef PGR_DBG_REPLAY
                dbg_s5 = __builtin_amdgcn_s_memtime();
#endif
                if (status == RUNNING && need_init) {
                    // ---- fresh solve_ivp: RK45.__init__ (SCIPY/rk.py:84-104) ----
                    double c;
                    C.rhs(t, y1, y2, f0, f1, f2, c);
                    // select_initial_step, SCIPY/common.py:68-134 (order 4, direction +1, max_step inf)
                    double interval = fabs(t_bound - t);
                    double s0 = atol + fabs(y0) * rtol, s1 = atol + fabs(y1) * rtol,
                           s2 = atol + fabs(y2) * rtol;
                    // (fdiv: correctly rounded like the compiler's division, a third of its instructions)
                    double d0 = rms3(fdiv(y0, s0), fdiv(y1, s1), fdiv(y2, s2));
                    double d1 = rms3(fdiv(f0, s0), fdiv(f1, s1), fdiv(f2, s2));
                    double h0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : fdiv(0.01 * d0, d1);
                    if (!(h0 < interval)) h0 = interval;
                    double e0, e1, e2, cdummy;
                    C.rhs(t + h0 * 1.0, y1 + h0 * 1.0 * f1, y2 + h0 * 1.0 * f2, e0, e1, e2, cdummy);
                    double d2 = fdiv(rms3(fdiv(e0 - f0, s0), fdiv(e1 - f1, s1), fdiv(e2 - f2, s2)), h0);
                    double h1;
                    if (d1 <= 1e-15 && d2 <= 1e-15) {
                        h1 = h0 * 1e-3;
                        if (!(h1 > 1e-6)) h1 = 1e-6;
                    } else {
                        h1 = pgr_cr_pow_p02(fdiv(0.01, (d2 > d1) ? d2 : d1));
                    }
                    h_abs = 100 * h0;
                    if (h1 < h_abs) h_abs = h1;
                    if (interval < h_abs) h_abs = interval;
                    // g = [event(t0, y0) ...], SCIPY/ivp.py:649
                    g = C.events(t, y1, y2, c);
                    rejected = false;
                    need_init = false;
                    if (save) {
                        jnext = G.nearest(t, as.inv_dsave);
                        rnext = G.at(jnext);
                    }
                }
#ifdef PGR_DBG_REPLAY
                {
                    const unsigned long long dbg_s6 = __builtin_amdgcn_s_memtime();
                    const int ln = threadIdx.x & 63;
                    // lanes 6..11: whole service; stage replay + Q; Newton + band edges; (replay: lanes 2-4); samples + root + reflection; init
                    fallbacks += (ln == 6) ? (int)(dbg_s6 - dbg_s0) : (ln == 7) ? (int)(dbg_s1 - dbg_s0) : (ln == 8) ? (int)(dbg_r0 - dbg_s1)
                               : (ln == 9) ? (int)(dbg_s4 - dbg_r3) : (ln == 10) ? (int)(dbg_s5 - dbg_s4) : (ln == 11) ? (int)(dbg_s6 - dbg_s5)
                               : (ln == 12) ? (int)(dbg_n1 - dbg_s1) : (ln == 13) ? (int)(dbg_n2 - dbg_n1) : (ln == 14) ? (int)(dbg_n3 - dbg_n2)
                               : (ln == 15) ? (int)(dbg_b1 - dbg_s4) : (ln == 16) ? (int)(dbg_b2 - dbg_b1) : (ln == 17) ? (int)(dbg_s5 - dbg_b2) : 0;
                }
#endif
            }
        }
    } while (ballot64(status == RUNNING) != 0);

    if (valid) {
        bool ok = (status == PGR_RAY_OK);
        double nan = __longlong_as_double(0x7ff8000000000000LL);
        // the output pointers are read from the kernel-argument segment HERE (through a pointer the
        // compiler cannot trace back to the arguments): as ordinary arguments they are loaded in the
        // prologue and hold 14 SGPRs across the step loop, which spills loop values to VGPR lanes
        typedef const FanArgs __attribute__((address_space(4))) * KArgs;
        const char __attribute__((address_space(4))) * kp =
            (const char __attribute__((address_space(4))) *)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(kp));
        const FanArgs __attribute__((address_space(4))) & a = *(KArgs)(kp + 8);  // (shadows the argument)
        if (save) {
            if (ok) {
                // last column = exact final state (REF/launch_rays.py:775-777)
                Tp[(int64_t)(S - 1) * a.stride_smp] = y0;
                Zp[(int64_t)(S - 1) * a.stride_smp] = SGN(y1);
                Pp[(int64_t)(S - 1) * a.stride_smp] = SGN(y2);
            } else {
                for (int j = 0; j < S; j++) {
                    Tp[(int64_t)j * a.stride_smp] = nan;
                    Zp[(int64_t)j * a.stride_smp] = nan;
                    Pp[(int64_t)j * a.stride_smp] = nan;
                }
            }
        }
        if (a.end_state) {
            if (a.flags & PGR_PACKED_END) {
                // the 40-byte end record of the multi-GPU all-gather, written in place: T, z, p and
                // {n_bott, n_surf}, {status, valid = 1} as int32 pairs in two more double slots
                double* rec = a.end_state + 5 * ray;
                rec[0] = ok ? y0 : nan; rec[1] = ok ? y1 : nan; rec[2] = ok ? y2 : nan;
                int* ir = (int*)(rec + 3);
                ir[0] = nb; ir[1] = ns; ir[2] = status; ir[3] = 1;
            } else {
                a.end_state[3 * ray + 0] = ok ? y0 : nan;
                a.end_state[3 * ray + 1] = ok ? y1 : nan;
                a.end_state[3 * ray + 2] = ok ? y2 : nan;
            }
        }
        a.n_bott[ray] = nb;
        a.n_surf[ray] = ns;
        a.status[ray] = status;
        if (a.n_steps) a.n_steps[ray] = n_steps;
#ifdef PGR_TIMING
        if ((threadIdx.x & 63) < 24) {
            unsigned v = 0;
            for (int k = 0; k < 24; k++) v = ((threadIdx.x & 63) == k) ? tacc[k] : v;
            n_rej = (int)v;
        }
#endif
        if (a.n_rej) a.n_rej[ray] = (a.flags & PGR_DEBUG_TRIPS) ? (((threadIdx.x & 63) == 0) ? trips : (((threadIdx.x & 63) == 1) ? services : fallbacks)) : n_rej;
    }
#undef Tp
#undef Zp
#undef Pp
}

// ------------------------------------------------------------------------------------
// Cost-aware wave placement for fans of 1-2 waves per SIMD.
//
// With the LDS table there is one workgroup per CU, and a 1e5-ray fan is only ~1.5 waves per
// SIMD: the launch lasts exactly as long as its slowest wave (the steepest rays: most steps,
// most bounces), and that wave runs ~20 % slower when another wave shares its SIMD.  A
// workgroup's waves go to the CU's four SIMDs cyclically, so waves k and k+4 of a workgroup
// share a SIMD and, in a W-wave workgroup (4 < W <= 8), waves W-4..3 have a SIMD to themselves.
// The grid is widened to every CU, which leaves spare slots; the most expensive waves (cost
// proxy: the largest |p0| of the wave's rays -- steep rays bounce) get the natural lone slots,
// the next ones get a pair slot whose partner slot stays empty, and the rest are paired
// expensive-with-cheap.  Placement only changes WHERE a wave runs, never what it computes.
// ------------------------------------------------------------------------------------
__global__ void pgr_wave_cost(const double* __restrict__ y0, int64_t N, int n_waves, float* __restrict__ cost)
{
    int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (w >= n_waves) return;
    int64_t ray = (int64_t)w * 64 + (threadIdx.x & 63);
    float c = (ray < N) ? fabsf((float)y0[3 * ray + 2]) : 0.0f;
    for (int o = 32; o > 0; o >>= 1) c = fmaxf(c, __shfl_xor(c, o));
    if ((threadIdx.x & 63) == 0) cost[w] = c;
}

// Ranks the waves by descending cost with a 4096-bin counting sort (order inside a bin is
// irrelevant for scheduling) and writes slot -> wave.  mode 1: strided deal + priorities;
// mode 2, single round (n_waves <= B*W): lone / empty-partner / expensive-with-cheap placement;
// mode 3, several rounds: workgroup b gets the waves of rank b*W .. b*W+W-1, so that every
// workgroup is homogeneous (it holds its CU and LDS until its LAST wave ends) and workgroups
// are dispatched longest first.
__global__ void __launch_bounds__(1024)
pgr_wave_place(const float* __restrict__ cost, int n_waves, int B, int W, int mode,
               int* __restrict__ map)
{
    constexpr int NB = 4096;
    __shared__ int bins[NB];      // count, then exclusive prefix from the expensive end
    __shared__ int cursor[NB];
    __shared__ float smax[1024];
    float mx = 0.0f;
    for (int i = threadIdx.x; i < n_waves; i += blockDim.x) mx = fmaxf(mx, cost[i]);
    smax[threadIdx.x] = mx;
    for (int i = threadIdx.x; i < NB; i += blockDim.x) { bins[i] = 0; cursor[i] = 0; }
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) smax[threadIdx.x] = fmaxf(smax[threadIdx.x], smax[threadIdx.x + o]);
        __syncthreads();
    }
    const float scale = smax[0] > 0.0f ? (float)(NB - 1) / smax[0] : 0.0f;
    for (int i = threadIdx.x; i < n_waves; i += blockDim.x)
        atomicAdd(&bins[min(NB - 1, (int)(cost[i] * scale))], 1);
    __syncthreads();
    {   // exclusive prefix over the bins, most expensive bin first: 4 bins per thread + a block scan
        __shared__ int part[1024];
        const int t = threadIdx.x, hi = NB - 1 - 4 * t;   // this thread's bins: hi, hi-1, hi-2, hi-3
        const int c0 = bins[hi], c1 = bins[hi - 1], c2 = bins[hi - 2], c3 = bins[hi - 3];
        part[t] = c0 + c1 + c2 + c3;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {               // Hillis-Steele inclusive scan
            int v = (t >= o) ? part[t - o] : 0;
            __syncthreads();
            part[t] += v;
            __syncthreads();
        }
        const int ex = part[t] - (c0 + c1 + c2 + c3);
        bins[hi] = ex; bins[hi - 1] = ex + c0; bins[hi - 2] = ex + c0 + c1; bins[hi - 3] = ex + c0 + c1 + c2;
    }
    __syncthreads();
    const int lone_per_block = 8 - W;                 // waves W-4 .. 3
    const int n_lone = B * lone_per_block;
    const int pairs = B * (W - 4);
    int spare = B * W - n_waves;
    int E = spare < pairs ? spare : pairs;            // pair slots run with an empty partner
    const int P = pairs - E;                          // fully used pairs
    for (int w = threadIdx.x; w < n_waves; w += blockDim.x) {
        int bin = min(NB - 1, (int)(cost[w] * scale));
        int r = bins[bin] + atomicAdd(&cursor[bin], 1);   // rank by descending cost
        int idx;
        if (mode == 1) {                              // keep the round-robin deal
            idx = (w % B) * W + w / B;
        } else if (mode == 3) {                       // cost-sorted, homogeneous workgroups
            idx = r;
        } else if (r < n_lone) {
            idx = (r % B) * W + (W - 4) + r / B;
        } else if (r < n_lone + E) {
            int q = r - n_lone;
            idx = (q % B) * W + q / B;                // partner slot + 4 stays empty
        } else {
            int p = r - n_lone - E;                   // 0 .. 2P-1, descending cost
            int first = p < P;
            int pi = first ? p : (2 * P - 1 - p);     // expensive half meets cheap half
            int q = E + pi;
            idx = (q % B) * W + q / B + (first ? 0 : 4);
        }
        int prio = 3 - min(3, (int)((4LL * r) / n_waves));  // cost quartile
        map[idx] = w | (prio << 28);
    }
}

// one RK45 step attempt from given (t, y, h) -- rk_step + the error norm + the controller's power --
// exactly as the fan kernel computes it (same macros), for step-by-step comparison with the oracle's
// trace (tests / scripts/trace_diff.py): out[k] = {y_new[3], f_new[3], error_norm, 0.9 err**-0.2, f[3]}
template <int ZM>
__global__ void pgr_step_kernel(const EnvDev* __restrict__ env_p, const double* __restrict__ tt,
                                const double* __restrict__ yy, const double* __restrict__ hh, int64_t M,
                                double rtol, double atol, double* __restrict__ out)
{
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= M) return;
    const EnvDev& env = *env_p;
    const Ctx<false, ZM> C(env, nullptr);
    const double vA21 = A21, vA31 = A31, vA32 = A32, vA41 = A41, vA42 = A42, vA43 = A43, vA51 = A51, vA52 = A52,
                 vA53 = A53, vA54 = A54, vA61 = A61, vA62 = A62, vA63 = A63, vA64 = A64, vA65 = A65, vB1 = B1,
                 vB3 = B3, vB4 = B4, vB5 = B5, vB6 = B6, vE1 = E1, vE3 = E3, vE4 = E4, vE5 = E5, vE6 = E6, vE7 = E7;
    const double t = tt[k], h = hh[k], y0 = yy[3 * k], y1 = yy[3 * k + 1], y2 = yy[3 * k + 2];
    double f0, f1, f2, c0;
    C.rhs(t, y1, y2, f0, f1, f2, c0);
#ifdef PGR_TIMING
    unsigned tacc[24] = {0}, tprev = 0;  // (the stage macro's stamps)
#endif
    PGR_RK_STAGES(t, h);
    const double sc0 = atol + fmax(fabs(y0), fabs(n0)) * rtol;
    const double sc1 = atol + fmax(fabs(y1), fabs(n1)) * rtol;
    const double sc2 = atol + fmax(fabs(y2), fabs(n2)) * rtol;
    const double error_norm = rms3(fdiv(es0 * h, sc0), fdiv(es1 * h, sc1), fdiv(es2 * h, sc2));
    double* o = out + 11 * k;
    o[0] = n0; o[1] = n1; o[2] = n2; o[3] = k70; o[4] = k71; o[5] = k72;
    o[6] = error_norm; o[7] = 0.9 * pow_m02(error_norm);
    o[8] = f0; o[9] = f1; o[10] = f2;
    (void)c_new; (void)cs;
}

// unit-level evaluation of a1-a8 at arbitrary points (parity tests)
__global__ void pgr_eval_kernel(EnvDev env, const double* x, const double* y, int64_t M, double* out)
{
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= M) return;
    const Ctx<false, 0> C(env, nullptr);
    double d0, d1, d2, c;
    C.rhs(x[k], y[3 * k + 1], y[3 * k + 2], d0, d1, d2, c);
    double* o = out + 10 * k;
    o[0] = d0; o[1] = d1; o[2] = d2; o[3] = c;
    o[4] = pgr_cr_asin(y[3 * k + 2] * c) * (180.0 / M_PI);
    unsigned g = C.events(x[k], y[3 * k + 1], y[3 * k + 2], c);
    for (int q = 0; q < 4; q++) o[5 + q] = ((g >> q) & 1u) ? 1.0 : -1.0;
    o[9] = C.bathy(x[k]);
}

// accuracy probe for the arithmetic building blocks (tests only)
__global__ void pgr_math_kernel(const double* a, const double* b, int64_t M, double* out)
{
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= M) return;
    double* o = out + 9 * k;
    o[0] = fdiv(a[k], b[k]);
    o[1] = frcp(b[k]);
    o[2] = frsqrt(b[k]);
    o[3] = fsqrt(b[k]);
    o[4] = pow_m02(b[k]);
    o[5] = min_step_of(a[k]);
    o[6] = pgr_cr_pow_p02(b[k]);
    o[7] = pgr_cr_asin(a[k]);
    o[8] = pgr_cr_sin(a[k]);
}

// ====================================================================================
// host side
// ====================================================================================
static thread_local std::string g_err;

static int fail(const std::string& m)
{
    g_err = m;
    return -1;
}
#define HIPCHK(call)                                                                       \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess)                                                              \
            return fail(std::string(#call) + ": " + hipGetErrorString(e_));                \
    } while (0)

struct pgr_env {
    int device = 0;
    // tuning options of THIS environment (pgr_env_set_option; per-ray results never depend on them).
    // No process-wide state: two host threads driving two GPUs keep two environments.
    int waves_per_block = 0;          // 0 = automatic
    int depth_search = 0;             // 0: automatic, 1: binary search only, 2: bucket table but no index polynomial (tests)
    int park_lanes = 64, park_trips = 16;
    int place = 2;                    // 0 off, 1 issue priorities only, 2 cost-aware placement + priorities
    hipStream_t stream = nullptr;     // the host-pointer entry's own stream (created on first use)
    EnvDev d{};
    const EnvDev* d_dev = nullptr;  // device copy of `d` (kernel argument by pointer)
    // grow-only staging workspace of the host-pointer entry (kept while <= 256 MB so the many
    // small fans of an eigenray search do not pay 11 hipMalloc/hipFree per call)
    void* ws = nullptr;
    size_t ws_bytes = 0;
    std::mutex ws_mutex;
    // ring of small buffers for the per-launch wave placement (cost[2048] + map[2048])
    static constexpr int kPlaceRing = 8;
    void* place_buf = nullptr;     // owned through `allocs`
    size_t place_slot_bytes = 0;
    int place_next = 0;
    std::mutex place_mutex;
    int range_indep = 0;
    int lds_path = 0;
    std::vector<void*> allocs;
    int num_cus = 256;
    size_t max_lds = 64 * 1024;
};

extern "C" const char* pgr_last_error(void) { return g_err.c_str(); }

// What the build did to this library: the second pass of the build (pygenray_amd/_isa_layout.py, run
// by pygenray_amd/_lib.py) re-encodes the device code and, when it has succeeded, overwrites this tag
// in the host object -- "plain hipcc" means the pass did not run or failed and the unmodified hipcc
// output is what is loaded.  The arithmetic switches come from the preprocessor.
extern "C" {
__attribute__((used)) char pgr_build_tag[96] = "PGR_BUILD_TAG:plain hipcc                                                                     ";
}
extern "C" const char* pgr_build_info(void)
{
    static std::string info;
    static std::once_flag once;
    std::call_once(once, [] {
        std::string t(pgr_build_tag + 14);
        while (!t.empty() && t.back() == ' ') t.pop_back();
        info = "layout: " + t + "; arithmetic: ";
#if defined(PGR_FMA)
        info += "FMA contraction (experiments only)";
#elif defined(PGR_STRICT)
        info += "compiler IEEE divide/sqrt";
#else
        info += "reference order, correctly rounded div/sqrt/pow/asin/sin";
#endif
#ifdef PGR_EXACT_RSQRT
        info += ", exact 1/sqrt";
#endif
#ifdef PGR_POW_2ULP
        info += ", 2-ulp pow (NOT bit-identical)";
#endif
#ifdef PGR_NO_REPLAY
        info += ", no brentq replay (NOT bit-identical)";
#endif
#ifdef PGR_LIBM_TRIG
        info += ", device-library asin/sin (NOT bit-identical)";
#endif
    });
    return info.c_str();
}

extern "C" int pgr_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { fail("hipGetDeviceCount failed"); return -1; }
    return n;
}

extern "C" int pgr_env_set_option(pgr_env* env, int what, int a, int b)
{
    if (!env) return fail("pgr_env_set_option: null env");
    switch (what) {
    case PGR_OPT_WAVES_PER_BLOCK:
        if (a < 0 || a > 8) return fail("waves per block must be in [0,8]");
        env->waves_per_block = a;
        return 0;
    case PGR_OPT_DEPTH_SEARCH:
        if (a < 0 || a > 2) return fail("depth search: 0 = automatic, 1 = binary search, 2 = bucket table (no index polynomial)");
        env->depth_search = a;
        return 0;
    case PGR_OPT_PARK:
        if (a < 1 || a > 64 || b < 0 || b > 100000) return fail("park: lanes in [1,64], trips >= 0");
        env->park_lanes = a;
        env->park_trips = b;
        return 0;
    case PGR_OPT_PLACEMENT:
        if (a < 0 || a > 2) return fail("placement: 0 = off, 1 = priorities only, 2 = placement + priorities");
        env->place = a;
        return 0;
    default:
        return fail("pgr_env_set_option: unknown option");
    }
}

// grid[j] == g0 + j*dg for all j, evaluated exactly as the device does (mul, then add)
static bool exactly_uniform(const double* g, int64_t n, double& g0, double& dg)
{
    if (n < 2) return false;
    g0 = g[0];
    dg = g[1] - g[0];
    if (!(dg > 0) || !std::isfinite(dg)) return false;
    for (int64_t j = 0; j < n; j++) {
        volatile double m = (double)j * dg;
        volatile double v = g0 + m;
        if (v != g[j]) return false;
    }
    return true;
}

// Not-a-knot cubic through (x, y): scipy.interpolate.interp1d(kind="cubic") ==
// make_interp_spline(k=3, bc_type=None) (REF/launch_rays.py:397-399).  Built in
// piecewise-polynomial form with the standard not-a-knot end rows; pp[4i..] = {y_i, s_i, c2, c3}.
static bool build_notaknot(const double* x, const double* y, int64_t n, std::vector<double>& pp)
{
    if (n < 4) return false;
    std::vector<double> dx(n), sl(n), lo(n), di(n), up(n), b(n);
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
    for (int64_t i = 1; i < n; i++) {
        double m = lo[i] / di[i - 1];
        di[i] -= m * up[i - 1];
        b[i] -= m * b[i - 1];
    }
    b[n - 1] /= di[n - 1];
    for (int64_t i = n - 2; i >= 0; i--) b[i] = (b[i] - up[i] * b[i + 1]) / di[i];
    pp.assign(4 * (size_t)(n - 1), 0.0);
    for (int64_t i = 0; i < n - 1; i++) {
        pp[4 * i + 0] = y[i];
        pp[4 * i + 1] = b[i];
        pp[4 * i + 2] = (3 * sl[i] - 2 * b[i] - b[i + 1]) / dx[i];
        pp[4 * i + 3] = (b[i] + b[i + 1] - 2 * sl[i]) / (dx[i] * dx[i]);
    }
    return true;
}

template <class T>
static int upload(pgr_env* e, const T* host, size_t count, const T** dev)
{
    void* p = nullptr;
    HIPCHK(hipMalloc(&p, count * sizeof(T)));
    e->allocs.push_back(p);
    HIPCHK(hipMemcpy(p, host, count * sizeof(T), hipMemcpyHostToDevice));
    *dev = (const T*)p;
    return 0;
}

extern "C" void pgr_env_destroy(pgr_env* env)
{
    if (!env) return;
    (void)hipSetDevice(env->device);
    for (void* p : env->allocs) (void)hipFree(p);
    if (env->ws) (void)hipFree(env->ws);
    if (env->stream) (void)hipStreamDestroy(env->stream);
    delete env;
}

extern "C" int pgr_env_create(pgr_env** out, int device, const double* cin, const double* cpin,
                              const double* rin, const double* zin, int64_t nr, int64_t nz,
                              const double* depths, const double* depth_ranges,
                              const double* bottom_angles, int64_t nb)
{
    if (!out || !cin || !cpin || !rin || !zin || !depths || !depth_ranges || !bottom_angles)
        return fail("pgr_env_create: null argument");
    if (nr < 2 || nz < 2) return fail("sound speed table needs at least 2 range and 2 depth points");
    if (nr > (1 << 30) || nz > (1 << 30) || nb > (1 << 30)) return fail("table too large");
    if (nb < 4) return fail("x and y arrays must have at least 4 entries");  // interp1d(kind='cubic')
    // REF/launch_rays.py:79-90
    for (int64_t i = 1; i < nr; i++)
        if (!(rin[i] - rin[i - 1] >= 0))
            return fail("Sound speed range coordinates must be monotonically increasing.");
    for (int64_t i = 1; i < nz; i++)
        if (!(zin[i] - zin[i - 1] >= 0))
            return fail("Sound speed depth coordinates must be monotonically increasing.");
    for (int64_t i = 1; i < nb; i++)
        if (!(depth_ranges[i] - depth_ranges[i - 1] >= 0))
            return fail("Bathymetry range coordinates must be monotonically increasing.");
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail("pgr_env_create: no such HIP device");
    HIPCHK(hipSetDevice(device));

    pgr_env* e = new pgr_env();
    e->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
        e->num_cus = prop.multiProcessorCount;
        e->max_lds = prop.sharedMemPerBlock;
        if (strncmp(prop.gcnArchName, "gfx950", 6) == 0) e->max_lds = 160 * 1024;  // CDNA4 LDS per CU
    }
    int optin = 0;
    if (hipDeviceGetAttribute(&optin, hipDeviceAttributeMaxSharedMemoryPerBlock, device) == hipSuccess &&
        (size_t)optin > e->max_lds)
        e->max_lds = (size_t)optin;

    // range independence: every row bitwise equal to row 0 (both tables)
    bool indep = true;
    for (int64_t i = 1; i < nr && indep; i++)
        indep = memcmp(cin + i * nz, cin, sizeof(double) * nz) == 0 &&
                memcmp(cpin + i * nz, cpin, sizeof(double) * nz) == 0;
    e->range_indep = indep;
    size_t rows = indep ? 1 : (size_t)nr;
    std::vector<double2> tab(rows * (size_t)nz);
    for (size_t i = 0; i < rows; i++)
        for (int64_t j = 0; j < nz; j++) tab[i * nz + j] = make_double2(cin[i * nz + j], cpin[i * nz + j]);
    e->lds_path = indep && ((size_t)nz * sizeof(double2) <= e->max_lds);
    std::vector<double> pp;
    if (!build_notaknot(depth_ranges, bottom_angles, nb, pp)) {
        delete e;
        return fail("x and y arrays must have at least 4 entries");
    }
    EnvDev& d = e->d;
    int rc = 0;
    rc |= upload(e, tab.data(), tab.size(), &d.tab);
    rc |= upload(e, rin, (size_t)nr, &d.rin);
    rc |= upload(e, zin, (size_t)nz, &d.zin);
    rc |= upload(e, depths, (size_t)nb, &d.depths);
    rc |= upload(e, depth_ranges, (size_t)nb, &d.depth_ranges);
    rc |= upload(e, pp.data(), pp.size(), &d.pp);
    if (rc) { pgr_env_destroy(e); return -1; }
    d.nr = (int)nr; d.nz = (int)nz; d.nb = (int)nb;
    d.row_stride = indep ? 0 : (int)nz;
    d.z_uniform = exactly_uniform(zin, nz, d.z0, d.dz);
    d.inv_dz = d.z_uniform ? 1.0 / d.dz : 0.0;
    d.z_pow2 = 0;
    if (d.z_uniform) {
        int ex = 0;
        bool pow2 = (std::frexp(d.dz, &ex) == 0.5);
        for (int64_t j = 0; j + 1 < nz && pow2; j++) pow2 = (zin[j + 1] - zin[j] == d.dz);
        d.z_pow2 = pow2 ? 1 : 0;
    }
    d.z_simple = (d.z_pow2 && d.z0 == 0.0) ? 1 : 0;
    d.b_zmin = depths[0];
    for (int64_t i = 1; i < nb; i++) d.b_zmin = depths[i] < d.b_zmin ? depths[i] : d.b_zmin;
    d.b_zmin -= 1.0;
    d.b_xlo = depth_ranges[0];
    d.b_xhi = depth_ranges[nb - 1];
    d.r_uniform = exactly_uniform(rin, nr, d.r0, d.dr);
    d.inv_dr = d.r_uniform ? 1.0 / d.dr : 0.0;
    d.b_uniform = exactly_uniform(depth_ranges, nb, d.b0, d.db);
    d.beta_zero = 1;
    for (double v : pp) if (v != 0.0) d.beta_zero = 0;
    d.inv_db = d.b_uniform ? 1.0 / d.db : 0.0;
    d.c_lo = cin[0]; d.c_hi = cin[0];
    for (int64_t k = 0; k < nr * nz; k++) {
        d.c_lo = cin[k] < d.c_lo ? cin[k] : d.c_lo;
        d.c_hi = cin[k] > d.c_hi ? cin[k] : d.c_hi;
    }
    if (!(d.c_lo > 0) || !std::isfinite(d.c_hi)) { d.c_lo = 0.0; d.c_hi = INFINITY; }  // no shortcut for such a table
    d.c_hi *= 1.001;
    const double tol = 1e-6;
    d.zhi_tol = zin[nz - 1] + tol;
    d.zlo_tol = zin[0] - tol;
    d.rlo_tol = rin[0] - tol;
    d.rhi_tol = rin[nr - 1] + tol;
    // bucketed depth search for a non-uniform zin (see EnvDev): bins of 0.9 min(diff(zin))
    d.z_bucket = 0; d.zbucket = nullptr; d.zb_B = 0; d.zb_z0 = 0.0; d.zb_inv_w = 0.0;
    if (!d.z_uniform && nz >= 3 && nz <= 65535) {
        double min_dz = zin[1] - zin[0];
        for (int64_t j = 1; j + 1 < nz; j++) min_dz = (zin[j + 1] - zin[j] < min_dz) ? zin[j + 1] - zin[j] : min_dz;
        const double span = zin[nz - 1] - zin[0];
        const double w = 0.9 * min_dz;
        if (min_dz > 0 && span > 0 && std::floor(span / w) + 2 <= 32768.0) {
            const int B = (int)(std::floor(span / w) + 2);
            std::vector<unsigned short> bk((size_t)B);
            bool ok = true;
            int64_t j = 0;
            for (int k = 0; k < B && ok; k++) {
                // every z the device maps to bin k (floor((z - z0) * (1/w)), two roundings) lies in [L, U)
                const double L = zin[0] + w * ((double)k - (double)(k + 1) * 1e-12);
                const double U = zin[0] + w * ((double)(k + 1) + (double)(k + 1) * 1e-12);
                while (j + 1 <= nz - 2 && zin[j + 1] < L) j++;  // j = max{ j : zin[j] < L } in [0, nz-2]
                bk[(size_t)k] = (unsigned short)j;
                if (j + 2 <= nz - 1 && !(U <= zin[j + 2])) ok = false;  // the cell is j or j+1, never beyond
            }
            if (ok && upload(e, bk.data(), bk.size(), &d.zbucket) == 0) {
                d.z_bucket = 1; d.zb_B = B; d.zb_z0 = zin[0]; d.zb_inv_w = 1.0 / w;
            }
        }
    }
    // quadratic index estimate of a smooth non-uniform zin (least squares on (u_j, j), u in [0, 1])
    d.z_quad = 0; d.zq_c0 = d.zq_c1 = d.zq_c2 = d.zq_inv_span = 0.0;
    if (!d.z_uniform && nz >= 4 && zin[nz - 1] > zin[0]) {
        const double span = zin[nz - 1] - zin[0], inv_span = 1.0 / span;
        long double S0 = 0, S1 = 0, S2 = 0, S3 = 0, S4 = 0, T0 = 0, T1 = 0, T2 = 0;
        for (int64_t j = 0; j < nz; j++) {
            const long double u = (long double)((zin[j] - zin[0]) * inv_span), y = (long double)j;
            S0 += 1; S1 += u; S2 += u * u; S3 += u * u * u; S4 += u * u * u * u;
            T0 += y; T1 += y * u; T2 += y * u * u;
        }
        // normal equations [S0 S1 S2; S1 S2 S3; S2 S3 S4] c = [T0 T1 T2] by Cramer's rule
        const long double D = S0 * (S2 * S4 - S3 * S3) - S1 * (S1 * S4 - S3 * S2) + S2 * (S1 * S3 - S2 * S2);
        if (D != 0) {
            const double c0 = (double)((T0 * (S2 * S4 - S3 * S3) - S1 * (T1 * S4 - S3 * T2) + S2 * (T1 * S3 - S2 * T2)) / D);
            const double c1 = (double)((S0 * (T1 * S4 - T2 * S3) - T0 * (S1 * S4 - S3 * S2) + S2 * (S1 * T2 - S2 * T1)) / D);
            const double c2 = (double)((S0 * (S2 * T2 - S3 * T1) - S1 * (S1 * T2 - S2 * T1) + T0 * (S1 * S3 - S2 * S2)) / D);
            bool ok = (c1 > 0) && (c1 + 2 * c2 > 0);   // g' > 0 on [0, 1]
            for (int64_t j = 0; j < nz && ok; j++) {
                volatile double u = (zin[j] - zin[0]) * inv_span;   // the device's own arithmetic
                volatile double q = c1 + u * c2;
                volatile double g = c0 + u * q;
                ok = std::fabs((double)g - (double)j) <= 0.45;
            }
            if (ok) { d.z_quad = 1; d.zq_c0 = c0; d.zq_c1 = c1; d.zq_c2 = c2; d.zq_inv_span = inv_span; d.zb_z0 = zin[0]; }
        }
    }
    if (upload(e, &e->d, 1, &e->d_dev)) { pgr_env_destroy(e); return -1; }
    *out = e;
    return 0;
}

extern "C" int pgr_env_query(const pgr_env* env, int what)
{
    if (!env) return fail("null env");
    switch (what) {
    case 0: return env->range_indep;
    case 1: return env->d.z_uniform;
    case 2: return env->d.r_uniform;
    case 3: return env->lds_path;
    case 4: return env->device;
    default: return fail("pgr_env_query: unknown property");
    }
}

// Builds the slot -> wave map for this launch on `st` (see pgr_wave_place); returns the map and
// the grid size through the references, or leaves map null when scheduling is off / not useful.
static int schedule_waves(pgr_env* env, const double* y0, int64_t N, int64_t waves, int W, hipStream_t st,
                          const int*& map_out, int64_t& blocks)
{
    map_out = nullptr;
    if (env->place == 0 || env->waves_per_block != 0 || W < 5 || waves > (1 << 27)) return 0;
    const int64_t cus = env->num_cus;
    int mode, B;
    if (waves <= 8 * cus && W <= 8 && waves > 4 * cus) {  // single round, 1-2 waves per SIMD
        mode = env->place;                                  // 1 or 2
        B = (mode == 1) ? (int)((waves + W - 1) / W) : (int)cus;
    } else if (waves > 8 * cus) {                           // several rounds
        mode = 3;
        B = (int)((waves + W - 1) / W);
    } else {
        return 0;
    }
    std::lock_guard<std::mutex> lock(env->place_mutex);  // host threads may share an env
    size_t n_slots = (size_t)B * W;
    size_t need = ((size_t)waves * 4 + n_slots * 4 + 511) & ~(size_t)255;
    if (need > env->place_slot_bytes) {  // grow-only ring (previous launches may still read theirs:
        // the old buffer is kept until the env dies)
        size_t sz = need > 65536 ? need : 65536;
        void* nb = nullptr;
        HIPCHK(hipMalloc(&nb, sz * pgr_env::kPlaceRing));
        env->allocs.push_back(nb);
        env->place_buf = nb;
        env->place_slot_bytes = sz;
    }
    char* slot = (char*)env->place_buf + (size_t)(env->place_next++ % pgr_env::kPlaceRing) * env->place_slot_bytes;
    float* cost = (float*)slot;
    int* map = (int*)(slot + (((size_t)waves * 4 + 255) & ~(size_t)255));
    HIPCHK(hipMemsetAsync(map, 0xFF, n_slots * sizeof(int), st));
    hipLaunchKernelGGL(pgr_wave_cost, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, y0, N, (int)waves, cost);
    hipLaunchKernelGGL(pgr_wave_place, dim3(1), dim3(1024), 0, st, cost, (int)waves, B, W, mode, map);
    map_out = map;
    blocks = B;
    return 0;
}

extern "C" int pgr_shoot_fan_device(pgr_env* env, const double* y0, int64_t N, double source_range,
                                    double receiver_range, const double* r_save, int32_t S,
                                    double rtol, double atol, uint32_t flags, int64_t max_steps,
                                    double* T, double* z, double* p, double* end_state,
                                    int32_t* n_bott, int32_t* n_surf, int32_t* status,
                                    int32_t* n_steps, int32_t* n_rej, void* stream)
{
    if (!env) return fail("pgr_shoot_fan: null env");
    if (N < 0) return fail("pgr_shoot_fan: negative ray count");
    if (N == 0) return 0;
    if (!y0 || !n_bott || !n_surf || !status) return fail("pgr_shoot_fan: null argument");
    bool save = (T != nullptr);
    if (save && (!z || !p || !r_save)) return fail("pgr_shoot_fan: T, z, p and r_save go together");
    if (save && S < 1) return fail("pgr_shoot_fan: num_range_save must be >= 1");
    if (!(rtol > 0) || !(atol >= 0)) return fail("pgr_shoot_fan: bad tolerances");
    if (max_steps <= 0 || max_steps > (1LL << 30)) return fail("pgr_shoot_fan: max_steps out of range");
    // REF/launch_rays.py:404: an empty `while x < receiver_range` leaves `sols` empty and the
    // reference fails with IndexError; backwards shots are mirrored by the caller first
    if (!(source_range < receiver_range)) return fail("pgr_shoot_fan: need source_range < receiver_range (mirror backwards shots)");
    HIPCHK(hipSetDevice(env->device));

    FanArgs a{};
    a.y0 = y0; a.r_save = r_save; a.T = T; a.Z = z; a.P = p; a.end_state = end_state;
    a.n_bott = n_bott; a.n_surf = n_surf; a.status = status; a.n_steps = n_steps; a.n_rej = n_rej;
    a.N = N; a.S = save ? S : 1;
    if (flags & PGR_SAMPLE_MAJOR) { a.stride_ray = 1; a.stride_smp = N; }
    else { a.stride_ray = S; a.stride_smp = 1; }
    a.x0 = source_range; a.x1 = receiver_range; a.rtol = rtol; a.atol = atol;
    a.inv_dsave = (S > 1 && receiver_range != source_range) ? (double)(S - 1) / (receiver_range - source_range) : 0.0;
    // np.linspace: step = (stop - start) / (num - 1); y = arange(num) * step + start; y[-1] = stop
    a.save_step = (S > 1) ? (receiver_range - source_range) / (double)(S - 1) : 0.0;
    a.save_formula = (flags & PGR_SAVE_LINSPACE) ? 1 : 0;
    a.park_lanes = env->park_lanes;
    a.park_trips = env->park_trips;
    a.max_steps = max_steps; a.flags = flags;

    int64_t waves = (N + 63) / 64;
    hipStream_t st = (hipStream_t)stream;
    // kernel variant: where the table lives (LDS copy of the single profile / HBM) and how a depth
    // cell is found (1: zin[j] = j dz exactly, 2: bucket table + zin in LDS, 0: closed form for other
    // uniform grids or binary search)
    const EnvDev& D = env->d;
    const size_t tab_bytes = (size_t)D.nz * sizeof(double2);
    const size_t zb_bytes = D.z_bucket ? ((size_t)D.nz * sizeof(double) + (((size_t)D.zb_B * 2 + 15) & ~(size_t)15)) : 0;
    bool lds_tab = env->lds_path != 0;
    int zm = D.z_simple ? ((D.dz == 1.0) ? 4 : 1) : 0;
    const size_t zq_bytes = (size_t)D.nz * sizeof(double);
    size_t zx_bytes = 0;  // LDS taken by the depth search of the chosen variant
    if (!D.z_simple && env->depth_search != 1) {
        if (D.z_quad && env->depth_search == 0) {
            if (env->range_indep && tab_bytes + zq_bytes <= env->max_lds) { lds_tab = true; zm = 3; zx_bytes = zq_bytes; }
            else if (zq_bytes <= env->max_lds) { lds_tab = false; zm = 3; zx_bytes = zq_bytes; }
        }
        if (zm == 0 && D.z_bucket) {
            if (env->range_indep && tab_bytes + zb_bytes <= env->max_lds) { lds_tab = true; zm = 2; zx_bytes = zb_bytes; }
            else if (zb_bytes <= env->max_lds) { lds_tab = false; zm = 2; zx_bytes = zb_bytes; }
        }
    }
    int wpb, threads;
    int64_t blocks;
    size_t lds;
    if (lds_tab) {
        // one workgroup per CU (the LDS table is per workgroup): the smallest workgroup that
        // covers the fan in a single round, capped at 8 waves
        wpb = env->waves_per_block;
        if (wpb == 0) {
            wpb = (int)((waves + env->num_cus - 1) / env->num_cus);
            if (wpb < 1) wpb = 1;
            if (wpb > 8) wpb = 8;
        }
        threads = wpb * 64;
        blocks = (N + threads - 1) / threads;
        // cost-aware scheduling of the waves (placement, priorities, homogeneous workgroups)
        if (schedule_waves(env, y0, N, waves, wpb, st, a.wave_map, blocks)) return -1;
        lds = tab_bytes + zx_bytes;
    } else {
        wpb = env->waves_per_block ? env->waves_per_block : 4;
        blocks = (waves + wpb - 1) / wpb;
        // the same scheduling; a fan too small for it keeps 4-wave workgroups
        if (waves > 4 * (int64_t)env->num_cus) {
            int W = waves <= 8 * (int64_t)env->num_cus ? (int)((waves + env->num_cus - 1) / env->num_cus) : 8;
            const int* m = nullptr;
            int64_t nb2 = blocks;
            if (schedule_waves(env, y0, N, waves, W, st, m, nb2)) return -1;
            if (m) { a.wave_map = m; blocks = nb2; wpb = W; }
        }
        threads = wpb * 64;
        lds = zx_bytes;
    }
    // {depth_ranges, depths} behind everything else in the LDS when 16 nb bytes are left
    a.bathy_lds_off = -1;
    {
        const size_t at = (lds + 15) & ~(size_t)15, need = (size_t)D.nb * 16;
        if (at + need <= env->max_lds) { a.bathy_lds_off = (int)at; lds = at + need; }
    }
#define PGR_LAUNCH1(LT, ZMV, SV)                                                                     \
    do {                                                                                             \
        if (lds > 64 * 1024)                                                                         \
            HIPCHK(hipFuncSetAttribute((const void*)pgr_fan_kernel<LT, ZMV, SV>,                     \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));       \
        hipLaunchKernelGGL((pgr_fan_kernel<LT, ZMV, SV>), dim3((unsigned)blocks), dim3(threads), lds, \
                           st, env->d_dev, a);                                                       \
    } while (0)
#define PGR_LAUNCH(LT, ZMV)                                                                          \
    do {                                                                                             \
        if (!save) PGR_LAUNCH1(LT, ZMV, 0);                                                          \
        else if (a.save_formula && !(flags & PGR_EXACT_SAMPLES)) PGR_LAUNCH1(LT, ZMV, 1);            \
        else PGR_LAUNCH1(LT, ZMV, 2);                                                                \
    } while (0)
    if (lds_tab) {
        if (zm == 1) PGR_LAUNCH(true, 1); else if (zm == 2) PGR_LAUNCH(true, 2);
        else if (zm == 3) PGR_LAUNCH(true, 3); else if (zm == 4) PGR_LAUNCH(true, 4); else PGR_LAUNCH(true, 0);
    } else {
        if (zm == 1) PGR_LAUNCH(false, 1); else if (zm == 2) PGR_LAUNCH(false, 2);
        else if (zm == 3) PGR_LAUNCH(false, 3); else if (zm == 4) PGR_LAUNCH(false, 4); else PGR_LAUNCH(false, 0);
    }
#undef PGR_LAUNCH
#undef PGR_LAUNCH1
    HIPCHK(hipGetLastError());
    return 0;
}

namespace {
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 8) == hipSuccess ? 0 : -1; }
};
}  // namespace

// PGR_COMPACT: squeeze the columns of dropped rays out of a sample-major [S][N] array:
// dst[s][m] = src[s][idx[m]], m < M (one pass at HBM speed; idx is increasing, so reads coalesce)
__global__ void pgr_gather_cols(const double* __restrict__ src, double* __restrict__ dst,
                                const int* __restrict__ idx, int64_t M, int64_t N)
{
    int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    int64_t s = blockIdx.y;
    dst[s * M + m] = src[s * N + idx[m]];
}

// Fault in every page of the caller's output buffers from up to 16 threads: each page's first byte is
// read and written back UNCHANGED (a write access, so the page is really allocated, but nothing the
// caller may still want -- a reused buffer whose tail the copies below do not overwrite -- is altered).
static void prefault_outputs(std::initializer_list<double*> bufs, size_t bytes)
{
    const size_t page = 4096;
    unsigned nt = std::thread::hardware_concurrency();
    nt = nt < 1 ? 1 : (nt > 16 ? 16 : nt);
    std::vector<std::thread> th;
    for (double* b : bufs) {
        char* base = (char*)b;
        size_t npages = (bytes + page - 1) / page;
        size_t per = (npages + nt - 1) / nt;
        for (unsigned k = 0; k < nt; k++) {
            size_t p0 = (size_t)k * per, p1 = p0 + per < npages ? p0 + per : npages;
            if (p0 >= p1) break;
            th.emplace_back([base, bytes, p0, p1, page]() {
                for (size_t q = p0; q < p1; q++) {
                    size_t o = q * page;
                    if (o < bytes) {
                        volatile char* c = (volatile char*)base + o;
                        *c = *c;
                    }
                }
            });
        }
    }
    for (auto& t : th) t.join();
}

extern "C" int pgr_shoot_fan(pgr_env* env, const double* y0, int64_t N, double source_range,
                             double receiver_range, const double* r_save, int32_t S, double rtol,
                             double atol, uint32_t flags, int64_t max_steps, double* T, double* z,
                             double* p, double* end_state, int32_t* n_bott, int32_t* n_surf,
                             int32_t* status, int32_t* n_steps, int32_t* n_rej)
{
    if (!env) return fail("pgr_shoot_fan: null env");
    if (N < 0) return fail("pgr_shoot_fan: negative ray count");
    if (N == 0) return 0;
    if (!y0 || !n_bott || !n_surf || !status) return fail("pgr_shoot_fan: null argument");
    bool save = (T != nullptr);
    if (save && (!z || !p || !r_save || S < 1)) return fail("pgr_shoot_fan: T, z, p, r_save, S go together");
    HIPCHK(hipSetDevice(env->device));
    std::lock_guard<std::mutex> lock(env->ws_mutex);
    size_t ns_bytes = (size_t)N * (size_t)(save ? S : 0) * sizeof(double);
    // carve one workspace: y0, r_save, T, Z, P, end, 5 int arrays (256-byte aligned pieces)
    const size_t sizes[11] = {(size_t)N * 24, (size_t)(save ? S : 1) * 8, ns_bytes, ns_bytes, ns_bytes,
                              (size_t)N * 24, (size_t)N * 4, (size_t)N * 4, (size_t)N * 4, (size_t)N * 4,
                              (size_t)N * 4};
    size_t off[11], total = 0;
    for (int k = 0; k < 11; k++) { off[k] = total; total += (sizes[k] + 255) & ~(size_t)255; }
    if (total > env->ws_bytes) {
        if (env->ws) (void)hipFree(env->ws);
        env->ws = nullptr; env->ws_bytes = 0;
        if (hipMalloc(&env->ws, total) != hipSuccess) { env->ws = nullptr; return fail("pgr_shoot_fan: device allocation failed"); }
        env->ws_bytes = total;
    }
    struct Piece { void* p; } dy0{(char*)env->ws + off[0]}, dr{(char*)env->ws + off[1]}, dT{(char*)env->ws + off[2]},
        dZ{(char*)env->ws + off[3]}, dP{(char*)env->ws + off[4]}, dE{(char*)env->ws + off[5]},
        dnb{(char*)env->ws + off[6]}, dns{(char*)env->ws + off[7]}, dst{(char*)env->ws + off[8]},
        dn1{(char*)env->ws + off[9]}, dn2{(char*)env->ws + off[10]};
    struct Trim {  // give a very large workspace (> 16 GB of the 288 GB) back when the call ends
        pgr_env* e;
        ~Trim() { if (e->ws_bytes > ((size_t)16 << 30)) { (void)hipFree(e->ws); e->ws = nullptr; e->ws_bytes = 0; } }
    } trim{env};
    // everything of this call goes through the environment's own stream and waits for THAT stream only
    // (not the device: other streams of the process -- another environment's fan, a framework's copies --
    // are none of its business)
    if (!env->stream) HIPCHK(hipStreamCreateWithFlags(&env->stream, hipStreamNonBlocking));
    hipStream_t st = env->stream;
    HIPCHK(hipMemcpyAsync(dy0.p, y0, N * 3 * sizeof(double), hipMemcpyHostToDevice, st));
    if (save) HIPCHK(hipMemcpyAsync(dr.p, r_save, (size_t)S * sizeof(double), hipMemcpyHostToDevice, st));
    if (save) {
        // is r_save exactly np.linspace(source_range, receiver_range, S)?  then the kernel
        // recomputes it per index instead of loading it
        double step = (S > 1) ? (receiver_range - source_range) / (double)(S - 1) : 0.0;
        bool lin = true;
        for (int32_t j = 0; j < S && lin; j++) {
            volatile double m = (double)j * step;
            volatile double v = m + source_range;
            double want = (j == S - 1 && S > 1) ? receiver_range : (double)v;
            lin = (r_save[j] == want);
        }
        if (lin) flags |= PGR_SAVE_LINSPACE; else flags &= ~PGR_SAVE_LINSPACE;
    }
    int rc = pgr_shoot_fan_device(env, (const double*)dy0.p, N, source_range, receiver_range,
                                  (const double*)dr.p, S, rtol, atol, flags, max_steps,
                                  save ? (double*)dT.p : nullptr, save ? (double*)dZ.p : nullptr,
                                  save ? (double*)dP.p : nullptr, (double*)dE.p, (int32_t*)dnb.p,
                                  (int32_t*)dns.p, (int32_t*)dst.p, (int32_t*)dn1.p, (int32_t*)dn2.p,
                                  (void*)st);
    if (rc) return rc;
    // While the kernel runs: fault in the caller's (typically fresh, untouched) output buffers on
    // several threads.  A D2H copy into untouched pageable memory runs at the page-fault rate of
    // one thread (15 GB/s measured), into touched memory at 56 GB/s (scripts/probes/pcie_probe.py).
    if (save && ns_bytes >= ((size_t)32 << 20)) prefault_outputs({T, z, p}, ns_bytes);
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipMemcpyAsync(status, dst.p, N * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    bool squeezed = false;
    if (save && (flags & PGR_COMPACT)) {
        if (!(flags & PGR_SAMPLE_MAJOR)) return fail("pgr_shoot_fan: PGR_COMPACT needs PGR_SAMPLE_MAJOR");
        if (N > 0x7fffffff) return fail("pgr_shoot_fan: PGR_COMPACT supports at most 2^31 rays per call");
        std::vector<int> keep;
        keep.reserve((size_t)N);
        for (int64_t k = 0; k < N; k++) if (status[k] == 0) keep.push_back((int)k);
        const int64_t M = (int64_t)keep.size();
        if (M < N) {
            // dropped rays leave the trajectories on the device: [S][N] -> [S][M], then one linear copy
            squeezed = true;
            struct Buf { void* p = nullptr; ~Buf() { if (p) (void)hipFree(p); } } tmp, didx;
            if (M > 0) {
                HIPCHK(hipMalloc(&tmp.p, (size_t)S * (size_t)M * sizeof(double)));
                HIPCHK(hipMalloc(&didx.p, (size_t)M * sizeof(int)));
                HIPCHK(hipMemcpyAsync(didx.p, keep.data(), (size_t)M * sizeof(int), hipMemcpyHostToDevice, st));
                double* host[3] = {T, z, p};
                void* dev[3] = {dT.p, dZ.p, dP.p};
                for (int a3 = 0; a3 < 3; a3++) {
                    hipLaunchKernelGGL(pgr_gather_cols, dim3((unsigned)((M + 255) / 256), (unsigned)S), dim3(256), 0,
                                       st, (const double*)dev[a3], (double*)tmp.p, (const int*)didx.p, M, N);
                    HIPCHK(hipGetLastError());
                    HIPCHK(hipMemcpyAsync(host[a3], tmp.p, (size_t)S * (size_t)M * sizeof(double), hipMemcpyDeviceToHost, st));
                    HIPCHK(hipStreamSynchronize(st));  // tmp is reused by the next array
                }
            }
        }
    }
    if (save && !squeezed) {
        HIPCHK(hipMemcpyAsync(T, dT.p, ns_bytes, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(z, dZ.p, ns_bytes, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(p, dP.p, ns_bytes, hipMemcpyDeviceToHost, st));
    }
    if (end_state) HIPCHK(hipMemcpyAsync(end_state, dE.p, N * 3 * sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(n_bott, dnb.p, N * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(n_surf, dns.p, N * 4, hipMemcpyDeviceToHost, st));
    if (n_steps) HIPCHK(hipMemcpyAsync(n_steps, dn1.p, N * 4, hipMemcpyDeviceToHost, st));
    if (n_rej) HIPCHK(hipMemcpyAsync(n_rej, dn2.p, N * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return 0;
}

// ------------------------------------------------------------------------------------
// Eigenray refinement on the device: pygenray's _find_single_eigenray (REF/eigenrays.py:206-268) for all
// brackets at once.  One launch of pgr_eigen_step per iteration applies the reference's loop body to the
// result of the previous trial fan and writes the next trial rays' initial states; the fan kernel runs
// between two of them (finished brackets carry a NaN y0 and are skipped, PGR_SKIP_NAN_Y0).
// ------------------------------------------------------------------------------------
struct EigenState {
    double* th1; double* th2; double* z1; double* z2;   // bracket ends (user angle, stored-convention depth)
    double* theta;       // current trial angle; the found angle at the end
    double* y0;          // [nbk][3] initial states of the trial rays
    const double* end;   // [nbk][3] end states of the last trial fan (ODE convention)
    const int32_t* status;
    int32_t* state;      // 0 active, 1 found, 2 trial ray dropped, 3 iteration limit
    int32_t* n_trial;
    double* z_end; double* t_end;
    int32_t* n_active;   // [1] brackets still active after this step
};

__global__ void pgr_eigen_step(EigenState e, int64_t nbk, int first, int iter_count, int max_iter, double rd,
                               double ztol, double source_depth, double c_source)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nbk) return;
    int st = e.state[k];
    double th = e.theta[k];
    if (first) {
        // REF/eigenrays.py:118-120
        th = e.th1[k] - (e.z1[k] + rd) * (e.th2[k] - e.th1[k]) / (e.z2[k] - e.z1[k]);
    } else if (st == 0) {
        const double zr = -e.end[3 * k + 1];  // ray.z[-1], stored convention (REF/ray_objects.py:51)
        e.z_end[k] = zr;
        e.t_end[k] = e.end[3 * k + 0];
        if (e.status[k] != PGR_RAY_OK) {
            st = 2;                                                  // REF/eigenrays.py:241-245
        } else if (fabs(zr + rd) < ztol) {
            st = 1;                                                  // :247-250
        } else {
            const double s1 = e.z1[k] + rd, sr = zr + rd;
            // np.sign(ray.z[-1] + rd) == np.sign(z1 + rd)            :253-259
            const bool same = ((sr > 0) - (sr < 0)) == ((s1 > 0) - (s1 < 0));
            if (same) { e.z1[k] = zr; e.th1[k] = th; } else { e.z2[k] = zr; e.th2[k] = th; }
            th = e.th1[k] - (e.z1[k] + rd) * (e.th2[k] - e.th1[k]) / (e.z2[k] - e.z1[k]);   // :261-263
            if (iter_count > max_iter) st = 3;                       // :265-268 (checked with the count BEFORE its increment)
        }
        e.state[k] = st;
    }
    e.theta[k] = th;
    const double nan = __longlong_as_double(0x7ff8000000000000LL);
    if (st == 0) {
        // shoot_ray(theta): ODE angle = -theta (REF/launch_rays.py:251), y0 = [0, z_s, sin(radians(.)) / c] (:284-285)
        e.y0[3 * k + 0] = 0.0;
        e.y0[3 * k + 1] = source_depth;
        e.y0[3 * k + 2] = pgr_cr_sin((-th) * (M_PI / 180.0)) / c_source;
        e.n_trial[k] += 1;
        atomicAdd(e.n_active, 1);
    } else {
        e.y0[3 * k + 0] = 0.0; e.y0[3 * k + 1] = source_depth; e.y0[3 * k + 2] = nan;
    }
}

extern "C" int pgr_eigen_refine(pgr_env* env, int64_t nbk, const double* th1, const double* th2, const double* z1,
                                const double* z2, double receiver_depth, double source_depth, double source_range,
                                double receiver_range, double c_source, double rtol, double atol, uint32_t flags,
                                int64_t max_steps, double ztol, int32_t max_iter, double* theta, int32_t* state,
                                int32_t* n_trial, double* z_end, double* t_end, int32_t* launches)
{
    if (!env) return fail("pgr_eigen_refine: null env");
    if (nbk < 0) return fail("pgr_eigen_refine: negative bracket count");
    if (launches) *launches = 0;
    if (nbk == 0) return 0;
    if (!th1 || !th2 || !z1 || !z2 || !theta || !state || !n_trial || !z_end || !t_end)
        return fail("pgr_eigen_refine: null argument");
    if (!(c_source > 0) || !(ztol > 0) || max_iter < 0) return fail("pgr_eigen_refine: bad argument");
    HIPCHK(hipSetDevice(env->device));
    std::lock_guard<std::mutex> lock(env->ws_mutex);
    if (!env->stream) HIPCHK(hipStreamCreateWithFlags(&env->stream, hipStreamNonBlocking));
    hipStream_t st = env->stream;
    // one device block: 4 bracket arrays, theta, z_end, t_end (doubles), y0[3], end[3], 5 int arrays, the counter
    const size_t nd = (size_t)nbk;
    const size_t bytes = nd * 8 * (7 + 3 + 3) + nd * 4 * 6 + 256;
    DevBuf buf;
    if (buf.alloc(bytes)) return fail("pgr_eigen_refine: device allocation failed");
    double* d = (double*)buf.p;
    EigenState e{};
    e.th1 = d; e.th2 = d + nd; e.z1 = d + 2 * nd; e.z2 = d + 3 * nd; e.theta = d + 4 * nd;
    e.z_end = d + 5 * nd; e.t_end = d + 6 * nd; e.y0 = d + 7 * nd;
    double* end = d + 10 * nd;
    e.end = end;
    int32_t* ib = (int32_t*)(d + 13 * nd);
    int32_t* status = ib; e.status = status;
    e.state = ib + nd; e.n_trial = ib + 2 * nd;
    int32_t* nbott = ib + 3 * nd; int32_t* nsurf = ib + 4 * nd;
    e.n_active = ib + 5 * nd;
    HIPCHK(hipMemsetAsync(buf.p, 0, bytes, st));
    HIPCHK(hipMemcpyAsync(e.th1, th1, nd * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(e.th2, th2, nd * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(e.z1, z1, nd * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(e.z2, z2, nd * 8, hipMemcpyHostToDevice, st));
    const dim3 grid((unsigned)((nbk + 127) / 128)), block(128);
    int n_launch = 0;
    for (int it = 0;; it++) {
        // iter_count of the reference when it tests the limit after trial ray number `it`: it - 1
        HIPCHK(hipMemsetAsync(e.n_active, 0, 4, st));
        hipLaunchKernelGGL(pgr_eigen_step, grid, block, 0, st, e, nbk, it == 0 ? 1 : 0, it - 1, (int)max_iter, receiver_depth,
                           ztol, source_depth, c_source);
        HIPCHK(hipGetLastError());
        int32_t active = 0;
        HIPCHK(hipMemcpyAsync(&active, e.n_active, 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (active == 0) break;
        if (it > max_iter + 2) return fail("pgr_eigen_refine: iteration guard");
        int rc = pgr_shoot_fan_device(env, e.y0, nbk, source_range, receiver_range, nullptr, 1, rtol, atol,
                                      (flags & PGR_TERMINATE_BACKWARDS) | PGR_SKIP_NAN_Y0, max_steps, nullptr, nullptr, nullptr,
                                      end, nbott, nsurf, status, nullptr, nullptr, (void*)st);
        if (rc) return rc;
        n_launch++;
    }
    HIPCHK(hipMemcpyAsync(theta, e.theta, nd * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(z_end, e.z_end, nd * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(t_end, e.t_end, nd * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(state, e.state, nd * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(n_trial, e.n_trial, nd * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (launches) *launches = n_launch;
    return 0;
}

// ------------------------------------------------------------------------------------
// arrival-time histogram of a fan's surviving rays (BASELINE configs[4]; the reduction behind
// pygenray's time-front scatter, REF/ray_objects.py:157-222).  Bin rule = np.histogram(t,
// bins=nbins, range=(t_min, t_max)) to the bit: uniform-bin index from ((t - first) / width) * nbins,
// corrected against the np.linspace edges, last bin closed on the right; NaN and rays with
// status != 0 are skipped.  Per-workgroup counts in LDS, one global atomic per non-empty bin.
// ------------------------------------------------------------------------------------
__global__ void pgr_hist_kernel(const double* __restrict__ t, int64_t t_stride, const int32_t* __restrict__ status,
                                int64_t s_stride, int64_t N, double first, double last, int nbins,
                                unsigned long long* __restrict__ counts)
{
    extern __shared__ unsigned int hist_lds[];
    for (int i = threadIdx.x; i < nbins; i += blockDim.x) hist_lds[i] = 0;
    __syncthreads();
    const double denom = last - first;
    const double step = denom / nbins;  // np.linspace: step = delta / div; edges = arange * step + start
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < N; k += (int64_t)gridDim.x * blockDim.x) {
        if (status && status[k * s_stride] != 0) continue;
        const double v = t[k * t_stride];
        if (!((v >= first) & (v <= last))) continue;  // also drops NaN
        int idx = (int)(((v - first) / denom) * nbins);
        if (idx == nbins) idx--;
        const double e_lo = (idx == nbins) ? last : grid_at(first, step, idx);
        if (v < e_lo) idx--;
        const double e_hi = (idx + 1 >= nbins) ? last : grid_at(first, step, idx + 1);
        if ((v >= e_hi) & (idx != nbins - 1)) idx++;
        atomicAdd(&hist_lds[idx], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nbins; i += blockDim.x)
        if (hist_lds[i]) atomicAdd(&counts[i], (unsigned long long)hist_lds[i]);
}

extern "C" int pgr_arrival_histogram_device(int device, const double* t_end, int64_t t_stride,
                                            const int32_t* status, int64_t status_stride, int64_t N,
                                            double t_min, double t_max, int32_t nbins, int64_t* counts,
                                            void* stream)
{
    if ((!t_end && N > 0) || !counts || N < 0 || t_stride < 1 || (status && status_stride < 1))
        return fail("pgr_arrival_histogram_device: bad argument");
    if (nbins < 1 || nbins > 16384) return fail("pgr_arrival_histogram_device: nbins must be 1..16384");
    if (!(t_max > t_min) || !isfinite(t_min) || !isfinite(t_max))
        return fail("pgr_arrival_histogram_device: need finite t_min < t_max");
    HIPCHK(hipSetDevice(device));
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(hipMemsetAsync(counts, 0, (size_t)nbins * 8, st));
    if (N == 0) return 0;
    const int threads = 256;
    int64_t blocks = (N + threads * 8 - 1) / (threads * 8);
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(pgr_hist_kernel, dim3((unsigned)blocks), dim3(threads), (size_t)nbins * 4, st, t_end, t_stride,
                       status, status_stride, N, t_min, t_max, (int)nbins, (unsigned long long*)counts);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int pgr_debug_math(const double* a, const double* b, int64_t M, double* out9)
{
    if (!a || !b || !out9 || M <= 0) return fail("pgr_debug_math: bad argument");
    struct Buf { void* p = nullptr; ~Buf() { if (p) (void)hipFree(p); } } da, db, dout;
    HIPCHK(hipMalloc(&da.p, M * 8));
    HIPCHK(hipMalloc(&db.p, M * 8));
    HIPCHK(hipMalloc(&dout.p, M * 72));
    HIPCHK(hipMemcpy(da.p, a, M * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(db.p, b, M * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(pgr_math_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, 0,
                       (const double*)da.p, (const double*)db.p, M, (double*)dout.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out9, dout.p, M * 72, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int pgr_debug_step(pgr_env* env, const double* t, const double* y, const double* h, int64_t M,
                              double rtol, double atol, double* out11)
{
    if (!env || !t || !y || !h || !out11 || M <= 0) return fail("pgr_debug_step: bad argument");
    HIPCHK(hipSetDevice(env->device));
    DevBuf dt, dy, dh, dout;
    if (dt.alloc(M * 8) || dy.alloc(M * 24) || dh.alloc(M * 8) || dout.alloc(M * 88)) return fail("device allocation failed");
    HIPCHK(hipMemcpy(dt.p, t, M * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dy.p, y, M * 24, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dh.p, h, M * 8, hipMemcpyHostToDevice));
    const dim3 grid((unsigned)((M + 63) / 64)), block(64);
    const int zm = env->d.z_simple ? ((env->d.dz == 1.0) ? 4 : 1) : 0;
    if (zm == 4)
        hipLaunchKernelGGL((pgr_step_kernel<4>), grid, block, 0, 0, env->d_dev, (const double*)dt.p, (const double*)dy.p,
                           (const double*)dh.p, M, rtol, atol, (double*)dout.p);
    else if (zm == 1)
        hipLaunchKernelGGL((pgr_step_kernel<1>), grid, block, 0, 0, env->d_dev, (const double*)dt.p, (const double*)dy.p,
                           (const double*)dh.p, M, rtol, atol, (double*)dout.p);
    else
        hipLaunchKernelGGL((pgr_step_kernel<0>), grid, block, 0, 0, env->d_dev, (const double*)dt.p, (const double*)dy.p,
                           (const double*)dh.p, M, rtol, atol, (double*)dout.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out11, dout.p, M * 88, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int pgr_eval_points(pgr_env* env, const double* x, const double* y, int64_t M, double* out10)
{
    if (!env || !x || !y || !out10) return fail("pgr_eval_points: null argument");
    if (M <= 0) return 0;
    HIPCHK(hipSetDevice(env->device));
    DevBuf dx, dy, dout;
    if (dx.alloc(M * 8) || dy.alloc(M * 24) || dout.alloc(M * 80)) return fail("device allocation failed");
    HIPCHK(hipMemcpy(dx.p, x, M * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dy.p, y, M * 24, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(pgr_eval_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, 0, env->d,
                       (const double*)dx.p, (const double*)dy.p, M, (double*)dout.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out10, dout.p, M * 80, hipMemcpyDeviceToHost));
    return 0;
}
