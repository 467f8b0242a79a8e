// pgr_device.h -- device-side building blocks of the ray-fan integrator (included by pgr_hip.hip only):
// the environment descriptor and the kernel arguments, the arithmetic (correctly rounded divide, square
// root, reciprocal square root; ulp helpers), Ctx = the reference's table look-ups, right-hand side and
// event predicates over that descriptor, the dense-output polynomial, the Runge-Kutta stage macros and the
// save grid.  What each piece restates is cited where it stands (REF / SCIPY as in pgr_hip.hip's header).
#ifndef PGR_DEVICE_H
#define PGR_DEVICE_H

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
    // ... and when a CUBIC in z itself estimates the node index to a small fraction of a cell (the flat-earth
    // transform of a uniform grid IS a cubic in the depth: 1.5e-8 cells; a linspace grid that is not bitwise
    // uniform: ~1e-12), two nodes instead of three are read and a z that belongs to the next cell is a rare
    // event with its own (out-of-line) block: t = zc_g0 + z (zc_g1 + z (zc_g2 + z zc_g3)), evaluated with fma,
    // satisfies j - 1 <= t < j + 1 for every z of cell j (zin[j] < z <= zin[j + 1]; host verified at every node
    // with the device's own arithmetic, and the cubic is increasing), so trunc(t) is j or j - 1.  The reciprocal
    // of the cell width zin[j + 1] - zin[j] is smooth too: zc_s0 + j (zc_s1 + j zc_s2) seeds the weight's
    // division to <= 1e-8 (host verified for every cell) instead of v_rcp_f64.
    int z_cubic;
    double zc_g0, zc_g1, zc_g2, zc_g3;
    double zc_s0, zc_s1, zc_s2;
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
    int blk_lds_off;      // PGR_SAMPLE_BLOCKED: byte offset of the per-lane sample staging area in LDS (768 doubles per wave)
    // persistent waves (fans of several rounds): wave_map[0 .. n_queue) is the list of 64-ray packets, most expensive
    // first; a wave claims the next one with atomicAdd(wave_queue, 1) until the list is empty.  Null: one packet per wave.
    int* wave_queue;
    int n_queue;
    int n_queue_tail;     // the last n_queue_tail entries are the FIRST packets of the workgroups' waves 4 .. 7 (4 per workgroup), not queued
};

// The fan kernel's service phase and epilogue re-read FanArgs from the kernel-argument segment (so that what only they
// need does not sit in SGPRs across the step loop): it is the kernel's SECOND argument, behind one pointer.
constexpr int kFanArgsKernargOffset = 8;
static_assert(sizeof(const EnvDev*) == kFanArgsKernargOffset && alignof(FanArgs) <= kFanArgsKernargOffset,
              "pgr_fan_kernel(const EnvDev*, FanArgs): FanArgs must start at kernel-argument offset 8 -- change "
              "kFanArgsKernargOffset together with the kernel's signature");

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
    // y made ODD (one v_or_b32): the classic exception of Newton-Raphson reciprocals is a divisor with an
    // all-ones significand, here s = RN(sqrt x) = 1 - 2^-53 (x in {1 - 2^-53, 1 - 2^-52}: a stage within
    // |p c| < 1.7e-8 of a turning point, 2 rays in 10 000 of the headline fan, found by scripts/trace_diff.py),
    // where 1/s = 1 + 2^-53 + 2^-106 sits 2^-106 above a tie: with the even neighbour y = 1 the correction below
    // is 1 + 2^-53 exactly and rounds to even, 1 instead of 1 + 2^-52; with the odd neighbour y = 1 + 2^-52 the
    // residual is r = -2^-53 + 2^-105 (exact) and y + y r = 1 + 2^-53 + 2^-157 rounds up as it must.  Everywhere
    // else y moves by at most an ulp (it is good to ~2^-51 either way).  Against IEEE 1/sqrt on the device: 0
    // differences in 13 M operands, the doubles next to 1, 1/4 and 4 included (test_arithmetic_building_blocks);
    // before: exactly those two x per binade, and the 0.008 % of the rays that met them.
    y = __longlong_as_double(__double_as_longlong(y) | 1LL);
    double r = fma(-s, y, 1.0);      // y ~ 1/s to ~4e-15
    // one correction: y (1 + r) = 1/s to ~2e-29 relative, rounded once by the fma -- RN(1/s)
    // unless 1/s lies within ~2^-96 (relative) of a rounding boundary, the same class as fdiv().
    double out = fma(y, r, y);
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
    mutable double h_zc_g0, h_zc_g1, h_zc_g2, h_zc_g3, h_zc_s0, h_zc_s1, h_zc_s2;  // ZM == 5 (mutable: the fan kernel pins them in VGPRs)
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
          h_zc_g0(e_.zc_g0), h_zc_g1(e_.zc_g1), h_zc_g2(e_.zc_g2), h_zc_g3(e_.zc_g3),
          h_zc_s0(e_.zc_s0), h_zc_s1(e_.zc_s1), h_zc_s2(e_.zc_s2),
          h_inv_dz(e_.inv_dz), h_dz(e_.dz), h_r0(e_.r0), h_dr(e_.dr), h_inv_dr(e_.inv_dr),
          h_zhi_tol(e_.zhi_tol), h_zlo_tol(e_.zlo_tol), h_b0(e_.b0), h_db(e_.db),
          h_inv_db(e_.inv_db), h_nb(e_.nb), h_b_uniform(e_.b_uniform), h_tab(e_.tab),
          h_row_stride(e_.row_stride), h_z_uniform(e_.z_uniform), h_z_pow2(e_.z_pow2), h_z0(e_.z0),
          h_zin(e_.zin),
          h_rin(e_.rin), h_nz(e_.nz), h_nr(e_.nr), h_r_uniform(e_.r_uniform)
    {
        r_lo = 1.0; r_hi = 0.0; r_yden = 1.0; r_hi2 = 0.0; r_i = 0;  // empty interval: first use refills
    }
    __device__ __forceinline__ void reset_range_cache() const
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
        // ZM == 5: the weight is formed in the arithmetic half (blend), from the two nodes read here
        double z, za, zb;
        int i, j;
    };
    __device__ __forceinline__ Fetch fetch(int i, double z) const
    {
        Fetch f;
        int j;
        if (ZM == 5) {
            // cubic index estimate (EnvDev::z_cubic): trunc(t) is the cell of z or the one below it.
            // v_cvt_i32_f64 truncates, saturates and maps NaN to 0; one v_med3_i32 clamps.  Only the READS are
            // issued here (two nodes of zin in one ds_read2_b64, the two table nodes); whether z belongs to the
            // next cell, and the weight, are settled in blend() -- behind the independent work the stage
            // macro puts between the two halves of a look-up.
            const double t = fma(z, fma(z, fma(z, h_zc_g3, h_zc_g2), h_zc_g1), h_zc_g0);
            j = clamp_index((int)t, h_nz - 2);
            f.z = z; f.j = j; f.i = i;
            f.za = lds_z[j];
            f.zb = lds_z[j + 1];
            f.wy = 0.0;
            fetch_nodes(f, i, j);
            return f;
        } else if (ZM == 2 || ZM == 3) {
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
        fetch_nodes(f, i, j);
        return f;
    }
    // the four corner nodes of (range cell i, depth cell j)
    __device__ __forceinline__ void fetch_nodes(Fetch& f, int i, int j) const
    {
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
#ifdef PGR_CELL_RECORDS
            // (experiment, round 6: the four corner nodes of cell (i, j) as ONE 64-byte record -- one line per look-up instead
            // of two pieces of two rows nz x 16 B apart; 4 x the table.  Measured, not kept: LABNOTES round 6)
            const GlobalTab row = (GlobalTab)h_tab + ((size_t)i * (size_t)(h_nz - 1) + (size_t)j) * 4;
            const d2v t00 = row[0], t01 = row[1], t10 = row[2], t11 = row[3];
#elif defined(PGR_ROW_PAIRS)
            // (experiment, round 6: rows i and i + 1 interleaved node by node, [nr - 1][nz][2]: the four corner nodes are 64
            // contiguous bytes as with the cell records, for 2 x the table instead of 4 x.  Measured: LABNOTES round 6)
            const GlobalTab row = (GlobalTab)h_tab + ((size_t)i * (size_t)h_nz + (size_t)j) * 2;
            const d2v t00 = row[0], t10 = row[1], t01 = row[2], t11 = row[3];
#else
            const GlobalTab row = (GlobalTab)h_tab + (size_t)i * h_row_stride + j;
            const d2v t00 = row[0], t01 = row[1], t10 = row[h_row_stride], t11 = row[h_row_stride + 1];
#endif
            f.v00 = make_double2(t00.x, t00.y);
            f.v01 = make_double2(t01.x, t01.y);
            f.v10 = make_double2(t10.x, t10.y);
            f.v11 = make_double2(t11.x, t11.y);
        }
    }
    // ... and the arithmetic half: the reference's four-corner blend
    __device__ __forceinline__ void blend(const Fetch& f_in, double wx, double& c, double& cp) const
    {
        Fetch f = f_in;
        if (ZM == 5) {
            // z beyond the upper node of the estimated cell: it belongs to the next one (never further,
            // EnvDev::z_cubic).  Rare (the estimate is good to ~1e-8 cells on the flat-earth grid), so the
            // re-read sits in a block of its own behind a wave-uniform test.
            // (the test is z > zb alone: a z below the deepest node also enters the block, and stays in its cell)
            if (__builtin_expect(ballot64(f.z > f.zb) != 0, 0)) {
                const bool up = (f.z > f.zb) & (f.j < h_nz - 2);
                f.j += up ? 1 : 0;
                f.za = lds_z[f.j];
                f.zb = lds_z[f.j + 1];
                fetch_nodes(f, f.i, f.j);
            }
            // (z - zin[j]) / (zin[j+1] - zin[j]), correctly rounded: the seed polynomial is good to 1e-8 (host
            // verified per cell), one Newton step squares that, the Markstein correction squares it again
            const double a = f.z - f.za, den = f.zb - f.za, jf = (double)f.j;
            double y = fma(jf, fma(jf, h_zc_s2, h_zc_s1), h_zc_s0);
            const double e = fma(-den, y, 1.0);
            y = fma(y, e, y);
            f.wy = fdiv_y(a, den, y);
        }
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
// fixed order, so a replay reproduces every bit (what the ZM = 5 trajectory instances do; the others keep K3 ... K7).
#define PGR_SB() __builtin_amdgcn_sched_barrier(0)
// -DPGR_TIMING (experiments only): s_memtime stamps along one step attempt; the time between stamp
// k-1 and stamp k accumulates in tacc[k] and comes back in n_rej[] of lanes 0..23 (scripts/phase_times.py)
#if defined(PGR_TIMING) || defined(PGR_SVC_TIMING)
#define PGR_STAMP_(k)                                                                                 \
    do {                                                                                             \
        unsigned long long _t;                                                                       \
        PGR_SB();                                                                                    \
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t) : : "memory"); \
        tacc[k] += (unsigned)_t - tprev;                                                             \
        tprev = (unsigned)_t;                                                                        \
        PGR_SB();                                                                                    \
    } while (0)
#endif
// -DPGR_TIMING: the stamps of the step attempt; -DPGR_SVC_TIMING: the stamps of the bounce SERVICE phase instead (tacc[23]
// collects everything outside it), scripts/service_times.py
#ifdef PGR_TIMING
#define PGR_STAMP(k) PGR_STAMP_(k)
#else
#define PGR_STAMP(k) do { } while (0)
#endif
#ifdef PGR_SVC_TIMING
#define PGR_SSTAMP(k) PGR_STAMP_(k)
#else
#define PGR_SSTAMP(k) do { } while (0)
#endif
// ZM == 5: the arithmetic half of a look-up starts with a (rarely taken) branch, and the compiler sinks the
// independent sums placed in the read's shadow below it -- the wave would then wait for the LDS with nothing to
// do.  An empty asm that takes the sums as operands keeps them in front of the branch.
#define PGR_KEEP2(a_, b_) do { if (ZM == 5) asm volatile("" : "+v"(a_), "+v"(b_)); } while (0)
#define PGR_KEEP6(a_, b_, c_, d_, e_, f_) do { if (ZM == 5) asm volatile("" : "+v"(a_), "+v"(b_), "+v"(c_), "+v"(d_), "+v"(e_), "+v"(f_)); } while (0)
#define PGR_RK_STAGES(T_, H_)                                                                        \
    double k30, k31, k32, k40, k41, k42, k50, k51, k52, k60, k61, k62, k70, k71, k72;                \
    PGR_RK_STAGES_BODY(T_, H_)
// (the same with K3 ... K7 declared by the caller: the fan kernel keeps them across trips for its parked lanes)
#define PGR_RK_STAGES_BODY(T_, H_)                                                                   \
    double k20, k21, k22, cs;                                                                        \
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
    PGR_KEEP6(a31, a32, a41, a42, a51, a52); PGR_KEEP2(a61, a62);                                    \
    PGR_KEEP6(bs0, bs1, bs2, es0, es1, es2);                                                         \
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
    PGR_KEEP6(a41, a42, a51, a52, a61, a62);                                                         \
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
    PGR_KEEP6(bs0, bs1, bs2, es0, es1, es2); PGR_KEEP2(a51, a52); PGR_KEEP2(a61, a62);               \
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
    PGR_KEEP6(bs0, bs1, bs2, es0, es1, es2); PGR_KEEP2(a61, a62);                                    \
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
    PGR_KEEP6(bs0, bs1, bs2, es0, es1, es2);                                                         \
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
    { double n0k_ = n0; PGR_KEEP2(n0k_, es0); PGR_KEEP2(es1, es2); }                                 \
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

#endif  // PGR_DEVICE_H
