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
#include <stdlib.h>
#include <time.h>
#include <float.h>
#include <string>
#include <vector>
#include <mutex>
#include <thread>
#include <atomic>
#include <initializer_list>

#include "../../include/pgr.h"
#include "pgr_crmath.h"
#include "pgr_device.h"      // descriptor, arithmetic, look-ups, events, dense output, stage macros
#include "pgr_fan_kernel.h"  // the fan kernel

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

// PGR_TRACE=1 in the environment: wall-clock marks of the host-pointer paths on stderr (diagnostics)
static bool trace_on() { static const bool on = getenv("PGR_TRACE") != nullptr; return on; }
static double trace_now()
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}
#define PGR_MARK(t0, what) do { if (trace_on()) fprintf(stderr, "[pgr] %8.2f ms  %s\n", trace_now() - (t0), what); } while (0)

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
    int park_lanes = 64, park_trips = 10;
    int place = 2;                    // 0 off, 1 issue priorities only, 2 cost-aware placement + priorities
    hipStream_t stream = nullptr;     // the host-pointer entry's own stream (created on first use)
    EnvDev d{};
    const EnvDev* d_dev = nullptr;  // device copy of `d` (kernel argument by pointer)
    // grow-only staging workspace of the host-pointer entry (kept while <= 256 MB so the many
    // small fans of an eigenray search do not pay 11 hipMalloc/hipFree per call)
    void* ws = nullptr;
    size_t ws_bytes = 0;
    // device buffers of destroyed pgr_fan handles, kept for the next one (hipMalloc + hipFree of 2.4 GB per fan cost
    // more than the kernel's launch; hipFree also waits for the whole device): at most 4 buffers / 64 GB
    std::vector<std::pair<void*, size_t>> fan_pool;
    std::mutex fan_pool_mutex;
    // device-resident fans (pgr_fan_*) that still point at this environment, and whether pgr_env_destroy has been called
    // meanwhile (the last fan to go then releases the environment): both under fan_pool_mutex
    int live_fans = 0;
    bool doomed = false;
    void* stage = nullptr;   // page-locked host staging of the compacted per-ray fetch (grow-only)
    size_t stage_bytes = 0;
    void* ws2 = nullptr;   // second grow-only workspace: the compacted trajectories of PGR_COMPACT
    size_t ws2_bytes = 0;
    std::mutex ws_mutex;
    // small buffers for the per-launch wave placement (cost[waves] + map[slots]): a slot is handed to a launch and
    // an event is recorded on that launch's stream behind its fan kernel; the slot is taken again only when the
    // event has completed -- however many launches are in flight on however many user streams, none reads a map
    // another launch is writing (the pool grows instead)
    struct PlaceSlot {
        void* buf = nullptr;
        size_t bytes = 0;
        hipEvent_t ev = nullptr;
        bool in_flight = false;   // claimed by a launch ...
        bool recorded = false;    // ... whose event has been recorded for THIS use (only then may hipEventQuery release it)
    };
    std::vector<PlaceSlot> place_slots;
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
        if (a < 0 || a > 3) return fail("depth search: 0 = automatic, 1 = binary search, 2 = bucket table, 3 = quadratic estimate + three nodes (no cubic)");
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


// Least-squares polynomial of degree `deg` (<= 3) through (x_k, y_k), x normalised to [0, 1] by the caller:
// normal equations in long double, Gaussian elimination with partial pivoting.  c[0..deg]; false if singular.
static bool polyfit_ld(const std::vector<long double>& x, const std::vector<long double>& y, int deg, long double* c)
{
    const int m = deg + 1;
    long double A[4][5] = {};
    for (size_t k = 0; k < x.size(); k++) {
        long double p[7];
        p[0] = 1;
        for (int q = 1; q <= 2 * deg; q++) p[q] = p[q - 1] * x[k];
        for (int r = 0; r < m; r++) {
            for (int q = 0; q < m; q++) A[r][q] += p[r + q];
            A[r][m] += p[r] * y[k];
        }
    }
    for (int col = 0; col < m; col++) {
        int piv = col;
        for (int r = col + 1; r < m; r++) if (fabsl(A[r][col]) > fabsl(A[piv][col])) piv = r;
        if (A[piv][col] == 0) return false;
        for (int q = 0; q <= m; q++) { long double t = A[col][q]; A[col][q] = A[piv][q]; A[piv][q] = t; }
        for (int r = 0; r < m; r++) {
            if (r == col) continue;
            const long double f = A[r][col] / A[col][col];
            for (int q = col; q <= m; q++) A[r][q] -= f * A[col][q];
        }
    }
    for (int r = 0; r < m; r++) c[r] = A[r][m] / A[r][r];
    return true;
}

// EnvDev::z_cubic: a cubic in z that estimates the node index of a smooth non-uniform depth grid to a small
// fraction of a cell, and a quadratic in the cell index for the reciprocal of the cell width.  Everything the
// device relies on is VERIFIED here, with the device's own operations (fma Horner forms), for every node / cell;
// a grid that fails any check keeps the three-node search (z_quad / z_bucket) or the binary search.
static void fit_cubic_index(const double* zin, int64_t nz, EnvDev& d)
{
    d.z_cubic = 0;
    d.zc_g0 = d.zc_g1 = d.zc_g2 = d.zc_g3 = d.zc_s0 = d.zc_s1 = d.zc_s2 = 0.0;
    if (d.z_uniform || nz < 8 || !(zin[nz - 1] > zin[0])) return;
    const long double z0 = zin[0], span = (long double)zin[nz - 1] - z0;
    std::vector<long double> u((size_t)nz), jj((size_t)nz);
    for (int64_t j = 0; j < nz; j++) { u[(size_t)j] = ((long double)zin[j] - z0) / span; jj[(size_t)j] = (long double)j; }
    long double c[4];
    if (!polyfit_ld(u, jj, 3, c)) return;
    // t(z) = sum_k c_k ((z - z0) / span)^k expanded in powers of z
    const long double a = 1 / span, b = -z0 / span;   // u = a z + b
    long double g[4];
    g[0] = c[0] + b * (c[1] + b * (c[2] + b * c[3]));
    g[1] = a * (c[1] + b * (2 * c[2] + 3 * b * c[3]));
    g[2] = a * a * (c[2] + 3 * b * c[3]);
    g[3] = a * a * a * c[3];
    double G[4] = {(double)g[0], (double)g[1], (double)g[2], (double)g[3]};
    auto idx = [&](double z) { return std::fma(z, std::fma(z, std::fma(z, G[3], G[2]), G[1]), G[0]); };
    // the estimate at the nodes: bias it down by its worst error (plus a margin that dwarfs the rounding of the
    // Horner form, ~1e-12 cells) so that t(zin[j]) <= j; with t increasing, a z of cell j then has
    // j - 1 <= t(z) < j + 1
    double worst = 0;
    for (int64_t j = 0; j < nz; j++) worst = std::fmax(worst, std::fabs(idx(zin[j]) - (double)j));
    if (!(worst <= 0.01)) return;
    const double bias = 2 * worst + 1e-7;
    G[0] -= bias;
    for (int64_t j = 0; j < nz; j++) {
        const double t = idx(zin[j]);
        if (!(t <= (double)j - 0.5e-7) || !(t >= (double)j - 0.05)) return;
        // t'(z) > 0 at every node and at the vertex of t' (a parabola: its extremum) when that lies inside the grid
        const double dt = G[1] + zin[j] * (2 * G[2] + 3 * zin[j] * G[3]);
        if (!(dt > 0)) return;
    }
    if (G[3] != 0) {
        const double zv = -G[2] / (3 * G[3]);
        if (zv > zin[0] && zv < zin[nz - 1] && !(G[1] + zv * (2 * G[2] + 3 * zv * G[3]) > 0)) return;
    }
    // reciprocal cell width as a quadratic in the cell index
    std::vector<long double> ju((size_t)nz - 1), inv((size_t)nz - 1);
    const long double jn = (long double)(nz - 2 > 0 ? nz - 2 : 1);
    for (int64_t j = 0; j + 1 < nz; j++) {
        const double den = zin[j + 1] - zin[j];
        if (!(den > 0)) return;
        ju[(size_t)j] = (long double)j / jn;
        inv[(size_t)j] = 1 / (long double)den;
    }
    long double sc[3];
    if (!polyfit_ld(ju, inv, 2, sc)) return;
    const double S[3] = {(double)sc[0], (double)(sc[1] / jn), (double)(sc[2] / (jn * jn))};
    for (int64_t j = 0; j + 1 < nz; j++) {
        const double den = zin[j + 1] - zin[j], jf = (double)j;
        const double y = std::fma(jf, std::fma(jf, S[2], S[1]), S[0]);
        if (!(std::fabs(std::fma(-den, y, 1.0)) <= 1e-8)) return;
    }
    d.z_cubic = 1;
    d.zc_g0 = G[0]; d.zc_g1 = G[1]; d.zc_g2 = G[2]; d.zc_g3 = G[3];
    d.zc_s0 = S[0]; d.zc_s1 = S[1]; d.zc_s2 = S[2];
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

static void env_release(pgr_env* env)
{
    (void)hipSetDevice(env->device);
    for (void* p : env->allocs) (void)hipFree(p);
    for (auto& ps : env->place_slots) {
        if (ps.ev) { (void)hipEventSynchronize(ps.ev); (void)hipEventDestroy(ps.ev); }
        if (ps.buf) (void)hipFree(ps.buf);
    }
    if (env->ws) (void)hipFree(env->ws);
    if (env->ws2) (void)hipFree(env->ws2);
    if (env->stage) (void)hipHostFree(env->stage);
    for (auto& pb : env->fan_pool) (void)hipFree(pb.first);
    if (env->stream) (void)hipStreamDestroy(env->stream);
    delete env;
}

extern "C" void pgr_env_destroy(pgr_env* env)
{
    if (!env) return;
    {
        std::lock_guard<std::mutex> lock(env->fan_pool_mutex);
        if (env->live_fans > 0) { env->doomed = true; return; }   // its fans still use its stream and tables: the last one releases it
    }
    env_release(env);
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
    fit_cubic_index(zin, nz, d);
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
    case 5: return env->d.z_cubic;
    case 6: return env->d.z_quad;
    case 7: return env->d.z_bucket;
    default: return fail("pgr_env_query: unknown property");
    }
}

// Builds the slot -> wave map for this launch on `st` (see pgr_wave_place); returns the map and
// the grid size through the references, or leaves map null when scheduling is off / not useful.
static int schedule_waves(pgr_env* env, const double* y0, int64_t N, int64_t waves, int W, hipStream_t st,
                          const int*& map_out, int64_t& blocks, int& slot_out)
{
    map_out = nullptr;
    slot_out = -1;
    if (env->place == 0 || env->waves_per_block != 0 || W < 5 || waves > (1 << 27)) return 0;
    const int64_t cus = env->num_cus;
    int mode, B;
    if (waves <= (int64_t)W * cus && W <= 8 && waves > 4 * cus) {  // single round, 1-2 waves per SIMD
        mode = env->place;                                  // 1 or 2
        B = (mode == 1) ? (int)((waves + W - 1) / W) : (int)cus;
    } else if (waves > (int64_t)W * cus) {                  // several rounds
        mode = 3;
        B = (int)((waves + W - 1) / W);
    } else {
        return 0;
    }
    std::lock_guard<std::mutex> lock(env->place_mutex);  // host threads may share an env
    size_t n_slots = (size_t)B * W;
    size_t need = ((size_t)waves * 4 + n_slots * 4 + 511) & ~(size_t)255;
    int pick = -1;
    for (size_t k = 0; k < env->place_slots.size() && pick < 0; k++) {
        pgr_env::PlaceSlot& ps = env->place_slots[k];
        // (a slot that is claimed but whose event has not been recorded yet -- another host thread between its
        // schedule_waves and its launch -- still carries the completed record of its previous use: not reclaimable)
        if (ps.in_flight && ps.recorded && hipEventQuery(ps.ev) == hipSuccess) ps.in_flight = false;
        if (!ps.in_flight) pick = (int)k;
    }
    if (pick < 0) {
        if (env->place_slots.size() >= 4096) return fail("pgr_shoot_fan: more than 4096 fans in flight on one environment");
        env->place_slots.emplace_back();
        pick = (int)env->place_slots.size() - 1;
        HIPCHK(hipEventCreateWithFlags(&env->place_slots[pick].ev, hipEventDisableTiming));
    }
    pgr_env::PlaceSlot& ps = env->place_slots[pick];
    if (need > ps.bytes) {
        if (ps.buf) (void)hipFree(ps.buf);    // (not in flight: nobody reads it)
        ps.buf = nullptr; ps.bytes = 0;
        const size_t sz = need > 65536 ? need : 65536;
        HIPCHK(hipMalloc(&ps.buf, sz));
        ps.bytes = sz;
    }
    ps.in_flight = true;   // (the event is recorded by the caller behind the fan kernel: PlaceGuard)
    ps.recorded = false;
    slot_out = pick;
    char* slot = (char*)ps.buf;
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
    // solve_ivp's validate_tol (SCIPY/common.py:44-51): an rtol below 100 EPS is raised to it (SciPy warns)
    if (rtol < 100 * DBL_EPSILON) rtol = 100 * DBL_EPSILON;
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
        if (D.z_cubic && env->depth_search == 0) {
            if (env->range_indep && tab_bytes + zq_bytes <= env->max_lds) { lds_tab = true; zm = 5; zx_bytes = zq_bytes; }
            else if (zq_bytes <= env->max_lds) { lds_tab = false; zm = 5; zx_bytes = zq_bytes; }
        }
        if (zm == 0 && D.z_quad && (env->depth_search == 0 || env->depth_search == 3)) {
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
    int place_slot = -1;
    // the claimed placement slot becomes reclaimable when everything queued on `st` so far has run: its event is recorded
    // behind the fan kernel, or -- on an early error return -- behind the map-building kernels already queued
    struct PlaceGuard {
        pgr_env* env; hipStream_t st; int& slot;
        void release() {
            if (slot < 0) return;
            std::lock_guard<std::mutex> lock(env->place_mutex);
            pgr_env::PlaceSlot& ps = env->place_slots[slot];
            if (hipEventRecord(ps.ev, st) == hipSuccess) ps.recorded = true;
            else { (void)hipStreamSynchronize(st); ps.in_flight = false; }
            slot = -1;
        }
        ~PlaceGuard() { release(); }
    } guard{env, st, place_slot};
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
        if (schedule_waves(env, y0, N, waves, wpb, st, a.wave_map, blocks, place_slot)) return -1;
        lds = tab_bytes + zx_bytes;
    } else {
        const int cap = 8;
        wpb = env->waves_per_block ? env->waves_per_block : 4;
        if (wpb > cap) wpb = cap;
        blocks = (waves + wpb - 1) / wpb;
        // the same scheduling; a fan too small for it keeps 4-wave workgroups
        if (waves > 4 * (int64_t)env->num_cus) {
            int W = waves <= cap * (int64_t)env->num_cus ? (int)((waves + env->num_cus - 1) / env->num_cus) : cap;
            const int* m = nullptr;
            int64_t nb2 = blocks;
            if (schedule_waves(env, y0, N, waves, W, st, m, nb2, place_slot)) return -1;
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
    // PGR_SAMPLE_BLOCKED: 6 KB of per-lane sample staging per wave behind everything else
    a.blk_lds_off = 0;
    if (flags & PGR_SAMPLE_BLOCKED) {
        if (!save || !(flags & PGR_SAMPLE_MAJOR)) return fail("pgr_shoot_fan: PGR_SAMPLE_BLOCKED goes with trajectories and PGR_SAMPLE_MAJOR");
        if (lds_tab) return fail("pgr_shoot_fan: PGR_SAMPLE_BLOCKED is for environments whose tables stay in HBM (this one is on the LDS-table path)");
        if (!a.save_formula || (flags & PGR_EXACT_SAMPLES)) return fail("pgr_shoot_fan: PGR_SAMPLE_BLOCKED needs a linspace save grid (PGR_SAVE_LINSPACE) and the default sample form");
        const size_t at = (lds + 15) & ~(size_t)15, need = (size_t)wpb * 6144;
        if (at + need > env->max_lds) return fail("pgr_shoot_fan: no LDS left for PGR_SAMPLE_BLOCKED");
        a.blk_lds_off = (int)at; lds = at + need;
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
    if (flags & PGR_SAMPLE_BLOCKED) {   // (HBM-table path, trajectories, sample-major: checked above)
        if (zm == 1) PGR_LAUNCH1(false, 1, 3); else if (zm == 2) PGR_LAUNCH1(false, 2, 3);
        else if (zm == 3) PGR_LAUNCH1(false, 3, 3); else if (zm == 4) PGR_LAUNCH1(false, 4, 3); else if (zm == 5) PGR_LAUNCH1(false, 5, 3);
        else PGR_LAUNCH1(false, 0, 3);
    } else if (lds_tab) {
        if (zm == 1) PGR_LAUNCH(true, 1); else if (zm == 2) PGR_LAUNCH(true, 2);
        else if (zm == 3) PGR_LAUNCH(true, 3); else if (zm == 4) PGR_LAUNCH(true, 4); else if (zm == 5) PGR_LAUNCH(true, 5);
        else PGR_LAUNCH(true, 0);
    } else {
        if (zm == 1) PGR_LAUNCH(false, 1); else if (zm == 2) PGR_LAUNCH(false, 2);
        else if (zm == 3) PGR_LAUNCH(false, 3); else if (zm == 4) PGR_LAUNCH(false, 4); else if (zm == 5) PGR_LAUNCH(false, 5);
        else PGR_LAUNCH(false, 0);
    }
#undef PGR_LAUNCH
#undef PGR_LAUNCH1
    const hipError_t launch_err = hipGetLastError();
    // (the placement map is this launch's until its fan kernel has run: `guard` records the slot's event on `st` here
    // and on every error return between the slot's pick and this point)
    guard.release();
    if (launch_err != hipSuccess) return fail(std::string("fan kernel launch: ") + hipGetErrorString(launch_err));
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

// Device -> host copy of a list of (large) arrays into the caller's pageable buffers, pipelined with the page
// faults those buffers still owe.  Measured on the one-GPU box (scripts/probes/pcie_probe2.py, 0.8 GB pieces): a D2H
// copy into never-touched NumPy memory runs at the page-fault rate of one thread (15-17 GB/s), into touched memory
// at 56 GB/s; touching 0.8 GB from 16 threads takes 6 ms.  So: helper threads fault the destination pages in, IN
// ORDER (every page's first byte is read and written back unchanged -- a write access, so the page is really
// allocated, but a reused buffer keeps what the copies do not overwrite), and publish how far they are; the calling
// thread waits for `ready` (the kernel, typically running meanwhile), then copies piece by piece as soon as a
// piece's pages are there.  The copies of the first array overlap the faults of the following ones.
namespace {
struct D2HJob { void* dst; const void* src; size_t bytes; };

struct OrderedPrefault {
    std::vector<D2HJob> jobs;
    std::vector<size_t> start;          // byte offset of each job in the concatenation
    size_t total = 0;
    static constexpr size_t kPiece = (size_t)16 << 20;
    std::vector<std::thread> th;
    std::atomic<size_t> next_piece{0};
    std::vector<std::atomic<unsigned char>> done;   // per piece
    std::vector<std::atomic<int>> reg;              // per job: 0 pages not all there, 1 being registered, 2 registered, 3 registration failed
    size_t n_pieces = 0;
    int device = 0;

    OrderedPrefault(const std::vector<D2HJob>& j, int dev) : jobs(j), device(dev)
    {
        for (auto& q : jobs) { start.push_back(total); total += q.bytes; }
        n_pieces = (total + kPiece - 1) / kPiece;
        done = std::vector<std::atomic<unsigned char>>(n_pieces);
        for (auto& d : done) d.store(0);
        reg = std::vector<std::atomic<int>>(jobs.size());
        for (auto& r : reg) r.store(0);
    }
    // touch the pages of the concatenation's bytes [a, b)
    void touch(size_t a, size_t b)
    {
        for (size_t k = 0; k < jobs.size(); k++) {
            const size_t lo = a > start[k] ? a : start[k], hi = b < start[k] + jobs[k].bytes ? b : start[k] + jobs[k].bytes;
            if (lo >= hi) continue;
            char* base = (char*)jobs[k].dst;
            size_t o = lo - start[k];
            const size_t e = hi - start[k];
            const size_t first_page = ((uintptr_t)(base + o) + 4095) & ~(uintptr_t)4095;
            { volatile char* c = (volatile char*)base + o; *c = *c; }
            for (uintptr_t q = first_page; q < (uintptr_t)(base + e); q += 4096) { volatile char* c = (volatile char*)q; *c = *c; }
        }
    }
    bool job_pages_there(size_t k) const
    {
        if (jobs[k].bytes == 0) return true;
        for (size_t pc = start[k] / kPiece; pc <= (start[k] + jobs[k].bytes - 1) / kPiece; pc++)
            if (!done[pc].load(std::memory_order_acquire)) return false;
        return true;
    }
    void run(unsigned nt)
    {
        for (unsigned t = 0; t < nt; t++)
            th.emplace_back([this]() {
                bool dev_set = false;
                for (;;) {
                    const size_t pc = next_piece.fetch_add(1);
                    if (pc >= n_pieces) break;
                    const size_t a = pc * kPiece, b = a + kPiece < total ? a + kPiece : total;
                    touch(a, b);
                    done[pc].store(1, std::memory_order_release);
                    // whoever completes an array's pages page-locks it (2 ms per 0.8 GB once the pages exist; 33 ms
                    // when they do not): the copy into it is then one DMA at the link's rate instead of the
                    // runtime's staged copy (57 against 49 GB/s, scripts/probes/pcie_probe2.py)
                    for (size_t k = 0; k < jobs.size(); k++) {
                        if (start[k] + jobs[k].bytes <= a || start[k] >= b) continue;
                        int expect = 0;
                        if (job_pages_there(k) && reg[k].compare_exchange_strong(expect, 1)) {
                            if (!dev_set) { (void)hipSetDevice(device); dev_set = true; }
                            const bool ok = jobs[k].bytes > 0 && hipHostRegister(jobs[k].dst, jobs[k].bytes, hipHostRegisterDefault) == hipSuccess;
                            if (!ok) (void)hipGetLastError();
                            reg[k].store(ok ? 2 : 3, std::memory_order_release);
                        }
                    }
                }
            });
    }
    void wait_piece(size_t pc) { while (!done[pc].load(std::memory_order_acquire)) std::this_thread::yield(); }
    int wait_registered(size_t k)
    {
        int v;
        while ((v = reg[k].load(std::memory_order_acquire)) < 2) std::this_thread::yield();
        return v;
    }
    hipStream_t stream = nullptr;       // the copies' stream, once one has been enqueued
    bool stream_used = false;
    double t0 = 0;                      // (PGR_TRACE)
    ~OrderedPrefault()
    {
        for (auto& t : th) t.join();
        // (an error return between two copies gets here with DMAs still in flight: never unlock memory under them)
        if (stream_used) (void)hipStreamSynchronize(stream);
        for (size_t k = 0; k < jobs.size(); k++)
            if (reg[k].load() == 2) (void)hipHostUnregister(jobs[k].dst);
        if (t0 != 0) PGR_MARK(t0, "destination buffers unlocked");   // (0.1-0.4 ms for eighteen 128 MB sub-jobs)
    }
};
}  // namespace

// `ready`: called once before the first copy (waits for the kernel and may decide, from the status array, to
// replace the jobs' sources -- the compaction of dropped rays); returns 0 or an error
template <class Ready>
static int d2h_pipelined(std::vector<D2HJob> jobs, hipStream_t st, int device, Ready ready)
{
    std::vector<D2HJob> whole;
    std::vector<size_t> sub_of;
    size_t total = 0;
    for (auto& q : jobs) total += q.bytes;
    if (total < ((size_t)32 << 20)) {    // small: not worth threads
        int rc = ready(jobs);
        if (rc) return rc;
        for (auto& q : jobs) HIPCHK(hipMemcpyAsync(q.dst, q.src, q.bytes, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        return 0;
    }
    const double t0 = trace_now();
    {   // arrays are cut into sub-jobs of <= 128 MB: the first one is faulted in and page-locked long before the kernel
        // ends, and the copy stream never waits for a whole array's registration
        std::vector<D2HJob> cut;
        const size_t kSub = (size_t)128 << 20;
        for (auto& q : jobs)
            for (size_t o = 0; o < q.bytes; o += kSub)
                cut.push_back({(char*)q.dst + o, (const char*)q.src + o, q.bytes - o < kSub ? q.bytes - o : kSub});
        sub_of.clear();
        for (size_t k = 0; k < jobs.size(); k++)
            for (size_t o = 0; o < jobs[k].bytes; o += kSub) sub_of.push_back(k);
        whole = jobs;
        jobs = cut;
    }
    OrderedPrefault pf(jobs, device);
    pf.stream = st;
    if (trace_on()) pf.t0 = t0;
    unsigned nt = std::thread::hardware_concurrency();
    nt = nt < 1 ? 1 : (nt > 16 ? 16 : nt);
    pf.run(nt);
    PGR_MARK(t0, "prefault threads started");
    int rc = ready(whole);   // (sources may change and sizes shrink; the destinations stay)
    if (rc) { HIPCHK(hipStreamSynchronize(st)); return rc; }
    PGR_MARK(t0, "kernel finished, sources ready");
    // the sub-jobs of the (possibly re-sourced, shortened) arrays
    std::vector<D2HJob> live = jobs;
    {
        std::vector<size_t> seen(whole.size(), 0);
        for (size_t j = 0; j < live.size(); j++) {
            const size_t k = sub_of[j], o = seen[k];
            seen[k] += jobs[j].bytes;
            live[j].src = (const char*)whole[k].src + o;
            live[j].bytes = o >= whole[k].bytes ? 0 : (whole[k].bytes - o < jobs[j].bytes ? whole[k].bytes - o : jobs[j].bytes);
        }
    }
    pf.stream_used = true;
    for (size_t k = 0; k < live.size(); k++) {
        const D2HJob& q = live[k];
        if (q.bytes == 0) continue;
        // pages there but not page-locked yet (locking is slow while the helper threads still fault pages in): do not
        // wait for it -- claim the sub-job and copy it the staged way (49 GB/s instead of 57, but now)
        for (size_t pc = pf.start[k] / OrderedPrefault::kPiece; pc <= (pf.start[k] + pf.jobs[k].bytes - 1) / OrderedPrefault::kPiece; pc++)
            pf.wait_piece(pc);
        int expect = 0;
        if (pf.reg[k].compare_exchange_strong(expect, 4)) {
            HIPCHK(hipMemcpyAsync(q.dst, q.src, q.bytes, hipMemcpyDeviceToHost, st));
            if (trace_on() && (k == 0 || k + 1 == live.size()))
                fprintf(stderr, "[pgr] %8.2f ms  sub-job %zu of %zu: %zu MB, staged copy, returned\n", trace_now() - t0, k, live.size(), q.bytes >> 20);
            continue;
        }
        if (pf.wait_registered(k) == 2) {
            HIPCHK(hipMemcpyAsync(q.dst, q.src, q.bytes, hipMemcpyDeviceToHost, st));   // one DMA into page-locked memory
            if (trace_on() && (k == 0 || k + 1 == live.size()))
                fprintf(stderr, "[pgr] %8.2f ms  sub-job %zu of %zu: %zu MB into registered memory, enqueued\n", trace_now() - t0, k, live.size(), q.bytes >> 20);
            continue;
        }
        // page-locking failed (limits): staged copies, piece by piece as the pages arrive
        size_t o = 0;
        while (o < q.bytes) {
            const size_t a = pf.start[k] + o;
            size_t n = ((a / OrderedPrefault::kPiece) + 1) * OrderedPrefault::kPiece - a;
            if (n > q.bytes - o) n = q.bytes - o;
            pf.wait_piece(a / OrderedPrefault::kPiece);
            HIPCHK(hipMemcpyAsync((char*)q.dst + o, (const char*)q.src + o, n, hipMemcpyDeviceToHost, st));
            o += n;
        }
    }
    PGR_MARK(t0, "all copies issued");
    HIPCHK(hipStreamSynchronize(st));
    PGR_MARK(t0, "all copies done");
    return 0;
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
    // (the blocked layout is a device-side layout: its buffers hold 4 ceil(S/4) N doubles, this entry's hold S N)
    if (flags & PGR_SAMPLE_BLOCKED) return fail("pgr_shoot_fan: PGR_SAMPLE_BLOCKED is for pgr_shoot_fan_device (device-resident consumers)");
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
        ~Trim()
        {
            if (e->ws_bytes > ((size_t)16 << 30)) { (void)hipFree(e->ws); e->ws = nullptr; e->ws_bytes = 0; }
            if (e->ws2_bytes > ((size_t)16 << 30)) { (void)hipFree(e->ws2); e->ws2 = nullptr; e->ws2_bytes = 0; }
        }
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
    // The per-ray arrays are small; the trajectories go out through the pipelined copy (page faults of the caller's
    // -- typically fresh -- buffers in order on helper threads, starting now, while the kernel runs; copies as soon as
    // the kernel is done and a piece's pages are there).  PGR_COMPACT: dropped rays are squeezed out on the device
    // first ([S][N] -> [S][M], one pass at HBM speed into a second grow-only workspace).
    std::vector<D2HJob> jobs;
    if (save) jobs = {{T, dT.p, ns_bytes}, {z, dZ.p, ns_bytes}, {p, dP.p, ns_bytes}};
    std::vector<int> keep;   // (outlives the asynchronous upload of the index list)
    auto ready = [&](std::vector<D2HJob>& jb) -> int {
        HIPCHK(hipMemcpyAsync(status, dst.p, N * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));       // the kernel has finished
        if (!(save && (flags & PGR_COMPACT))) return 0;
        if (!(flags & PGR_SAMPLE_MAJOR)) return fail("pgr_shoot_fan: PGR_COMPACT needs PGR_SAMPLE_MAJOR");
        if (N > 0x7fffffff) return fail("pgr_shoot_fan: PGR_COMPACT supports at most 2^31 rays per call");
        keep.reserve((size_t)N);
        for (int64_t k = 0; k < N; k++) if (status[k] == 0) keep.push_back((int)k);
        const int64_t M = (int64_t)keep.size();
        if (M == N) return 0;
        const size_t mbytes = (size_t)S * (size_t)M * sizeof(double), piece = (mbytes + 255) & ~(size_t)255;
        const size_t need2 = 3 * piece + (((size_t)M * 4 + 255) & ~(size_t)255) + 256;
        if (need2 > env->ws2_bytes) {
            if (env->ws2) (void)hipFree(env->ws2);
            env->ws2 = nullptr; env->ws2_bytes = 0;
            if (hipMalloc(&env->ws2, need2) != hipSuccess) { env->ws2 = nullptr; return fail("pgr_shoot_fan: device allocation failed"); }
            env->ws2_bytes = need2;
        }
        int* didx = (int*)((char*)env->ws2 + 3 * piece);
        if (M > 0) {
            HIPCHK(hipMemcpyAsync(didx, keep.data(), (size_t)M * sizeof(int), hipMemcpyHostToDevice, st));
            for (int a3 = 0; a3 < 3; a3++) {
                double* tmp = (double*)((char*)env->ws2 + (size_t)a3 * piece);
                hipLaunchKernelGGL(pgr_gather_cols, dim3((unsigned)((M + 255) / 256), (unsigned)S), dim3(256), 0, st,
                                   (const double*)jb[a3].src, tmp, (const int*)didx, M, N);
                HIPCHK(hipGetLastError());
                jb[a3].src = tmp;
            }
        }
        for (int a3 = 0; a3 < 3; a3++) jb[a3].bytes = mbytes;
        return 0;
    };
    rc = d2h_pipelined(jobs, st, env->device, ready);
    if (rc) return rc;
    if (end_state) HIPCHK(hipMemcpyAsync(end_state, dE.p, N * 3 * sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(n_bott, dnb.p, N * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(n_surf, dns.p, N * 4, hipMemcpyDeviceToHost, st));
    if (n_steps) HIPCHK(hipMemcpyAsync(n_steps, dn1.p, N * 4, hipMemcpyDeviceToHost, st));
    if (n_rej) HIPCHK(hipMemcpyAsync(n_rej, dn2.p, N * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return 0;
}

// ------------------------------------------------------------------------------------
// Initial states on the device: y0 = [0, source_depth, sin(radians(angle)) / c_source] per ray
// (REF/launch_rays.py:140-144, 284-285), the sine correctly rounded (pgr_crmath.h) -- the same arithmetic
// pgr_eigen_step uses for its trial rays.  A million-ray fan saves the host's 1e6 libm sines and the upload of y0.
// ------------------------------------------------------------------------------------
__global__ void pgr_y0_kernel(const double* __restrict__ ang_deg, int64_t N, double source_depth, double c_source,
                              double* __restrict__ y0)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= N) return;
    y0[3 * k + 0] = 0.0;
    y0[3 * k + 1] = source_depth;
    const double a = ang_deg[k];
    y0[3 * k + 2] = (a != a) ? a : pgr_cr_sin(a * (M_PI / 180.0)) / c_source;   // (a NaN angle stays NaN: a padding ray, PGR_SKIP_NAN_Y0)
}

__global__ void pgr_y0_from_p0_kernel(const double* __restrict__ p0, int64_t N, double source_depth, double* __restrict__ y0)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= N) return;
    y0[3 * k + 0] = 0.0;
    y0[3 * k + 1] = source_depth;
    y0[3 * k + 2] = p0[k];
}

extern "C" int pgr_initial_states_device(int device, const double* ode_angles_deg, int64_t N, double source_depth,
                                         double c_source, double* y0, void* stream)
{
    if (N < 0 || (N > 0 && (!ode_angles_deg || !y0))) return fail("pgr_initial_states_device: bad argument");
    if (!(c_source > 0)) return fail("pgr_initial_states_device: c_source must be positive");
    if (N == 0) return 0;
    HIPCHK(hipSetDevice(device));
    hipLaunchKernelGGL(pgr_y0_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ode_angles_deg, N,
                       source_depth, c_source, y0);
    HIPCHK(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------
// A fan whose results stay in HBM (pgr_fan_*): launch, come back at once, fetch what is wanted when it is
// wanted.  pygenray hands its caller a RayFan of host arrays (REF/launch_rays.py:166-186); most callers then read a
// few columns of it (find_eigenrays: the end depths, REF/eigenrays.py:65-79) -- the 2.4 GB of a 1e5 x 1001 fan cross
// PCIe (43 ms, 8x the kernel) only if somebody asks for them.
// ------------------------------------------------------------------------------------
struct pgr_fan {
    pgr_env* env = nullptr;
    int64_t N = 0, M = -1;
    int32_t S = 0;
    uint32_t flags = 0;
    bool save = false, finished = false;
    void* buf = nullptr;
    size_t buf_bytes = 0;
    double *y0 = nullptr, *r = nullptr, *T = nullptr, *Z = nullptr, *P = nullptr, *end = nullptr;
    int32_t *nb = nullptr, *ns = nullptr, *st = nullptr, *n1 = nullptr, *n2 = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
    std::vector<int32_t> status_host;
    std::vector<int> keep;
    std::mutex m;
};

extern "C" void pgr_fan_destroy(pgr_fan* f)
{
    if (!f) return;
    pgr_env* env = f->env;
    (void)hipSetDevice(env->device);
    if (f->done) { (void)hipEventSynchronize(f->done); (void)hipEventDestroy(f->done); }
    bool last = false;
    {
        std::lock_guard<std::mutex> lock(env->fan_pool_mutex);
        if (f->buf) {
            size_t held = 0;
            for (auto& pb : env->fan_pool) held += pb.second;
            if (!env->doomed && env->fan_pool.size() < 4 && held + f->buf_bytes <= ((size_t)64 << 30)) env->fan_pool.emplace_back(f->buf, f->buf_bytes);
            else (void)hipFree(f->buf);
        }
        last = (--env->live_fans == 0) && env->doomed;
    }
    delete f;
    if (last) env_release(env);   // pgr_env_destroy came first: the environment goes with its last fan
}

extern "C" int pgr_fan_launch(pgr_env* env, const double* y0, const double* ode_angles_deg, double source_depth,
                              double c_source, int64_t N, double source_range, double receiver_range, int32_t S,
                              double rtol, double atol, uint32_t flags, int64_t max_steps, pgr_fan** out)
{
    if (!env || !out) return fail("pgr_fan_launch: null argument");
    *out = nullptr;
    const double t0 = trace_now();
    if (N <= 0) return fail("pgr_fan_launch: need at least one ray");
    if (!y0 && !ode_angles_deg) return fail("pgr_fan_launch: give y0 or launch angles");
    if (S < 0) return fail("pgr_fan_launch: negative num_range_save");
    if (flags & PGR_SAMPLE_BLOCKED) return fail("pgr_fan_launch: PGR_SAMPLE_BLOCKED is for pgr_shoot_fan_device (device-resident consumers)");
    HIPCHK(hipSetDevice(env->device));
    if (!env->stream) {
        std::lock_guard<std::mutex> lock(env->ws_mutex);
        if (!env->stream) HIPCHK(hipStreamCreateWithFlags(&env->stream, hipStreamNonBlocking));
    }
    pgr_fan* f = new pgr_fan();
    f->env = env; f->N = N; f->S = S; f->save = (S > 0);
    {
        std::lock_guard<std::mutex> lock(env->fan_pool_mutex);
        env->live_fans++;
    }
    f->flags = (flags & ~(uint32_t)(PGR_COMPACT | PGR_PACKED_END | PGR_LAUNCH_SLOWNESS)) | PGR_SAMPLE_MAJOR | PGR_SAVE_LINSPACE;
    f->stream = env->stream;
    const size_t ns_bytes = (size_t)N * (size_t)S * sizeof(double);
    const size_t sizes[11] = {(size_t)N * 24, (size_t)(S > 0 ? S : 1) * 8, ns_bytes, ns_bytes, ns_bytes, (size_t)N * 24,
                              (size_t)N * 4, (size_t)N * 4, (size_t)N * 4, (size_t)N * 4, (size_t)N * 4};
    size_t off[11], total = 0;
    for (int k = 0; k < 11; k++) { off[k] = total; total += (sizes[k] + 255) & ~(size_t)255; }
    {   // the smallest pooled buffer that fits (and is not more than twice too large), else a fresh one
        std::lock_guard<std::mutex> lock(env->fan_pool_mutex);
        int best = -1;
        for (size_t k = 0; k < env->fan_pool.size(); k++)
            if (env->fan_pool[k].second >= total && env->fan_pool[k].second <= 2 * total + ((size_t)1 << 20) &&
                (best < 0 || env->fan_pool[k].second < env->fan_pool[(size_t)best].second)) best = (int)k;
        if (best >= 0) {
            f->buf = env->fan_pool[(size_t)best].first; f->buf_bytes = env->fan_pool[(size_t)best].second;
            env->fan_pool.erase(env->fan_pool.begin() + best);
        }
    }
    if (!f->buf) {
        if (hipMalloc(&f->buf, total) != hipSuccess) { f->buf = nullptr; pgr_fan_destroy(f); return fail("pgr_fan_launch: device allocation failed"); }
        f->buf_bytes = total;
    }
    char* b = (char*)f->buf;
    f->y0 = (double*)(b + off[0]); f->r = (double*)(b + off[1]); f->T = (double*)(b + off[2]); f->Z = (double*)(b + off[3]);
    f->P = (double*)(b + off[4]); f->end = (double*)(b + off[5]); f->nb = (int32_t*)(b + off[6]); f->ns = (int32_t*)(b + off[7]);
    f->st = (int32_t*)(b + off[8]); f->n1 = (int32_t*)(b + off[9]); f->n2 = (int32_t*)(b + off[10]);
    hipStream_t st = f->stream;
    hipEvent_t up = nullptr;
    int rc = 0;
    do {
        if (hipEventCreateWithFlags(&f->done, hipEventDisableTiming) != hipSuccess) { rc = fail("pgr_fan_launch: event"); break; }
        if (y0) {
            if (hipMemcpyAsync(f->y0, y0, (size_t)N * 24, hipMemcpyHostToDevice, st) != hipSuccess) { rc = fail("pgr_fan_launch: upload of y0"); break; }
        } else {
            // the angles ride in the (not yet used) end_state array; y0 is computed on the device
            if (hipMemcpyAsync(f->end, ode_angles_deg, (size_t)N * 8, hipMemcpyHostToDevice, st) != hipSuccess) { rc = fail("pgr_fan_launch: upload of the angles"); break; }
            if (flags & PGR_LAUNCH_SLOWNESS) {
                // ... or assembled from the caller's own p0[k] = sin(radians(angle)) / c (REF/launch_rays.py:144): a third
                // of the bytes of y0 cross PCIe and nobody builds an [N][3] array on the host
                hipLaunchKernelGGL(pgr_y0_from_p0_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, (const double*)f->end, N,
                                   source_depth, f->y0);
                if (hipGetLastError() != hipSuccess) { rc = fail("pgr_fan_launch: y0 kernel"); break; }
            } else {
                rc = pgr_initial_states_device(env->device, f->end, N, source_depth, c_source, f->y0, (void*)st);
                if (rc) break;
            }
        }
        if (hipEventCreateWithFlags(&up, hipEventDisableTiming) != hipSuccess || hipEventRecord(up, st) != hipSuccess) { rc = fail("pgr_fan_launch: event"); break; }
        rc = pgr_shoot_fan_device(env, f->y0, N, source_range, receiver_range, f->r, S > 0 ? S : 1, rtol, atol, f->flags, max_steps,
                                  f->save ? f->T : nullptr, f->save ? f->Z : nullptr, f->save ? f->P : nullptr, f->end,
                                  f->nb, f->ns, f->st, f->n1, f->n2, (void*)st);
        if (rc) break;
        if (hipEventRecord(f->done, st) != hipSuccess) { rc = fail("pgr_fan_launch: event record"); break; }
        // the caller may release y0 / the angles when this returns: wait for the upload (not for the kernel behind it)
        if (hipEventSynchronize(up) != hipSuccess) { rc = fail("pgr_fan_launch: upload"); break; }
    } while (0);
    if (up) (void)hipEventDestroy(up);
    if (rc) { pgr_fan_destroy(f); return rc; }
    *out = f;
    PGR_MARK(t0, "pgr_fan_launch: enqueued, upload done");
    return 0;
}

// waits for the kernel, reads the status array back once and counts the surviving rays
static int fan_finish(pgr_fan* f)
{
    if (f->finished) return 0;
    HIPCHK(hipSetDevice(f->env->device));
    HIPCHK(hipEventSynchronize(f->done));
    f->status_host.resize((size_t)f->N);
    HIPCHK(hipMemcpy(f->status_host.data(), f->st, (size_t)f->N * 4, hipMemcpyDeviceToHost));
    f->keep.clear();
    for (int64_t k = 0; k < f->N; k++) if (f->status_host[(size_t)k] == 0) f->keep.push_back((int)k);
    f->M = (int64_t)f->keep.size();
    f->finished = true;
    return 0;
}

extern "C" int pgr_fan_wait(pgr_fan* f, int64_t* n_rays, int64_t* n_ok)
{
    if (!f) return fail("pgr_fan_wait: null fan");
    std::lock_guard<std::mutex> lock(f->m);
    int rc = fan_finish(f);
    if (rc) return rc;
    if (n_rays) *n_rays = f->N;
    if (n_ok) *n_ok = f->M;
    return 0;
}

extern "C" int pgr_fan_fetch_rays(pgr_fan* f, double* end_state, int32_t* n_bott, int32_t* n_surf, int32_t* status,
                                  int32_t* n_steps, int32_t* n_rej)
{
    if (!f) return fail("pgr_fan_fetch_rays: null fan");
    std::lock_guard<std::mutex> lock(f->m);
    const double t0 = trace_now();
    int rc = fan_finish(f);
    if (rc) return rc;
    PGR_MARK(t0, "pgr_fan_fetch_rays: kernel finished, status on the host");
    const size_t n = (size_t)f->N;
    if (end_state) HIPCHK(hipMemcpy(end_state, f->end, n * 24, hipMemcpyDeviceToHost));
    if (n_bott) HIPCHK(hipMemcpy(n_bott, f->nb, n * 4, hipMemcpyDeviceToHost));
    if (n_surf) HIPCHK(hipMemcpy(n_surf, f->ns, n * 4, hipMemcpyDeviceToHost));
    if (status) memcpy(status, f->status_host.data(), n * 4);
    if (n_steps) HIPCHK(hipMemcpy(n_steps, f->n1, n * 4, hipMemcpyDeviceToHost));
    if (n_rej) HIPCHK(hipMemcpy(n_rej, f->n2, n * 4, hipMemcpyDeviceToHost));
    PGR_MARK(t0, "pgr_fan_fetch_rays: done");
    return 0;
}

// The per-ray results of the SURVIVING rays only, in launch order, as pygenray's RayFan holds them (dropped rays
// vanish, REF/launch_rays.py:166-171; bounce counts as int64): the device arrays come over in one piece into a
// page-locked staging buffer of the environment (grow-only) and a few threads squeeze them into the caller's arrays.
extern "C" int pgr_fan_fetch_rays_compact(pgr_fan* f, const double* per_ray_in, double* per_ray_out, double* end_state,
                                          int64_t* n_bott, int64_t* n_surf)
{
    if (!f) return fail("pgr_fan_fetch_rays_compact: null fan");
    std::lock_guard<std::mutex> lock(f->m);
    int rc = fan_finish(f);
    if (rc) return rc;
    pgr_env* env = f->env;
    const size_t n = (size_t)f->N, M = (size_t)f->M;
    const size_t need = n * 32;     // end[N][3] doubles, n_bott[N], n_surf[N] int32
    std::lock_guard<std::mutex> wlock(env->ws_mutex);
    if (need > env->stage_bytes) {
        if (env->stage) (void)hipHostFree(env->stage);
        env->stage = nullptr; env->stage_bytes = 0;
        if (hipHostMalloc(&env->stage, need, hipHostMallocDefault) != hipSuccess) { env->stage = nullptr; return fail("pgr_fan_fetch_rays_compact: host allocation failed"); }
        env->stage_bytes = need;
    }
    char* sb = (char*)env->stage;
    const double* h_end = (const double*)sb;
    const int32_t* h_nb = (const int32_t*)(sb + n * 24);
    const int32_t* h_ns = (const int32_t*)(sb + n * 28);
    hipStream_t st = f->stream;
    if (end_state) HIPCHK(hipMemcpyAsync((void*)h_end, f->end, n * 24, hipMemcpyDeviceToHost, st));
    if (n_bott) HIPCHK(hipMemcpyAsync((void*)h_nb, f->nb, n * 4, hipMemcpyDeviceToHost, st));
    if (n_surf) HIPCHK(hipMemcpyAsync((void*)h_ns, f->ns, n * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    unsigned nt = std::thread::hardware_concurrency();
    nt = nt < 1 ? 1 : (nt > 8 ? 8 : nt);
    if (M < 200000) nt = 1;
    const int* keep = f->keep.data();
    auto work = [&](size_t m0, size_t m1) {
        for (size_t m = m0; m < m1; m++) {
            const size_t k = (size_t)keep[m];
            if (per_ray_in && per_ray_out) per_ray_out[m] = per_ray_in[k];
            if (end_state) { end_state[3 * m] = h_end[3 * k]; end_state[3 * m + 1] = h_end[3 * k + 1]; end_state[3 * m + 2] = h_end[3 * k + 2]; }
            if (n_bott) n_bott[m] = h_nb[k];
            if (n_surf) n_surf[m] = h_ns[k];
        }
    };
    if (nt == 1) work(0, M);
    else {
        std::vector<std::thread> th;
        const size_t per = (M + nt - 1) / nt;
        for (unsigned t = 0; t < nt; t++) {
            const size_t m0 = (size_t)t * per, m1 = m0 + per < M ? m0 + per : M;
            if (m0 < m1) th.emplace_back(work, m0, m1);
        }
        for (auto& t : th) t.join();
    }
    return 0;
}

extern "C" int pgr_fan_fetch_samples(pgr_fan* f, double* T, double* z, double* p, uint32_t flags)
{
    if (!f) return fail("pgr_fan_fetch_samples: null fan");
    if (!f->save) return fail("pgr_fan_fetch_samples: the fan was launched without trajectories (S = 0)");
    std::lock_guard<std::mutex> lock(f->m);
    HIPCHK(hipSetDevice(f->env->device));
    const bool compact = (flags & PGR_COMPACT) != 0;
    const size_t ns_bytes = (size_t)f->N * (size_t)f->S * sizeof(double);
    std::vector<D2HJob> jobs;
    std::vector<const double*> src;
    if (T) { jobs.push_back({T, f->T, ns_bytes}); src.push_back(f->T); }
    if (z) { jobs.push_back({z, f->Z, ns_bytes}); src.push_back(f->Z); }
    if (p) { jobs.push_back({p, f->P, ns_bytes}); src.push_back(f->P); }
    if (jobs.empty()) return 0;
    struct Tmp { std::vector<void*> p; ~Tmp() { for (void* q : p) if (q) (void)hipFree(q); } } tmp;
    hipStream_t st = f->stream;
    auto ready = [&](std::vector<D2HJob>& jb) -> int {
        int rc = fan_finish(f);
        if (rc) return rc;
        if (!compact || f->M == f->N) return 0;
        const int64_t M = f->M;
        const size_t mbytes = (size_t)f->S * (size_t)M * sizeof(double);
        if (M > 0) {
            void* didx = nullptr;
            HIPCHK(hipMalloc(&didx, (size_t)M * sizeof(int)));
            tmp.p.push_back(didx);
            HIPCHK(hipMemcpyAsync(didx, f->keep.data(), (size_t)M * sizeof(int), hipMemcpyHostToDevice, st));
            for (size_t a3 = 0; a3 < jb.size(); a3++) {
                void* t = nullptr;
                HIPCHK(hipMalloc(&t, mbytes));
                tmp.p.push_back(t);
                hipLaunchKernelGGL(pgr_gather_cols, dim3((unsigned)((M + 255) / 256), (unsigned)f->S), dim3(256), 0, st,
                                   (const double*)jb[a3].src, (double*)t, (const int*)didx, M, f->N);
                HIPCHK(hipGetLastError());
                jb[a3].src = t;
            }
        }
        for (auto& q : jb) q.bytes = mbytes;
        return 0;
    };
    return d2h_pipelined(jobs, st, f->env->device, ready);
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
    int64_t spread;      // bracket k's trial ray is ray k * spread of the trial fan (the rays between are NaN: skipped)
    const double* rd;    // [nbk] receiver depth of each bracket (the brackets of several receiver depths search together)
};

__global__ void pgr_eigen_step(EigenState e, int64_t nbk, int first, int iter_count, int max_iter,
                               double ztol, double source_depth, double c_source)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nbk) return;
    const double rd = e.rd[k];
    const int64_t r = k * e.spread;   // this bracket's ray in the trial fan
    int st = e.state[k];
    double th = e.theta[k];
    if (first) {
        // REF/eigenrays.py:118-120
        th = e.th1[k] - (e.z1[k] + rd) * (e.th2[k] - e.th1[k]) / (e.z2[k] - e.z1[k]);
    } else if (st == 0) {
        const double zr = -e.end[3 * r + 1];  // ray.z[-1], stored convention (REF/ray_objects.py:51)
        e.z_end[k] = zr;
        e.t_end[k] = e.end[3 * r + 0];
        if (e.status[r] != PGR_RAY_OK) {
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
        e.y0[3 * r + 0] = 0.0;
        e.y0[3 * r + 1] = source_depth;
        e.y0[3 * r + 2] = pgr_cr_sin((-th) * (M_PI / 180.0)) / c_source;
        e.n_trial[k] += 1;
        atomicAdd(e.n_active, 1);
    } else {
        e.y0[3 * r + 0] = 0.0; e.y0[3 * r + 1] = source_depth; e.y0[3 * r + 2] = nan;
    }
}

extern "C" int pgr_eigen_refine_depths(pgr_env* env, int64_t nbk, const double* th1, const double* th2, const double* z1,
                                       const double* z2, const double* receiver_depths, double source_depth, double source_range,
                                       double receiver_range, double c_source, double rtol, double atol, uint32_t flags,
                                       int64_t max_steps, double ztol, int32_t max_iter, double* theta, int32_t* state,
                                       int32_t* n_trial, double* z_end, double* t_end, int32_t* launches);

extern "C" int pgr_eigen_refine(pgr_env* env, int64_t nbk, const double* th1, const double* th2, const double* z1,
                                const double* z2, double receiver_depth, double source_depth, double source_range,
                                double receiver_range, double c_source, double rtol, double atol, uint32_t flags,
                                int64_t max_steps, double ztol, int32_t max_iter, double* theta, int32_t* state,
                                int32_t* n_trial, double* z_end, double* t_end, int32_t* launches)
{
    if (nbk < 0) return fail("pgr_eigen_refine: negative bracket count");
    const std::vector<double> rd((size_t)nbk, receiver_depth);
    return pgr_eigen_refine_depths(env, nbk, th1, th2, z1, z2, rd.data(), source_depth, source_range, receiver_range, c_source,
                                   rtol, atol, flags, max_steps, ztol, max_iter, theta, state, n_trial, z_end, t_end, launches);
}

// The same search with a receiver depth PER BRACKET: the brackets of all receiver depths of a find_eigenrays call
// (REF/eigenrays.py:62 loops over them) iterate together -- every iteration of the loop lasts as long as its slowest
// trial ray whatever the number of brackets, so R receiver depths cost one search instead of R.
extern "C" int pgr_eigen_refine_depths(pgr_env* env, int64_t nbk, const double* th1, const double* th2, const double* z1,
                                       const double* z2, const double* receiver_depths, double source_depth, double source_range,
                                       double receiver_range, double c_source, double rtol, double atol, uint32_t flags,
                                       int64_t max_steps, double ztol, int32_t max_iter, double* theta, int32_t* state,
                                       int32_t* n_trial, double* z_end, double* t_end, int32_t* launches)
{
    if (!env) return fail("pgr_eigen_refine: null env");
    if (nbk < 0) return fail("pgr_eigen_refine: negative bracket count");
    if (launches) *launches = 0;
    if (nbk == 0) return 0;
    if (!th1 || !th2 || !z1 || !z2 || !receiver_depths || !theta || !state || !n_trial || !z_end || !t_end)
        return fail("pgr_eigen_refine: null argument");
    if (!(c_source > 0) || !(ztol > 0) || max_iter < 0) return fail("pgr_eigen_refine: bad argument");
    HIPCHK(hipSetDevice(env->device));
    std::lock_guard<std::mutex> lock(env->ws_mutex);
    if (!env->stream) HIPCHK(hipStreamCreateWithFlags(&env->stream, hipStreamNonBlocking));
    hipStream_t st = env->stream;
    // The trial rays of different brackets have nothing in common -- launch angles anywhere in the fan, bounces at
    // different ranges: 64 of them in one wave make that wave service bounces all the time (a service costs the whole
    // wave ~22 k cycles whoever bounced) and every trial fan lasts several times its slowest ray.  So the trial fan is
    // SPREAD: bracket k's ray is ray k * spread, the rays between carry a NaN slowness and are skipped
    // (PGR_SKIP_NAN_Y0) -- up to 1024 brackets get a wave each (one per SIMD), more share waves 2, 4 ... 64 to a wave.
    int64_t per_wave = 1;
    while (per_wave < 64 && (nbk + per_wave - 1) / per_wave > 1024) per_wave *= 2;
    const int64_t spread = 64 / per_wave;
    // one device block: 4 bracket arrays, theta, z_end, t_end [nbk] (doubles), y0[3], end[3] [nbk * spread], 3 int arrays
    // [nbk * spread], 2 [nbk], the counter
    const size_t nd = (size_t)nbk, nr = (size_t)(nbk * spread);
    const size_t bytes = nd * 8 * 8 + nr * 8 * 6 + nr * 4 * 3 + nd * 4 * 2 + 256;
    // (the environment's grow-only workspace -- the host-pointer fan entry's, which this call does not use: the
    // many small searches of a receiver-depth loop pay no allocation)
    if (bytes > env->ws_bytes) {
        if (env->ws) (void)hipFree(env->ws);
        env->ws = nullptr; env->ws_bytes = 0;
        const size_t want = bytes > ((size_t)1 << 20) ? bytes : ((size_t)1 << 20);
        if (hipMalloc(&env->ws, want) != hipSuccess) { env->ws = nullptr; return fail("pgr_eigen_refine: device allocation failed"); }
        env->ws_bytes = want;
    }
    double* d = (double*)env->ws;
    EigenState e{};
    e.spread = spread;
    e.th1 = d; e.th2 = d + nd; e.z1 = d + 2 * nd; e.z2 = d + 3 * nd; e.theta = d + 4 * nd;
    e.z_end = d + 5 * nd; e.t_end = d + 6 * nd;
    double* d_rd = d + 7 * nd;
    e.rd = d_rd;
    e.y0 = d + 8 * nd;
    double* end = d + 8 * nd + 3 * nr;
    e.end = end;
    int32_t* ib = (int32_t*)(d + 8 * nd + 6 * nr);
    int32_t* status = ib; e.status = status;
    int32_t* nbott = ib + nr; int32_t* nsurf = ib + 2 * nr;
    e.state = ib + 3 * nr; e.n_trial = ib + 3 * nr + nd;
    e.n_active = ib + 3 * nr + 2 * nd;
    HIPCHK(hipMemsetAsync(env->ws, 0, bytes, st));
    HIPCHK(hipMemsetAsync(e.y0, 0xFF, nr * 24, st));   // every ray of the trial fan starts as "skipped" (an all-ones double is a NaN)
    HIPCHK(hipMemcpyAsync(e.th1, th1, nd * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(e.th2, th2, nd * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(e.z1, z1, nd * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(e.z2, z2, nd * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_rd, receiver_depths, nd * 8, hipMemcpyHostToDevice, st));
    const dim3 grid((unsigned)((nbk + 127) / 128)), block(128);
    int n_launch = 0;
    for (int it = 0;; it++) {
        // iter_count of the reference when it tests the limit after trial ray number `it`: it - 1
        HIPCHK(hipMemsetAsync(e.n_active, 0, 4, st));
        hipLaunchKernelGGL(pgr_eigen_step, grid, block, 0, st, e, nbk, it == 0 ? 1 : 0, it - 1, (int)max_iter,
                           ztol, source_depth, c_source);
        HIPCHK(hipGetLastError());
        int32_t active = 0;
        HIPCHK(hipMemcpyAsync(&active, e.n_active, 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (active == 0) break;
        if (it > max_iter + 2) return fail("pgr_eigen_refine: iteration guard");
        int rc = pgr_shoot_fan_device(env, e.y0, (int64_t)nr, source_range, receiver_range, nullptr, 1, rtol, atol,
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
