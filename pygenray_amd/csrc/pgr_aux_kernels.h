// pgr_aux_kernels.h -- the small kernels beside the fan kernel: per-wave cost + cost-aware wave placement (what the launch path
// queues in front of a fan), and the unit-level kernels of the parity tests (one step attempt, a1-a8 at points, the arithmetic building blocks).
// (Part of the ONE translation unit pgr_hip.hip, included there in this order; not a stand-alone header.)
#ifndef PGR_AUX_KERNELS_H
#define PGR_AUX_KERNELS_H

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
#if defined(PGR_TIMING) || defined(PGR_SVC_TIMING)
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

#endif  // PGR_AUX_KERNELS_H
