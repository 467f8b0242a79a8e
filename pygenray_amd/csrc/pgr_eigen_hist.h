// pgr_eigen_hist.h -- what runs on a fan's end states on the device: the eigenray false-position loop (pgr_eigen_refine*) and the
// arrival-time histogram (pgr_arrival_histogram_device).
// (Part of the ONE translation unit pgr_hip.hip, included there in this order; not a stand-alone header.)
#ifndef PGR_EIGEN_HIST_H
#define PGR_EIGEN_HIST_H

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

// host_p0: the trial rays' initial slowness comes from the caller's own sin(radians(.)) / c (pgr_slowness_fn: NumPy's, in the
// Python shim -- the arithmetic the reference uses for every ray, REF/launch_rays.py:284-285); this kernel then leaves a
// NaN there and pgr_eigen_set_p0 fills the active brackets in
__global__ void pgr_eigen_step(EigenState e, int64_t nbk, int first, int iter_count, int max_iter,
                               double ztol, double source_depth, double c_source, int host_p0)
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
        e.y0[3 * r + 2] = host_p0 ? nan : pgr_cr_sin((-th) * (M_PI / 180.0)) / c_source;
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

extern "C" int pgr_eigen_refine_depths_fn(pgr_env* env, int64_t nbk, const double* th1, const double* th2, const double* z1,
                                          const double* z2, const double* receiver_depths, double source_depth, double source_range,
                                          double receiver_range, double c_source, double rtol, double atol, uint32_t flags,
                                          int64_t max_steps, double ztol, int32_t max_iter, double* theta, int32_t* state,
                                          int32_t* n_trial, double* z_end, double* t_end, int32_t* launches,
                                          pgr_slowness_fn slowness, void* slowness_user);

__global__ void pgr_eigen_set_p0(EigenState e, int64_t nbk, const double* __restrict__ p0)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nbk) return;
    if (e.state[k] == 0) e.y0[3 * k * e.spread + 2] = p0[k];
}

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
    return pgr_eigen_refine_depths_fn(env, nbk, th1, th2, z1, z2, receiver_depths, source_depth, source_range, receiver_range,
                                      c_source, rtol, atol, flags, max_steps, ztol, max_iter, theta, state, n_trial, z_end, t_end,
                                      launches, nullptr, nullptr);
}

// ... and with the trial rays' initial slowness from the CALLER (`slowness`, may be null: the device's correctly rounded
// sine then): per iteration the trial angles of the active brackets come to the host (nbk doubles), the caller turns
// ODE angles into p0 = sin(radians(angle)) / c with ITS sine -- the Python shim hands NumPy's, so that a trial ray, the
// eigenray returned and pr.shoot_ray(theta) of the same angle start from the same bits, the reference's
// (REF/launch_rays.py:284-285) -- and the slownesses go back up.  Finished brackets carry NaN angles (-> NaN p0: skipped).
extern "C" int pgr_eigen_refine_depths_fn(pgr_env* env, int64_t nbk, const double* th1, const double* th2, const double* z1,
                                          const double* z2, const double* receiver_depths, double source_depth, double source_range,
                                          double receiver_range, double c_source, double rtol, double atol, uint32_t flags,
                                          int64_t max_steps, double ztol, int32_t max_iter, double* theta, int32_t* state,
                                          int32_t* n_trial, double* z_end, double* t_end, int32_t* launches,
                                          pgr_slowness_fn slowness, void* slowness_user)
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
    const size_t bytes = nd * 8 * 8 + nr * 8 * 6 + nr * 4 * 3 + nd * 4 * 2 + 256 + (slowness ? nd * 8 + 256 : 0);
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
    double* d_p0 = (double*)(((uintptr_t)(e.n_active + 1) + 255) & ~(uintptr_t)255);   // (callback mode only; inside `bytes`)
    std::vector<double> h_ang, h_p0;
    std::vector<int32_t> h_state;
    if (slowness) { h_ang.resize(nd); h_p0.resize(nd); h_state.resize(nd); }
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
                           ztol, source_depth, c_source, slowness ? 1 : 0);
        HIPCHK(hipGetLastError());
        int32_t active = 0;
        HIPCHK(hipMemcpyAsync(&active, e.n_active, 4, hipMemcpyDeviceToHost, st));
        if (slowness) {
            HIPCHK(hipMemcpyAsync(h_ang.data(), e.theta, nd * 8, hipMemcpyDeviceToHost, st));
            HIPCHK(hipMemcpyAsync(h_state.data(), e.state, nd * 4, hipMemcpyDeviceToHost, st));
        }
        HIPCHK(hipStreamSynchronize(st));
        if (active == 0) break;
        if (slowness) {
            // shoot_ray(theta): ODE angle = -theta (REF/launch_rays.py:251); a finished bracket's ray stays a NaN
            const double qnan = std::numeric_limits<double>::quiet_NaN();
            for (size_t k = 0; k < nd; k++) h_ang[k] = (h_state[k] == 0) ? -h_ang[k] : qnan;
            slowness(h_ang.data(), (int64_t)nd, h_p0.data(), slowness_user);
            HIPCHK(hipMemcpyAsync(d_p0, h_p0.data(), nd * 8, hipMemcpyHostToDevice, st));
            hipLaunchKernelGGL(pgr_eigen_set_p0, grid, block, 0, st, e, nbk, (const double*)d_p0);
            HIPCHK(hipGetLastError());
        }
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

#endif  // PGR_EIGEN_HIST_H
