// pgr_fan_handle.h -- initial states on the device and fans whose results stay in HBM (pgr_fan_*): launch, wait, fetch on demand.
// (Part of the ONE translation unit pgr_hip.hip, included there in this order; not a stand-alone header.)
#ifndef PGR_FAN_HANDLE_H
#define PGR_FAN_HANDLE_H

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
    bool blocked = false;   // T / Z / P are held sample-blocked, [ceil(S/4)][N][4] (PGR_SAMPLE_BLOCKED): every fetch un-blocks
    void* buf = nullptr;
    size_t buf_bytes = 0;
    double *y0 = nullptr, *r = nullptr, *T = nullptr, *Z = nullptr, *P = nullptr, *end = nullptr;
    int32_t *nb = nullptr, *ns = nullptr, *st = nullptr, *n1 = nullptr, *n2 = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
    std::vector<int32_t> status_host;
    std::vector<int> keep;
    // un-blocking / compaction scratch of pgr_fan_fetch_samples: grow-only, kept for the next fetch of this fan (a fetch per
    // array -- ts, then zs, then ps -- re-uses it) and freed with the handle: no hipMalloc / hipFree per fetch (hipFree waits
    // for the whole device, i.e. for other fans in flight on other streams)
    void* scratch = nullptr;
    size_t scratch_bytes = 0;
    std::mutex m;
};

extern "C" void pgr_fan_destroy(pgr_fan* f)
{
    if (!f) return;
    pgr_env* env = f->env;
    (void)hipSetDevice(env->device);
    if (f->done) { (void)hipEventSynchronize(f->done); (void)hipEventDestroy(f->done); }
    if (f->scratch) (void)hipFree(f->scratch);
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
    // environments whose tables stay in HBM / L2: the sample-blocked kernel, un-blocked when the samples are fetched
    f->blocked = f->save && !(flags & PGR_EXACT_SAMPLES) && blocked_layout_fits(env);
    if (f->blocked) f->flags |= PGR_SAMPLE_BLOCKED;
    const size_t ns_bytes = (size_t)N * (size_t)(f->blocked ? 4 * ((S + 3) / 4) : S) * sizeof(double);
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
    const size_t ns_bytes = (size_t)f->N * (size_t)f->S * sizeof(double);   // (what reaches the caller: [S][N], or less)
    std::vector<D2HJob> jobs;
    std::vector<const double*> src;
    if (T) { jobs.push_back({T, f->T, ns_bytes}); src.push_back(f->T); }
    if (z) { jobs.push_back({z, f->Z, ns_bytes}); src.push_back(f->Z); }
    if (p) { jobs.push_back({p, f->P, ns_bytes}); src.push_back(f->P); }
    if (jobs.empty()) return 0;
    hipStream_t st = f->stream;
    auto ready = [&](std::vector<D2HJob>& jb) -> int {
        int rc = fan_finish(f);
        if (rc) return rc;
        const bool squeeze = compact && f->M != f->N;
        if (!squeeze && !f->blocked) return 0;
        const int64_t M = squeeze ? f->M : f->N;
        const size_t mbytes = (size_t)f->S * (size_t)M * sizeof(double), piece = (mbytes + 255) & ~(size_t)255;
        if (M > 0) {
            const size_t need = jb.size() * piece + (((size_t)M * sizeof(int) + 255) & ~(size_t)255);
            if (need > f->scratch_bytes) {
                if (f->scratch) (void)hipFree(f->scratch);
                f->scratch = nullptr; f->scratch_bytes = 0;
                if (hipMalloc(&f->scratch, need) != hipSuccess) { f->scratch = nullptr; return fail("pgr_fan_fetch_samples: device allocation of the un-blocking scratch failed"); }
                f->scratch_bytes = need;
            }
            int* didx = nullptr;
            if (squeeze) {
                didx = (int*)((char*)f->scratch + jb.size() * piece);
                HIPCHK(hipMemcpyAsync(didx, f->keep.data(), (size_t)M * sizeof(int), hipMemcpyHostToDevice, st));
            }
            for (size_t a3 = 0; a3 < jb.size(); a3++) {
                double* t = (double*)((char*)f->scratch + a3 * piece);
                if (f->blocked)
                    hipLaunchKernelGGL(pgr_unblock_cols, dim3((unsigned)((M + 255) / 256), (unsigned)((f->S + 3) / 4)), dim3(256), 0, st,
                                       (const double*)jb[a3].src, t, (const int*)didx, M, f->N, (int)f->S);
                else
                    hipLaunchKernelGGL(pgr_gather_cols, dim3((unsigned)((M + 255) / 256), (unsigned)f->S), dim3(256), 0, st,
                                       (const double*)jb[a3].src, t, (const int*)didx, M, f->N);
                HIPCHK(hipGetLastError());
                jb[a3].src = t;
            }
        }
        for (auto& q : jb) q.bytes = mbytes;
        return 0;
    };
    return d2h_pipelined(jobs, st, f->env->device, ready);
}

#endif  // PGR_FAN_HANDLE_H
