/*
 * pgr.h -- C ABI of the MI355X-native pygenray hot path (libpgr_hip.so).
 *
 * pygenray has no FFI/plugin interface; the seam this library slots into is the
 * array-level contract pygenray itself uses to cross its process boundary
 * (REF = /root/reference/src/pygenray):
 *
 *   - the 7 environment arrays  cin, cpin, rin, zin, depths, depth_ranges,
 *     bottom_angle                      REF/multi_processing.py:37-45,90
 *   - per-ray y0 = [T0, z0, p0] + scalars, i.e. the argument list of
 *     _shoot_single_ray_process / _shoot_ray_array
 *                                       REF/launch_rays.py:487-497, 325-340
 *   - per-ray result [r; T; z; p] on linspace(source_range, receiver_range,
 *     num_range_save) + n_bottom, n_surface, or "None" for a dropped ray
 *                                       REF/launch_rays.py:561-576, 745-784
 *
 * Everything is plain pointers and sizes: no torch / HIP types in signatures
 * (a HIP stream is passed as void*).  All floating point is IEEE binary64.
 * Values are in the ODE convention (depth positive down); the caller applies
 * pygenray's storage sign flip z -> -z, p -> -p (REF/ray_objects.py:51-52).
 *
 * Return value of every int function: 0 = success, <0 = error (see
 * pgr_last_error()).  No exception or signal crosses this boundary; per-ray
 * failures are reported in status[] (pygenray swallows them and drops the
 * ray, REF/launch_rays.py:578-581).
 */
#ifndef PGR_H
#define PGR_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* per-ray status codes (status[k]) */
#define PGR_RAY_OK 0             /* reached receiver_range */
#define PGR_RAY_VERTICAL 1       /* vertical_ray event      REF/launch_rays.py:443-448 */
#define PGR_RAY_BBOX 2           /* left the c(r,z) table    REF/launch_rays.py:451-456 */
#define PGR_RAY_BACKWARD 3       /* |theta_bounce| > 90      REF/launch_rays.py:474-477 */
#define PGR_RAY_STEP_TOO_SMALL 4 /* solve_ivp status -1      REF/launch_rays.py:427-430 */
#define PGR_RAY_MAX_STEPS 5      /* guard: max_steps accepted steps exceeded (no reference counterpart) */
#define PGR_RAY_BETA_RANGE 6     /* bottom-angle interp1d out of bounds (ValueError) REF/launch_rays.py:469 */
#define PGR_RAY_EVENT_ERROR 7    /* event root not bracketed (brentq ValueError) */
#define PGR_RAY_SKIPPED 8        /* not integrated: PGR_SKIP_NAN_Y0 and a NaN initial slowness (a finished eigenray bracket) */

/* flags for pgr_shoot_fan* */
#define PGR_TERMINATE_BACKWARDS 1u /* REF/launch_rays.py:19,474 (default True) */
#define PGR_SAMPLE_MAJOR 2u        /* T/z/p laid out [S][N] instead of [N][S] */
#define PGR_EXACT_BISECTION 4u     /* locate events with brentq's ~42-step bisection, the true +-1 event evaluated
                                      at every iterate (SCIPY/ivp.py:51-76), instead of the default replay of the
                                      same iterates that evaluates it only inside the rounding-noise band around
                                      the root: the same root either way */

#define PGR_SAVE_LINSPACE 8u       /* (device entry) the caller asserts r_save is exactly
                                      np.linspace(source_range, receiver_range, S); the kernel then
                                      recomputes r_save[j] = j*step + start instead of loading it.
                                      The host entry sets/clears this bit itself after checking. */
#define PGR_DEBUG_TRIPS 16u       /* diagnostics: n_rej[] receives, per wave, the number of main-loop
                                      trips (lane 0), of service phases (lane 1) and, per ray, how often
                                      the exact event bisection ran (lanes 2..63) instead */

#define PGR_EXACT_SAMPLES 32u      /* evaluate the saved samples with SciPy's own summation order
                                      (Q = K.T @ P, then h * (Q @ p) + y_old, SCIPY/rk.py:552-574)
                                      instead of the default stage-major FMA form of the same quartic
                                      (a few ulp apart; samples never feed back into the integration) */

#define PGR_STORED_SIGN 64u        /* T/z/p trajectories in pygenray's stored convention: z -> -z,
                                      p -> -p (REF/ray_objects.py:51-52); end_state stays ODE-signed */

#define PGR_COMPACT 128u           /* (host entry, with PGR_SAMPLE_MAJOR) dropped rays vanish from the
                                      trajectories as they do from a pygenray RayFan
                                      (REF/launch_rays.py:166-171): T/z/p come back as [S][M], M = number
                                      of rays with status 0, in launch order, at the start of the caller's
                                      [S][N] buffers; the per-ray arrays keep all N entries */

#define PGR_PACKED_END 256u        /* (device entry) end_state is [N][5] doubles: T, z, p, then n_bott,
                                      n_surf, status and a valid mark (1) as four int32 in the last two
                                      slots -- the end record the multi-GPU all-gather ships, written by
                                      the kernel instead of packed afterwards */

#define PGR_SAMPLE_BLOCKED 2048u   /* (with PGR_SAMPLE_MAJOR; environments whose tables stay in HBM / L2) T / z / p are written as
                                     [ceil(S / 4)][N][4]: sample j of ray k at ((j / 4) * N + k) * 4 + j % 4 -- every lane stages four
                                     consecutive samples in LDS and stores them as one full 32-byte piece per array.  The buffers hold
                                     4 * ceil(S / 4) * N doubles (rows S ... 4 * ceil(S / 4) - 1 are padding).  pgr_shoot_fan_device only
                                     (a layout for device-resident consumers), with PGR_SAVE_LINSPACE and the default sample form; an
                                     error for environments on the LDS-table path and in the host-pointer / fan-handle entries. */
#define PGR_LAUNCH_SLOWNESS 1024u   /* (pgr_fan_launch without y0) the array of launch angles holds the initial vertical
                                      slowness p0[k] = sin(radians(angle_k)) / c_source itself, computed by the caller
                                      (REF/launch_rays.py:144): y0[k] = [0, source_depth, p0[k]] is assembled on the device */

#define PGR_SKIP_NAN_Y0 512u       /* rays whose y0[k][2] is NaN are not integrated (status PGR_RAY_SKIPPED): the
                                      eigenray refinement parks its finished brackets this way */

typedef struct pgr_env pgr_env; /* opaque: environment tables resident in HBM */

/* Number of visible HIP devices (<0 on error). */
int pgr_device_count(void);

/* Upload one environment (the 7-array contract, REF/multi_processing.py:37-45) to
 * `device`.  Replaces _init_shared_memory/_unpack_shared_memory: one H2D copy instead of
 * 7 POSIX shm blocks.  cin/cpin are [nr][nz] C-contiguous (range-major), bottom_angles in
 * degrees.  Coordinates must be non-decreasing (REF/launch_rays.py:79-90) and nb >= 4
 * (cubic interp1d, REF/launch_rays.py:397-399).  Host buffers stay owned by the caller. */
int pgr_env_create(pgr_env** env, int device, const double* cin, const double* cpin,
                   const double* rin, const double* zin, int64_t nr, int64_t nz,
                   const double* depths, const double* depth_ranges, const double* bottom_angles,
                   int64_t nb);
/* Releases the tables, workspaces and stream.  While device-resident fans (pgr_fan_launch below) of this
 * environment are alive the release is deferred to the pgr_fan_destroy of the last of them; the handle must not be
 * used for new calls after pgr_env_destroy either way. */
void pgr_env_destroy(pgr_env* env);

/* Properties the kernel selection depends on (for tests / diagnostics):
 * what = 0: tables range independent (all rows bitwise equal) ; 1: zin exactly uniform ;
 *        2: rin exactly uniform ; 3: LDS-resident table path selected ; 4: device index ;
 *        5 / 6 / 7: zin qualifies for the cubic index estimate / the quadratic one / the bin table ;
 *        8: a sample-major trajectory fan of this environment is best written PGR_SAMPLE_BLOCKED (tables in HBM / L2, LDS left
 *           for the staging, PGR_OPT_API_BLOCKED on): what pgr_shoot_fan / pgr_fan_launch do by themselves and what a
 *           device-resident consumer of pgr_shoot_fan_device should ask for (the Python DeviceFan does by default). */
int pgr_env_query(const pgr_env* env, int what);

/* Shoot N rays: batched _shoot_ray_array + _interpolate_ray (REF/launch_rays.py:325-484,
 * 745-784) with SciPy's RK45 defaults (REF/launch_rays.py:670-679: rtol given, atol 1e-6,
 * dense output, 4 terminal events).
 *
 *   rtol, atol    as solve_ivp takes them: rtol > 0 (below 100 EPS it is raised to 100 EPS, SCIPY/common.py:44-51),
 *                 atol >= 0
 *   y0[N][3]      initial [T, z, p]   (REF/launch_rays.py:140-144)
 *   r_save[S]     np.linspace(source_range, receiver_range, S), computed by the caller
 *   T,z,p         [N][S] (or [S][N] with PGR_SAMPLE_MAJOR); may all be NULL = end state only.
 *                 Rays with status != 0 are filled with NaN.
 *   end_state     [N][3] exact final [T,z,p] (= last column), may be NULL
 *   n_bott,n_surf [N] bounce counts ; status [N] ; n_steps [N] accepted RK45 steps ;
 *   n_rej [N] rejected attempts (n_steps/n_rej may be NULL)
 *
 * pgr_shoot_fan takes HOST pointers (copies in/out, synchronous): everything it does is enqueued on a
 * stream owned by `env` and it waits for that stream only; calls on one env are serialised.
 * pgr_shoot_fan_device takes DEVICE pointers on env's device, enqueues on `stream`
 * (hipStream_t as void*, NULL = default stream) and returns without synchronising. */
int pgr_shoot_fan(pgr_env* env, const double* y0, int64_t N, double source_range,
                  double receiver_range, const double* r_save, int32_t S, double rtol, double atol,
                  uint32_t flags, int64_t max_steps, double* T, double* z, double* p,
                  double* end_state, int32_t* n_bott, int32_t* n_surf, int32_t* status,
                  int32_t* n_steps, int32_t* n_rej);
int pgr_shoot_fan_device(pgr_env* env, const double* y0, int64_t N, double source_range,
                         double receiver_range, const double* r_save, int32_t S, double rtol,
                         double atol, uint32_t flags, int64_t max_steps, double* T, double* z,
                         double* p, double* end_state, int32_t* n_bott, int32_t* n_surf,
                         int32_t* status, int32_t* n_steps, int32_t* n_rej, void* stream);

/* Initial states on the device (REF/launch_rays.py:140-144, 284-285): y0[k] = [0, source_depth,
 * sin(radians(ode_angles_deg[k])) / c_source], the sine correctly rounded (the arithmetic pgr_eigen_refine uses for its
 * trial rays); DEVICE pointers, enqueued on `stream`, returns without synchronising.  c_source = c at the source (the
 * caller's bilinear_interp, REF/launch_rays.py:140). */
int pgr_initial_states_device(int device, const double* ode_angles_deg, int64_t N, double source_depth,
                              double c_source, double* y0, void* stream);

/* A fan whose results stay in HBM.  pgr_fan_launch uploads y0[N][3] (HOST) -- or, when y0 is NULL, the N ODE launch
 * angles ode_angles_deg (HOST, degrees), from which the initial states are computed on the device as above --
 * enqueues the fan on the environment's stream and returns as soon as the upload is done: the kernel is still running.
 * S = num_range_save of the trajectories kept on the device ([S][N], the save grid is np.linspace(source_range,
 * receiver_range, S)); S = 0: end states only.  flags as pgr_shoot_fan (PGR_TERMINATE_BACKWARDS, PGR_EXACT_*,
 * PGR_STORED_SIGN).  pgr_fan_wait blocks until the kernel has finished and tells the ray count and how many have
 * status 0.  pgr_fan_fetch_rays copies the per-ray arrays ([N], any may be NULL) to HOST buffers; pgr_fan_fetch_samples
 * copies the trajectories asked for (T, z, p: HOST [S][N], any may be NULL; with PGR_COMPACT [S][M], dropped rays
 * squeezed out as in a pygenray RayFan, REF/launch_rays.py:166-171) -- page faults of fresh destination buffers
 * pipelined with the copies.  Both wait for the kernel themselves.  What pygenray's RayFan holds as host arrays
 * (REF/launch_rays.py:166-186) crosses PCIe only when somebody asks for it. */
typedef struct pgr_fan pgr_fan;
int pgr_fan_launch(pgr_env* env, const double* y0, const double* ode_angles_deg, double source_depth, double c_source,
                   int64_t N, double source_range, double receiver_range, int32_t S, double rtol, double atol,
                   uint32_t flags, int64_t max_steps, pgr_fan** fan);
int pgr_fan_wait(pgr_fan* fan, int64_t* n_rays, int64_t* n_ok);
int pgr_fan_fetch_rays(pgr_fan* fan, double* end_state, int32_t* n_bott, int32_t* n_surf, int32_t* status,
                       int32_t* n_steps, int32_t* n_rej);
int pgr_fan_fetch_samples(pgr_fan* fan, double* T, double* z, double* p, uint32_t flags);
/* The per-ray results of the M surviving rays only, in launch order (what a pygenray RayFan keeps): end_state [M][3],
 * n_bott / n_surf [M] as int64; per_ray_in [N] -> per_ray_out [M] squeezes any per-ray HOST array of the caller's
 * (the stored launch angles) the same way.  Any pair may be NULL.  M from pgr_fan_wait. */
int pgr_fan_fetch_rays_compact(pgr_fan* fan, const double* per_ray_in, double* per_ray_out, double* end_state,
                               int64_t* n_bott, int64_t* n_surf);
void pgr_fan_destroy(pgr_fan* fan);

/* Eigenray refinement: pygenray's _find_single_eigenray (REF/eigenrays.py:206-268) for nbk brackets at
 * once, the whole false-position loop on the device.  Per bracket k the fan rays th1[k], th2[k] (user
 * launch angles, degrees) ended at stored-convention depths z1[k], z2[k] on either side of
 * -receiver_depth.  Each iteration is: trial angle (REF/eigenrays.py:118-120, 261-263) and its initial
 * state y0 = [0, source_depth, sin(radians(-theta)) / c_source] (REF/launch_rays.py:251,284-285; the
 * sine correctly rounded) in one small kernel, ONE fan launch (end state only) over all brackets still
 * active (the others are skipped in place, PGR_SKIP_NAN_Y0), the update of the reference's loop in the
 * next small kernel: dropped ray -> failed; |z_end + receiver_depth| < ztol -> found; else the trial ray
 * replaces the bracket end on its side; give up after max_iter + 2 trial rays (REF/eigenrays.py:265-268).
 * Everything is enqueued on the environment's stream; the host only reads back the count of active
 * brackets after each iteration.  HOST arrays in and out:
 *   theta[k]  state 1: the eigenray's launch angle (user convention); state 2: the angle of the trial ray that was
 *   dropped (what the reference prints in its failure message); state 3: the next trial angle the loop would have
 *   shot;   state[k]  1 found, 2 trial ray dropped (the reference prints "Failed to find eigen ray" and gives
 *   up), 3 iteration limit;
 *   n_trial[k] trial rays shot;  z_end[k], t_end[k] stored-convention end depth / arrival time of the
 *   last trial ray.  *launches = fan launches made.  c_source = c at the source (the caller's
 *   bilinear_interp, REF/launch_rays.py:284). */
int pgr_eigen_refine(pgr_env* env, int64_t nbk, const double* th1, const double* th2, const double* z1,
                     const double* z2, double receiver_depth, double source_depth, double source_range,
                     double receiver_range, double c_source, double rtol, double atol, uint32_t flags,
                     int64_t max_steps, double ztol, int32_t max_iter, double* theta, int32_t* state,
                     int32_t* n_trial, double* z_end, double* t_end, int32_t* launches);
/* The same search with a receiver depth per bracket, receiver_depths[nbk]: find_eigenrays loops over its receiver
 * depths (REF/eigenrays.py:62) and searches the brackets of each; an iteration of the device loop lasts as long as
 * its slowest trial ray whatever the number of brackets, so the brackets of ALL receiver depths iterate together --
 * R depths cost one search instead of R.  Same results, bracket by bracket, as R calls of pgr_eigen_refine. */
int pgr_eigen_refine_depths(pgr_env* env, int64_t nbk, const double* th1, const double* th2, const double* z1,
                            const double* z2, const double* receiver_depths, double source_depth, double source_range,
                            double receiver_range, double c_source, double rtol, double atol, uint32_t flags,
                            int64_t max_steps, double ztol, int32_t max_iter, double* theta, int32_t* state,
                            int32_t* n_trial, double* z_end, double* t_end, int32_t* launches);

/* ... and with the trial rays' initial slowness computed by the CALLER: `slowness(ode_angles_deg, n, p0_out, user)` must set
 * p0_out[k] = sin(radians(ode_angles_deg[k])) / c_source for k < n (HOST arrays; a NaN angle -- a finished bracket -- must give
 * a NaN) with the caller's own sine.  pygenray computes every initial slowness with NumPy's sine (REF/launch_rays.py:284-285),
 * which is faithful, not correctly rounded: the Python shim passes NumPy's here, so that a trial ray, the eigenray handed back and
 * shoot_ray(theta) of the same angle all start from the SAME bits, the reference's.  Costs one small D2H + H2D per iteration.
 * slowness == NULL: the device's correctly rounded sine, as pgr_eigen_refine_depths.
 * The callback runs on the calling thread INSIDE the search: the environment's workspace lock is held and its buffers are in
 * use, so it must NOT call back into the same environment: pgr_shoot_fan, pgr_eigen_refine*, pgr_fan_*, pgr_env_destroy on `env`
 * would deadlock or reallocate the workspace under the loop; other environments are fine.  It must not throw / longjmp
 * through the C frames. */
typedef void (*pgr_slowness_fn)(const double* ode_angles_deg, int64_t n, double* p0_out, void* user);
int pgr_eigen_refine_depths_fn(pgr_env* env, int64_t nbk, const double* th1, const double* th2, const double* z1,
                               const double* z2, const double* receiver_depths, double source_depth, double source_range,
                               double receiver_range, double c_source, double rtol, double atol, uint32_t flags,
                               int64_t max_steps, double ztol, int32_t max_iter, double* theta, int32_t* state,
                               int32_t* n_trial, double* z_end, double* t_end, int32_t* launches,
                               pgr_slowness_fn slowness, void* slowness_user);

/* Arrival-time histogram of a fan's surviving rays on the device (BASELINE configs[4]; the
 * reduction behind pygenray's time-front scatter RayFan.plot_time_front, REF/ray_objects.py:157-222;
 * no reference counterpart for the binning itself, so it is NumPy's):
 * counts[nbins] (DEVICE, int64, overwritten) = np.histogram(T, bins=nbins, range=(t_min, t_max))[0]
 * over the rays k < N with status[k * status_stride] == 0 (status may be NULL = all rays) and T =
 * t_end[k * t_stride] not NaN; t_end / status are DEVICE pointers on `device`: e.g. end_state with
 * t_stride 3, or the packed end records (PGR_PACKED_END) with t_stride 5 and (int32*)(rec + 4) with
 * status_stride 10.  Enqueued on `stream`, returns without synchronising; every rank of a sharded
 * fan then all-reduces its counts.  nbins <= 16384. */
int pgr_arrival_histogram_device(int device, const double* t_end, int64_t t_stride,
                                 const int32_t* status, int64_t status_stride, int64_t N,
                                 double t_min, double t_max, int32_t nbins, int64_t* counts,
                                 void* stream);

/* Tuning options of ONE environment (per-ray results never depend on them; there is no process-wide
 * state: host threads that drive different GPUs hold different environments).
 *   PGR_OPT_WAVES_PER_BLOCK  a = waves (of 64 rays) per workgroup, 0 = automatic
 *   PGR_OPT_DEPTH_SEARCH     depth-cell search for a non-uniform zin: a = 0 (default) from LDS -- when a cubic
 *                            in z estimates the node index to a small fraction of a cell (the flat-earth
 *                            transform of a uniform grid: 1.5e-8 cells), the cell is trunc(t(z)) or, rarely,
 *                            the next one, and two nodes are read; else the cell is j0 or j0 + 1 with j0 =
 *                            floor(g(z) - 0.5) from a quadratic estimate g (three nodes read), else j0 =
 *                            bucket[floor((z - z0)/w)] from a bin table; a = 1 always the binary search of
 *                            np.searchsorted (REF/integration_processes.py:152-157); a = 2 the bin table;
 *                            a = 3 the quadratic estimate (no cubic).  Tests compare the four.
 *   PGR_OPT_PARK             bounce-service batching: a wave services its parked (bounced) lanes when `a`
 *                            of them wait or the oldest waited `b` step attempts (default 64, 10)
 *   PGR_OPT_PLACEMENT        cost-aware wave scheduling for fans of 1-2 waves per SIMD: a = 2 (default) the
 *                            costliest waves get a SIMD to themselves / are paired with the cheapest, plus
 *                            issue priorities by cost quartile; 1 = priorities only; 0 = strided deal
 *   PGR_OPT_PERSISTENT       fans of several rounds (more 64-ray packets than the chip holds waves): a = 1 (default)
 *                            one workgroup per CU whose waves claim packet after packet from the cost-sorted list
 *                            (most expensive first) with one atomic each -- in fans of up to two rounds the first
 *                            packet of a workgroup's waves 4 .. 7 (the SIMD partners of waves 0 .. 3) comes from the
 *                            list's cheap end, so that every steep packet starts beside a cheap one; a = 2 / 3 never /
 *                            always so; a = 0 the static deal of whole cost-sorted workgroups (what replaces the
 *                            reference's pool.imap over single rays, REF/launch_rays.py:157-164, either way)
 *   PGR_OPT_API_BLOCKED      pgr_shoot_fan (sample-major) and pgr_fan_launch on environments whose tables stay in HBM / L2:
 *                            a = 1 (default) the trajectories are integrated by the sample-blocked kernel
 *                            (PGR_SAMPLE_BLOCKED: full 32-byte stores, 1.2x instead of 2.3x the sample bytes written) and
 *                            un-blocked to [S][M] by the pass that squeezes dropped rays out on the way to the host;
 *                            a = 0 plain [S][N] rows.  The caller sees the same arrays, bit for bit. */
#define PGR_OPT_WAVES_PER_BLOCK 0
#define PGR_OPT_DEPTH_SEARCH 1
#define PGR_OPT_PARK 2
#define PGR_OPT_PLACEMENT 3
#define PGR_OPT_PERSISTENT 4
#define PGR_OPT_API_BLOCKED 5
int pgr_env_set_option(pgr_env* env, int what, int a, int b);

/* Unit-level device entry points (for parity tests of a1-a8, REF/integration_processes.py):
 * evaluate on the GPU, for M query points (x[k], y[k][3]) given as HOST arrays:
 *   out[k][0..2] = derivsrd ; out[k][3] = bilinear c ; out[k][4] = ray angle (deg) ;
 *   out[k][5..8] = surface, bottom, vertical, bbox event values (+-1) ; out[k][9] = bathymetry
 *   linear_interp at x. */
int pgr_eval_points(pgr_env* env, const double* x, const double* y, int64_t M, double* out10);

/* Accuracy probe of the kernel's arithmetic building blocks (tests only): for HOST arrays a, b
 * of length M, out[k] = { a/b, 1/b, 1/sqrt(b), sqrt(b), b**-0.2, 10*ulp(a), b**0.2, asin(a), sin(a) }
 * as the kernel computes them (the last five of them correctly rounded, csrc/pgr_crmath.h). */
int pgr_debug_math(const double* a, const double* b, int64_t M, double* out9);

/* One RK45 step attempt per query (tests / debugging): for M HOST triples (t[k], y[k][3], h[k]) the
 * device evaluates f = derivsrd(t, y), rk_step (SCIPY/rk.py:14-71), the error norm (rk.py:106-110,
 * 146-147) and the controller's 0.9 * error_norm ** -0.2 with the fan kernel's own code:
 * out[k] = { y_new[3], f_new[3], error_norm, 0.9 err**-0.2, f[3] }. */
int pgr_debug_step(pgr_env* env, const double* t, const double* y, const double* h, int64_t M,
                   double rtol, double atol, double* out11);

/* Which instance of the fan kernel the LAST pgr_shoot_fan_device on `env` launched, and how (tests: the instance walk of
 * tests/test_hip_parity.py asserts that every instance it means to check is the one that ran):
 * out = { LDS_TAB, ZM, SAVE, PERSIST, grid (workgroups), threads per workgroup, dynamic LDS bytes, pre-assigned queue tail };
 * LDS_TAB: table in LDS (1) or HBM / L2 (0); ZM: depth look-up 0 division / binary search, 1 power-of-two dz, 2 bin table,
 * 3 quadratic estimate + three nodes, 4 dz = 1, 5 cubic estimate; SAVE: 0 end state, 1 trajectories (linspace grid, default sample
 * form), 2 any grid / PGR_EXACT_SAMPLES, 3 sample-blocked; PERSIST: persistent waves + packet queue.  -1s before any launch. */
int pgr_debug_last_instance(const pgr_env* env, int32_t out[8]);

/* What this build of the library is: whether the instruction-layout pass of the build was applied
 * ("relaid: 502 -> 31 straddles ..." or "plain hipcc") and which arithmetic variant was compiled.
 * bench.py puts it into its JSON line so that a measured number names the binary it came from. */
const char* pgr_build_info(void);

/* Message for the last error on the calling thread. */
const char* pgr_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* PGR_H */
