// pgr_fan_kernel.h -- the fan kernel: one ray per lane, lock-step step attempts, parked lanes and the
// bounce SERVICE phase (included by pgr_hip.hip only, after pgr_device.h).
#ifndef PGR_FAN_KERNEL_H
#define PGR_FAN_KERNEL_H

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
// waited `park_trips` trips, or nobody else can step); the service forms the step's dense output from
// the stage values the attempt left in the lane's registers (a parked lane takes no further attempt;
// the ZM = 5 trajectory instances replay the stages instead).  Per-ray arithmetic is unchanged by when the service runs.
//
// Control flow inside a trip is kept free of skipped blocks (a taken skip-branch costs a lone
// wave ~80 cycles): selects where both sides are cheap, ONE block per kind of rare work, and
// the memory half of each table look-up issued early with independent work behind it.
// ------------------------------------------------------------------------------------
// SAVE: 0 = end state only (no sample code); 1 = trajectories on a grid that IS np.linspace (recomputed
// per index, no loads) with the default sample evaluation; 2 = any grid / PGR_EXACT_SAMPLES; 3 = as 1, written in the
// sample-blocked layout [ceil(S/4)][N][4] (PGR_SAMPLE_BLOCKED; instantiated for the HBM-table kernels only)
// PERSIST: persistent waves claiming 64-ray packets from the cost-sorted list (fans of several rounds; see the packet loop
// below); false: one packet per wave, and the loop folds away -- the instances the 1e5-ray fans run are the code they were
template <bool LDS_TAB, int ZM, int SAVE, bool PERSIST>
__global__ void __launch_bounds__(512)
pgr_fan_kernel(const EnvDev* __restrict__ env_p, FanArgs a)
{
    // the environment descriptor lives in device memory: its ~50 dwords would otherwise occupy
    // half the wave's SGPRs as kernel arguments and push the Runge-Kutta tableau (60 fp64
    // literals) into constant re-materialisation + SGPR spills inside the step loop
    const EnvDev& env = *env_p;
    extern __shared__ double2 lds_tab[];
    // LDS layout: [{c, cp}[nz] when LDS_TAB][zin[nz] when ZM is 2, 3 or 5, zbucket[zb_B] when ZM == 2][bathymetry]
    double* const lds_after_tab = (double*)(lds_tab + (LDS_TAB ? env.nz : 0));
    double* const lds_z = lds_after_tab;
    unsigned short* const lds_zb = (unsigned short*)(lds_z + env.nz);
    if (LDS_TAB) {
        // stage the single depth profile {c, cp}[nz] into LDS (coalesced 16 B per lane)
        for (int j = threadIdx.x; j < env.nz; j += blockDim.x) lds_tab[j] = env.tab[j];
    }
    if (ZM == 2 || ZM == 3 || ZM == 5) {
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
    const int wv = threadIdx.x >> 6;
    const int n_cw = (int)(blockDim.x >> 6);   // waves of this workgroup
    __syncthreads();
    const Ctx<LDS_TAB, ZM> C(env, lds_tab, lds_z, lds_zb, lds_bx);
    C.declare_span(a.x0, a.x1);
    // PERSISTENT WAVES (a.wave_queue, fans of several rounds): the grid is one workgroup per CU, the table is staged into
    // the LDS once, and every wave of it integrates one 64-ray PACKET after the other -- the next entry of the
    // cost-sorted list pgr_wave_place wrote (most expensive first), claimed with one atomic -- until the list is
    // empty.  A workgroup with the 96 KB table holds its CU until its LAST wave ends; dealt whole workgroups of
    // packets (the static mode 3) the SIMD slots of the waves that end early idle until then, and the chip drains
    // workgroup by workgroup at the end of the launch.  Which wave integrates a packet, and when, never changes what it
    // computes.  !PERSIST: the loop body runs once.
    bool first_packet = true;
    for (;;) {
    // PERSIST: what is read at the start of a packet comes from the kernel-argument segment through a pointer the
    // compiler cannot trace back (as in the service phase and the epilogue): as ordinary arguments these values would
    // be live across the step loop of the packet before
    const char __attribute__((address_space(4))) * kq_p =
        (const char __attribute__((address_space(4))) *)__builtin_amdgcn_kernarg_segment_ptr();
    if (PERSIST) asm volatile("" : "+s"(kq_p));
    const FanArgs __attribute__((address_space(4))) & aq = *(const FanArgs __attribute__((address_space(4))) *)(kq_p + kFanArgsKernargOffset);
    int* const wave_queue = PERSIST ? aq.wave_queue : nullptr;
    const int* const q_map = PERSIST ? aq.wave_map : a.wave_map;
    const int64_t q_N = PERSIST ? aq.N : a.N, q_stride_ray = PERSIST ? aq.stride_ray : a.stride_ray;
    const double* const q_y0 = PERSIST ? aq.y0 : a.y0;
    const uint32_t q_flags = PERSIST ? aq.flags : a.flags;
    // waves are dealt to workgroups round-robin (wave w of block b = global wave w*grid + b):
    // neighbouring launch angles cost alike, so a strided deal balances the CUs
    int64_t gwave = (int64_t)wv * gridDim.x + blockIdx.x;
    if (q_map) {
        // cost-aware scheduling (pgr_wave_place): slot -> wave (-1 = slot left empty) and the
        // wave's issue priority in bits 28..29: the costlier wave of a SIMD's pair runs at its own
        // pace, the cheaper one fills the issue slots it leaves
        int slot = blockIdx.x * n_cw + wv;
        if (PERSIST && wave_queue) {
            // A SIMD's two resident waves are workgroup waves k and k + 4.  Straight down the list both would start on
            // packets of the list's expensive head -- two steep packets on one SIMD take the sum of their lone times,
            // and a fan of two or three rounds lasts as long as that pair (140 000 rays: 8.2 ms for 1.4 x the work of
            // the 4.9 ms 1e5-ray fan).  So the FIRST packet of waves 4 .. 7 comes from the list's cheap END (entry
            // n - 1 - (4 b + k - 4), no atomic), everything else from its head: each steep packet starts beside a cheap
            // one and, with the higher issue priority, runs at nearly its lone pace.  (n_queue_tail = 4 x grid, or 0.)
            const int n_tail = aq.n_queue_tail, n_head = aq.n_queue - n_tail;
            if (first_packet && wv >= 4 && n_tail) {
                slot = aq.n_queue - 1 - ((int)blockIdx.x * 4 + (wv - 4));
            } else {
                int got = 0;
                if ((threadIdx.x & 63) == 0) got = atomicAdd(wave_queue, 1);
                slot = __builtin_amdgcn_readfirstlane(got);
                if (slot >= n_head) break;      // (wave-uniform: the list is empty, this wave is done)
            }
            first_packet = false;
        }
        int m = q_map[slot];
        gwave = (m < 0) ? -1 : (m & 0x0fffffff);
        int prio = __builtin_amdgcn_readfirstlane((m < 0) ? 0 : ((m >> 28) & 3));
        if (prio == 3) __builtin_amdgcn_s_setprio(3);
        else if (prio == 2) __builtin_amdgcn_s_setprio(2);
        else if (prio == 1) __builtin_amdgcn_s_setprio(1);
        else if (PERSIST) __builtin_amdgcn_s_setprio(0);
    }
    if (PERSIST) C.reset_range_cache();   // (a new packet starts at x0 again)
#ifdef PGR_WAVE_TIMES   // (diagnostic build, scripts/wave_times.py: when and where every packet ran -- 100 MHz s_memrealtime stamps)
    const unsigned wt_start = (unsigned)__builtin_amdgcn_s_memrealtime();
#endif
    const int64_t ray = gwave * 64 + (threadIdx.x & 63);
    const bool valid = (gwave >= 0) && (ray < q_N);
    double SAFETY = 0.9, MIN_FACTOR = 0.2, MAX_FACTOR = 10,
           SQRT3 = 1.7320508075688772, INV_SQRT3 = 0.57735026918962584,
           POW_FIFTH = 0.2, POW_KLN2 = PGR_CR_POW_KLN2;
    const double rtol = a.rtol, atol = a.atol, t_bound = a.x1;
    const int S = a.S;
    constexpr bool save = (SAVE != 0);  // trajectories wanted (a.T != nullptr)
    const bool exact_samples = (SAVE == 2) && (a.flags & PGR_EXACT_SAMPLES) != 0;
    // (max_steps <= 2^30, host checked; the counters are ints)
    const int attempt_limit = (int)((4 * a.max_steps + 4096 < 0x3fffffff) ? 4 * a.max_steps + 4096 : 0x3fffffff);
    // PGR_STORED_SIGN: trajectories leave as pygenray stores them, z -> -z and p -> -p
    // (REF/ray_objects.py:51-52): a sign-bit xor, exact, and two host passes over 1.6 GB less
    const unsigned long long sgn = (a.flags & PGR_STORED_SIGN) ? 0x8000000000000000ULL : 0ULL;
#define SGN(v) __longlong_as_double(__double_as_longlong(v) ^ (long long)sgn)
    SaveGrid G;
    G.r = a.r_save; G.x0 = a.x0; G.x1 = a.x1; G.step = a.save_step;
    G.S = S; G.formula = (SAVE == 1 || SAVE == 3) ? 1 : a.save_formula;

    double t = a.x0, y0 = 0, y1 = 0, y2 = 0;
    if (valid) {
        y0 = q_y0[3 * ray + 0];
        y1 = q_y0[3 * ray + 1];
        y2 = q_y0[3 * ray + 2];
    }
    double f0 = 0, f1 = 0, f2 = 0, h_abs = 0;
    unsigned g = 0;
    int status = valid ? RUNNING : PGR_RAY_OK;
    if (valid && (q_flags & PGR_SKIP_NAN_Y0) && (y2 != y2)) status = PGR_RAY_SKIPPED;  // a parked eigenray bracket
    bool need_init = true, rejected = false, parked = false;
    int nb = 0, ns = 0, n_steps = 0, n_rej = 0;
    int jnext = 0;
    double rnext = 0;
    // a parked lane keeps the end of its step, which events fired and -- where they are, in K3 ... K7 -- its stage values
    double pk_tnew = 0;
    unsigned pk_active = 0;
    int waited = 0;
    // trips: the wave's trip count -- FUNCTIONAL: it gates the two step-count guards (no lane has made more attempts
    // than the wave has made trips) -- and, with services / fallbacks, a PGR_DEBUG_TRIPS diagnostic
    int trips = 0, services = 0, fallbacks = 0;
    const double min_step_bound = 10 * 0x1p-52 * fmax(fabs(a.x0), fabs(a.x1)) + 1e-300;  // see the attempt's head
    const int max_steps32 = (int)(a.max_steps < 0x7fffffff ? a.max_steps : 0x7fffffff);  // n_steps is an int
    const int guard_limit = (max_steps32 < attempt_limit ? max_steps32 : attempt_limit);
#if defined(PGR_TIMING) || defined(PGR_SVC_TIMING)
    unsigned tacc[24];
    for (int k = 0; k < 24; k++) tacc[k] = 0;
    unsigned tprev = (unsigned)clock64();
#endif
    const int64_t out_off = ray * q_stride_ray;
#define Tp (a.T + out_off)
#define Zp (a.Z + out_off)
#define Pp (a.P + out_off)
    // one saved sample (T, z, p) of row j leaves the lane: streaming stores when the table lives in HBM / L2 (2.4 GB
    // of samples per fan must not evict the table rows), plain ones with the LDS table (measured faster there).
    // (Built, measured and not kept -- an LDS sample ring with a writer wave, a per-wave row ring, deferred stores:
    // scripts/experiments/r03_sample_store_experiments.patch.)
    // SAVE == 3, PGR_SAMPLE_BLOCKED: a lane stages its samples 4b ... 4b + 3 in its OWN LDS slots (no cross-lane
    // protocol: [T, z, p][slot][lane], 6 KB per wave) and stores them itself when the fourth arrives, as one full 32-byte
    // piece per array into [S/4][N][4] -- half the store instructions of the row layout, no line written in pieces by
    // different trips.  A re-sample after a bounce rewrites at most sample jnext - 1 (REF/launch_rays.py:766-772): the
    // other slots still hold that block's samples (nothing of the next block has been staged yet), so the block is
    // simply stored again.
    // (Tried in the LDS-table kernels too, round 4: headline fan 5.86 against 5.70 ms, its lone steepest wave 5.40 against
    // 5.27 ms -- there the plain stores cost nothing to issue and the staging is pure overhead.  HBM-table instances only.)
    constexpr bool BLK = (SAVE == 3);
    static_assert(!(BLK && LDS_TAB), "the blocked sample layout is instantiated for the HBM-table kernels only");
    double* const blk = (double*)((char*)lds_tab + (BLK ? a.blk_lds_off : 0)) + (wv * 768 + (int)(threadIdx.x & 63));
    auto blk_flush = [&](int jb) __attribute__((always_inline)) {
        const int64_t o = ((int64_t)jb * a.N + ray) * 4;
        typedef double d2 __attribute__((ext_vector_type(2)));
        // (slots s and s + 1 of one array are 512 B apart: one ds_read2st64_b64 each, straight into a store's four registers)
        const d2 t0 = {blk[0], blk[64]}, t1 = {blk[128], blk[192]};
        const d2 z0 = {blk[256], blk[320]}, z1 = {blk[384], blk[448]};
        const d2 p0 = {blk[512], blk[576]}, p1 = {blk[640], blk[704]};
        __builtin_nontemporal_store(t0, (d2*)(a.T + o)); __builtin_nontemporal_store(t1, (d2*)(a.T + o + 2));
        __builtin_nontemporal_store(z0, (d2*)(a.Z + o)); __builtin_nontemporal_store(z1, (d2*)(a.Z + o + 2));
        __builtin_nontemporal_store(p0, (d2*)(a.P + o)); __builtin_nontemporal_store(p1, (d2*)(a.P + o + 2));
    };
    auto emit_sample = [&](int j, double vt, double vz, double vp) __attribute__((always_inline)) {
        if (BLK) {
            double* const e = blk + (j & 3) * 64;
            e[0] = vt; e[256] = vz; e[512] = vp;
            if ((j & 3) == 3) blk_flush(j >> 2);
            return;
        }
        const int64_t o = (int64_t)j * a.stride_smp;
        if (LDS_TAB) { Tp[o] = vt; Zp[o] = vz; Pp[o] = vp; }
        else {
            __builtin_nontemporal_store(vt, &Tp[o]);
            __builtin_nontemporal_store(vz, &Zp[o]);
            __builtin_nontemporal_store(vp, &Pp[o]);
        }
    };

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
    // ZM == 5: a Horner step whose multiplier AND addend are wave-uniform costs a v_mov (one scalar operand per
    // VALU instruction): the addend of the first step of the index cubic and of the seed quadratic sit in VGPRs
    // (6 look-ups per attempt use them)
#ifndef PGR_PIN_ZC   // (A/B, flat-earth fan: all seven in VGPRs 6.86 vs 7.02 ms with trajectories, 5.78 vs 5.75 ms without)
#define PGR_PIN_ZC (SAVE != 0 ? 2 : 1)
#endif
    if (ZM == 5 && (PGR_PIN_ZC) >= 1) {
        asm volatile("" : "+v"(C.h_zc_g2));
        asm volatile("" : "+v"(C.h_zc_s1));
    }
    if (ZM == 5 && (PGR_PIN_ZC) >= 2) {   // all seven: the scalar file of the trajectory kernels is full of loop values already
        asm volatile("" : "+v"(C.h_zc_g0));
        asm volatile("" : "+v"(C.h_zc_g1));
        asm volatile("" : "+v"(C.h_zc_g3));
        asm volatile("" : "+v"(C.h_zc_s0));
        asm volatile("" : "+v"(C.h_zc_s2));
    }
    // K3 ... K7 of a lane's latest attempt live across the trips: a lane that parks takes no further attempt (its lanes of
    // these registers are not written again), so the service finds the parked step's stage values where the attempt left
    // them instead of replaying the attempt's six right-hand sides (2.2 k of a service's 21.8 k cycles; the kernel is 1 350
    // instructions shorter).  Costs no register: the values are live to the end of an attempt anyway, and the early
    // stages of the next one have room (same VGPR count, same spill count, four v_readlane fewer in the step loop) -- except
    // in the ZM = 5 trajectory instances, whose look-up coefficients fill the file (12 -> 28 spills): those keep the replay.
#ifndef PGR_KEEP_K
#define PGR_KEEP_K (!(ZM == 5 && SAVE != 0))
#endif
    constexpr bool KEEPK = PGR_KEEP_K;
    double k30 = 0, k31 = 0, k32 = 0, k40 = 0, k41 = 0, k42 = 0, k50 = 0, k51 = 0, k52 = 0, k60 = 0, k61 = 0, k62 = 0,
           k70 = 0, k71 = 0, k72 = 0;
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
            // min_step = 10 ulp(t) <= 10 * 2^-52 max(|x0|, |x1|) =: min_step_bound for every t of the fan: a wave none
            // of whose lanes steps below that bound needs neither the clamp nor the "step too small" test (exactly:
            // h_abs >= bound >= min_step makes both no-ops) -- ten instructions of every attempt otherwise
            bool too_small = false;
            if (__builtin_expect(ballot64(h_abs < min_step_bound) != 0, 0)) {
                double min_step = min_step_of(t);
                if (!rejected && h_abs < min_step) h_abs = min_step;  // clamp only on entry
                too_small = h_abs < min_step;
                // (RK45 raises "step size too small": the ray is dropped here and now; what the rest of this attempt
                // does to the lane's state no longer matters, it is not accepted and never steps again)
                if (too_small) status = PGR_RAY_STEP_TOO_SMALL;
            }
            double h = h_abs;
            double t_new = t + h;
            if ((t_new - t_bound) > 0) t_new = t_bound;
            h = t_new - t;
            h_abs = fabs(h);

            PGR_RK_STAGES_BODY(t, h);
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
            h_abs = h_abs * (accepted ? fac_acc : fac_rej);
            rejected = reject;
            n_rej += reject ? 1 : 0;
            // the two step-count guards: no lane has made more attempts than the wave has made trips
            const bool guards_due = trips > guard_limit;
            if (__builtin_expect(guards_due, 0)) {
                const bool over = (n_rej + n_steps) > attempt_limit;
                status = (reject & over) ? PGR_RAY_MAX_STEPS : status;
                trips = trips < 0x7f000000 ? trips : 0x7f000000;   // (entered on every trip from here on: the counter cannot wrap)
            }
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
                    // park: the step is located, truncated and bounced in the next service phase, from (t, y, f),
                    // pk_tnew and this attempt's stage values (K3 ... K7 stay in this lane's registers: KEEPK)
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
                                emit_sample(jnext, o0, SGN(o1), SGN(o2));
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
#define PGR_SAMPLE_LOOP(NEXT)                                                                     \
    while (jnext < S - 1 && rnext <= t_new) {                                                     \
        const double xi = (rnext - t) * inv_h, x2 = xi * xi;                                      \
        const double b1 = xi * __builtin_fma(xi, __builtin_fma(xi, __builtin_fma(xi, vP13, vP12), vP11), P10); \
        const double b3 = x2 * __builtin_fma(xi, __builtin_fma(xi, vP33, vP32), vP31);               \
        const double b4 = x2 * __builtin_fma(xi, __builtin_fma(xi, vP43, vP42), vP41);               \
        const double b5 = x2 * __builtin_fma(xi, __builtin_fma(xi, vP53, vP52), vP51);               \
        const double b6 = x2 * __builtin_fma(xi, __builtin_fma(xi, vP63, vP62), vP61);               \
        const double b7 = x2 * __builtin_fma(xi, __builtin_fma(xi, vP73, vP72), vP71);               \
        emit_sample(jnext, __builtin_fma(h, PGR_KSUM(f0, k30, k40, k50, k60, k70), y0),            \
                    SGN(__builtin_fma(h, PGR_KSUM(f1, k31, k41, k51, k61, k71), y1)),               \
                    SGN(__builtin_fma(h, PGR_KSUM(f2, k32, k42, k52, k62, k72), y2)));              \
        jnext++;                                                                                  \
        rnext = NEXT;                                                                             \
    }
                            // two copies so that the linspace one holds no load: a load in the loop
                            // makes every iteration wait (vmcnt) for the stores of the one before
                            // (every use of rnext is guarded by jnext < S - 1, so the formula copy needs no
                            // select for the forced last grid point)
                            if (SAVE == 1 || SAVE == 3 || G.formula) { PGR_SAMPLE_LOOP(grid_at(G.x0, G.step, jnext)) }
                            else { PGR_SAMPLE_LOOP(G.r[jnext]) }
#undef PGR_SAMPLE_LOOP
#undef PGR_KSUM
                        }
                    }
                    t = t_new; y0 = n0; y1 = n1; y2 = n2;
                    f0 = k70; f1 = k71; f2 = k72;
                    // (selects, not a skipped block: a taken branch costs more than these four instructions)
                    status = ((t - t_bound) >= 0) ? PGR_RAY_OK : status;  // SCIPY/base.py:197
                    if (__builtin_expect(guards_due, 0))
                        status = (status == RUNNING && n_steps > max_steps32) ? PGR_RAY_MAX_STEPS : status;
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
                PGR_SSTAMP(23);   // (everything since the last service)
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
                const FanArgs __attribute__((address_space(4))) & as = *(const FanArgs __attribute__((address_space(4))) *)(ks_p + kFanArgsKernargOffset);
                const double svc_c_lo = es.c_lo, svc_c_hi = es.c_hi;
                if (pend && parked) {
                    parked = false;
                    const unsigned active = pk_active;
                    // the parked attempt: same t, y, f and h = t_new - t as when it ran
                    const double t_new = pk_tnew, h = t_new - t;
                    Dense D;
                    PGR_SSTAMP(0);    // descriptor / argument loads, entry
                    if (KEEPK) {
                        PGR_FORM_Q();          // from the stage values the parked attempt left in this lane's registers
                    } else {
                        // replay the parked attempt: same t, y, f and h = t_new - t as when it ran (the names shadow the outer ones)
                        PGR_RK_STAGES(t, h);
                        (void)n0; (void)c_new; (void)es0; (void)es1; (void)es2;
                        PGR_FORM_Q();
                    }
                    PGR_SSTAMP(1);    // Q
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
                        PGR_SSTAMP(2);    // bracket values, bathymetry at the step's ends
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
                            PGR_SSTAMP(3);    // Newton
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
                            PGR_TRUE_EVENT(xa, ga, bbox_a);
                            PGR_TRUE_EVENT(xb, gb, bbox_b);
                            (void)bbox_a;
                            // with the bounding-box event also active its flip must lie beyond xb, so
                            // that the surface root is the earlier one (SCIPY/ivp.py:100-131)
                            live = (xa < xb) && !ga && gb && !(with_bbox && bbox_b);
                            PGR_SSTAMP(4);    // band + the true event at its two edges
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
                            PGR_SSTAMP(5);    // the doubles inside a narrow band
                        }
#ifdef PGR_NO_REPLAY  // experiments: round 1's "a root within brentq's tolerance" (NOT bit-identical)
                        if (live) { best = xb; ev = bottom ? 1 : 0; }
#else
                        // ---- the replay.  brentq's state: cur = the latest iterate, xblk = the other end of
                        // the bracket, fcur = the event at cur; it starts from cur = t_new (fired), xblk = t.
                        const double xtol = 4 * DBL_EPSILON, brtol = 4 * DBL_EPSILON;
                        double cur = t_new, xblk = t;
                        bool fcur = true;
                        {
                            // phase 1: the halvings that can neither end the search nor take brentq's minimum
                            // step (|xblk - cur| / 2 stays above 4 delta), kept as (not fired end, fired end): 9
                            // instructions each, no tolerance arithmetic.  cur + (xblk - cur) / 2 and lo + (hi -
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
                            if (n1 > 0) { cur = lastc ? phi : plo; xblk = lastc ? plo : phi; fcur = lastc; }
                        }
                        PGR_SSTAMP(6);    // replay phase 1
                        // phase 2: brentq's loop as it stands (scipy/optimize/Zeros/brentq.c) for the last few
                        // iterations, all lanes in lock step; the true event is evaluated (for the whole wave,
                        // behind a uniform branch) whenever some lane's iterate lies inside the band, and
                        // decides for those lanes.  A lane whose search has ended (|sbis| < delta) stands still.
                        double fcv = fcur ? 1.0 : 0.0;  // the event at cur, as a number (a carried bool costs more)
                        bool done = !live;
                        for (int it = 0; it < 200; it++) {
                            const double dlt = (xtol + brtol * fabs(cur)) / 2;
                            const double sbis = (xblk - cur) / 2;
                            done = !live | (fabs(sbis) < dlt);
                            if (ballot64(!done) == 0) break;
                            const double nw = (fabs(sbis) > dlt) ? cur + sbis : cur + (sbis > 0 ? dlt : -dlt);
                            const bool inside = !done & (nw > xa) & (nw < xb);
                            double fnv = (nw >= xb) ? 1.0 : 0.0;
                            if (ballot64(inside) != 0) {
                                bool fired, bbox_q;
                                PGR_TRUE_EVENT(nw, fired, bbox_q);
                                (void)bbox_q;
                                fnv = inside ? (fired ? 1.0 : 0.0) : fnv;
                            }
                            xblk = (!done & (fnv != fcv)) ? cur : xblk;
                            cur = done ? cur : nw;
                            fcv = done ? fcv : fnv;
                        }
                        if (live && done) { best = cur; ev = bottom ? 1 : 0; }
                        PGR_SSTAMP(7);    // replay phase 2
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
                    PGR_SSTAMP(8);    // (exact bisection, if any)
                    if (status == RUNNING) {
                        const double t_end = best;
                        // samples of this (truncated) step, REF/launch_rays.py:763-772 (Q5)
                        if (save) {
                            while (jnext < S - 1 && rnext <= t_end) {
                                double o0, o1, o2;
                                D.eval(t, y0, y1, y2, rnext, o0, o1, o2);
                                emit_sample(jnext, o0, SGN(o1), SGN(o2));
                                jnext++;
                                rnext = G.at(jnext);
                            }
                        }
                        // terminal event: t = root, y = sol(root) (SCIPY/ivp.py:689-692), then the
                        // bounce logic of REF/launch_rays.py:432-480
                        PGR_SSTAMP(9);    // samples of the truncated step
                        double r0, r1, r2;
                        D.eval(t, y0, y1, y2, t_end, r0, r1, r2);
                        t = t_end; y0 = r0; y1 = r1; y2 = r2;
                        PGR_SSTAMP(10);   // state at the root
                        if (ev == 2) status = PGR_RAY_VERTICAL;
                        else if (ev == 3) status = PGR_RAY_BBOX;
                        else {
                            double c, cp;
                            C.lookup(t, y1, c, cp);
                            const double pc_b = y2 * c;
                            const PGR_ASIN_DD_T A_b = PGR_ASIN_DD(pc_b);
                            double theta = PGR_ASIN_DD_HI(A_b) * (180.0 / M_PI);  // ray_angle
                            double theta_b;
                            PGR_SSTAMP(11);   // look-up + arcsine
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
                            PGR_SSTAMP(12);   // reflection law (bottom: the angle's cubic)
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
                PGR_SSTAMP(13);   // new slowness (sine), end of the bounce block
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
                    PGR_SSTAMP(14);   // restart: first right-hand side, d0, d1, h0
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
                    PGR_SSTAMP(15);   // restart: second right-hand side, d2, the power
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
                    PGR_SSTAMP(16);   // restart: events, nearest save point
                }
                PGR_SSTAMP(17);
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
        const FanArgs __attribute__((address_space(4))) & a = *(KArgs)(kp + kFanArgsKernargOffset);  // (shadows the argument)
        if (BLK) {
            if (ok) {
                // last column = exact final state (REF/launch_rays.py:775-777): into its slot, and the last block goes out;
                // the slots behind it (an earlier block's samples) become NaN first, so that the padding rows S ... 4 ceil(S/4) - 1
                // hold NaN for every ray -- a device-resident consumer may reduce over the whole padded buffer
                const int last = (S - 1) & 3;
                double* const e = blk + last * 64;
                e[0] = y0; e[256] = SGN(y1); e[512] = SGN(y2);
                for (int k = last + 1; k < 4; k++) { blk[k * 64] = nan; blk[256 + k * 64] = nan; blk[512 + k * 64] = nan; }
                blk_flush((S - 1) >> 2);
            } else {
                for (int k = 0; k < 12; k++) blk[k * 64] = nan;
                for (int jb = 0; jb < (S + 3) / 4; jb++) blk_flush(jb);
            }
        } else if (save) {
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
#if defined(PGR_TIMING) || defined(PGR_SVC_TIMING)
        if ((threadIdx.x & 63) < 24) {
            unsigned v = 0;
            for (int k = 0; k < 24; k++) v = ((threadIdx.x & 63) == k) ? tacc[k] : v;
            n_rej = (int)v;
        }
#endif
#ifdef PGR_WAVE_TIMES
        if (a.flags & PGR_DEBUG_TRIPS) {
            const int ln = (int)(threadIdx.x & 63);
            const unsigned wt_end = (unsigned)__builtin_amdgcn_s_memrealtime();
            const unsigned hw_id = __builtin_amdgcn_s_getreg((4 /* HW_REG_HW_ID */) | (0 << 6) | (31 << 11));
            const unsigned xcc_id = __builtin_amdgcn_s_getreg((20 /* HW_REG_XCC_ID */) | (0 << 6) | (31 << 11));
            fallbacks = ln == 3 ? (int)wt_start : ln == 4 ? (int)wt_end : ln == 5 ? (int)hw_id : ln == 6 ? (int)xcc_id
                        : ln == 7 ? (int)blockIdx.x : fallbacks;
        }
#endif
        if (a.n_rej) a.n_rej[ray] = (a.flags & PGR_DEBUG_TRIPS) ? (((threadIdx.x & 63) == 0) ? trips : (((threadIdx.x & 63) == 1) ? services : fallbacks)) : n_rej;
    }
    if (!PERSIST || !wave_queue) break;   // (a persistent instance launched without a queue is one packet per wave too)
    }   // (the packet loop)
#undef Tp
#undef Zp
#undef Pp
#undef SGN
}

#endif  // PGR_FAN_KERNEL_H
