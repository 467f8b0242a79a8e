// pgr_launch.h -- launching a fan: per-launch wave scheduling (placement slots, pgr_wave_cost / pgr_wave_place) and pgr_shoot_fan_device --
// kernel-instance selection, LDS budget, the launch itself.
// (Part of the ONE translation unit pgr_hip.hip, included there in this order; not a stand-alone header.)
#ifndef PGR_LAUNCH_H
#define PGR_LAUNCH_H

// Builds the slot -> wave map for this launch on `st` (see pgr_wave_place); returns the map and
// the grid size through the references, or leaves map null when scheduling is off / not useful.
static int schedule_waves(pgr_env* env, const double* y0, int64_t N, int64_t waves, int W, hipStream_t st,
                          const int*& map_out, int64_t& blocks, int& slot_out, int*& queue_out, int& n_queue_out, int& n_tail_out,
                          bool persist_ok)
{
    n_tail_out = 0;
    map_out = nullptr;
    slot_out = -1;
    queue_out = nullptr;
    n_queue_out = 0;
    if (env->place == 0 || env->waves_per_block != 0 || W < 5 || waves > (1 << 27)) return 0;
    const int64_t cus = env->num_cus;
    int mode, B;
    if (waves <= (int64_t)W * cus && W <= 8 && waves > 4 * cus) {  // single round, 1-2 waves per SIMD
        mode = env->place;                                  // 1 or 2
        B = (mode == 1) ? (int)((waves + W - 1) / W) : (int)cus;
    } else if (waves > (int64_t)W * cus) {                  // several rounds
        mode = 3;
        B = (int)((waves + W - 1) / W);
        // persistent waves (default): one workgroup per CU, its waves claim the packets of the cost-sorted list one by
        // one (FanArgs::wave_queue); PGR_OPT_PERSISTENT 0 keeps the static deal of whole workgroups
        if (env->persistent && persist_ok) B = (int)cus;
    } else {
        return 0;
    }
    std::lock_guard<std::mutex> lock(env->place_mutex);  // host threads may share an env
    const bool persistent = (mode == 3) && env->persistent && persist_ok;
    size_t n_slots = persistent ? (size_t)waves : (size_t)B * W;
    // cost[waves] | the queue's counter (its own 256 bytes) | map[n_slots]
    size_t need = ((((size_t)waves * 4 + 255) & ~(size_t)255) + 256 + n_slots * 4 + 255) & ~(size_t)255;
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
    int* counter = (int*)(slot + (((size_t)waves * 4 + 255) & ~(size_t)255));
    int* map = counter + 64;
    HIPCHK(hipMemsetAsync(map, 0xFF, n_slots * sizeof(int), st));
    if (persistent) {
        HIPCHK(hipMemsetAsync(counter, 0, sizeof(int), st));
        queue_out = counter;
        n_queue_out = (int)waves;
        // Fans of up to two rounds (eight-wave workgroups): the first packets of waves 4 .. 7 come from the list's cheap
        // end, so that every steep packet starts beside a cheap one (140 000 rays: 6.6 instead of 7.9 ms, 200 000: 8.8 instead
        // of 9.6); beyond two rounds the steep packets are a small share of a long launch and the plain list is 1 - 3 %
        // faster (300 000 rays 11.9 against 12.2 ms, 1e6 34.3 against 34.5).  PGR_OPT_PERSISTENT 2 / 3: never / always.
        const bool tail_first = env->persistent == 3 || (env->persistent == 1 && waves <= 16 * (int64_t)B);
        if (tail_first && W == 8 && waves >= 8 * (int64_t)B) n_tail_out = 4 * B;
    }
    hipLaunchKernelGGL(pgr_wave_cost, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, y0, N, (int)waves, cost);
    hipLaunchKernelGGL(pgr_wave_place, dim3(1), dim3(1024), 0, st, cost, (int)waves, B, W, mode, map);
    map_out = map;
    blocks = B;
    return 0;
}

// Kernel variant of an environment: where the table lives (LDS copy of the single profile / HBM) and how a depth cell is
// found (zm 1 / 4: zin[j] = j dz exactly, 5: cubic index estimate, 3: quadratic estimate + three nodes, 2: bin table -- zin
// in LDS for those three --, 0: closed form for other uniform grids or binary search); zx_bytes = LDS the depth search takes.
static void select_variant(const pgr_env* env, bool& lds_tab, int& zm, size_t& zx_bytes)
{
    const EnvDev& D = env->d;
    const size_t tab_bytes = (size_t)D.nz * sizeof(double2);
    const size_t zb_bytes = D.z_bucket ? ((size_t)D.nz * sizeof(double) + (((size_t)D.zb_B * 2 + 15) & ~(size_t)15)) : 0;
    lds_tab = env->lds_path != 0;
    zm = D.z_simple ? ((D.dz == 1.0) ? 4 : 1) : 0;
    const size_t zq_bytes = (size_t)D.nz * sizeof(double);
    zx_bytes = 0;
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
}

// Would a trajectory fan of this environment run the sample-blocked kernel (PGR_SAMPLE_BLOCKED) if asked to?  The host-pointer
// entry and the fan handles ask before they size their device buffers: tables in HBM / L2 (the LDS-table kernels gain
// nothing from it), and room in the LDS for the staging of eight waves behind the depth search and the bathymetry.
static bool blocked_layout_fits(const pgr_env* env)
{
    bool lds_tab;
    int zm;
    size_t zx_bytes;
    select_variant(env, lds_tab, zm, zx_bytes);
    if (lds_tab || !env->api_blocked) return false;
    const size_t at = ((((zx_bytes + 15) & ~(size_t)15) + (size_t)env->d.nb * 16) + 15) & ~(size_t)15;
    return at + 8 * 6144 <= env->max_lds;
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
    // kernel variant: where the table lives and how a depth cell is found (select_variant)
    const EnvDev& D = env->d;
    const size_t tab_bytes = (size_t)D.nz * sizeof(double2);
    bool lds_tab;
    int zm;
    size_t zx_bytes;  // LDS taken by the depth search of the chosen variant
    select_variant(env, lds_tab, zm, zx_bytes);
    // PGR_SAMPLE_BLOCKED: everything that can be refused from the flags alone is refused HERE, before the scheduling below
    // claims a placement slot and queues its memset and two kernels on the caller's stream
    if (flags & PGR_SAMPLE_BLOCKED) {
        if (!save || !(flags & PGR_SAMPLE_MAJOR)) return fail("pgr_shoot_fan: PGR_SAMPLE_BLOCKED goes with trajectories and PGR_SAMPLE_MAJOR");
        if (lds_tab) return fail("pgr_shoot_fan: PGR_SAMPLE_BLOCKED is for environments whose tables stay in HBM (this one is on the LDS-table path)");
        if (!a.save_formula || (flags & PGR_EXACT_SAMPLES)) return fail("pgr_shoot_fan: PGR_SAMPLE_BLOCKED needs a linspace save grid (PGR_SAVE_LINSPACE) and the default sample form");
    }
    // SAVE of the kernel instance: 0 end state only, 1 trajectories on a linspace grid (default sample form), 2 any grid /
    // PGR_EXACT_SAMPLES, 3 = 1 in the sample-blocked layout; persistent waves are instantiated for 0, 1 and 3
    const int sv = !save ? 0 : (flags & PGR_SAMPLE_BLOCKED) ? 3 : (a.save_formula && !(flags & PGR_EXACT_SAMPLES)) ? 1 : 2;
    const bool persist_ok = (sv != 2);
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
            bool recorded;
            {
                std::lock_guard<std::mutex> lock(env->place_mutex);
                pgr_env::PlaceSlot& ps = env->place_slots[slot];   // (by index: the vector may have grown meanwhile)
                recorded = hipEventRecord(ps.ev, st) == hipSuccess;
                if (recorded) ps.recorded = true;
            }
            if (!recorded) {
                // no event to wait on: drain the stream WITHOUT the lock (other host threads keep launching on this
                // environment meanwhile; the slot stays claimed, so nobody takes it), then hand the slot back
                (void)hipStreamSynchronize(st);
                std::lock_guard<std::mutex> lock(env->place_mutex);
                env->place_slots[slot].in_flight = false;
            }
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
        if (schedule_waves(env, y0, N, waves, wpb, st, a.wave_map, blocks, place_slot, a.wave_queue, a.n_queue, a.n_queue_tail, persist_ok)) return -1;
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
            int* q = nullptr;
            int nq = 0, nt = 0;
            if (schedule_waves(env, y0, N, waves, W, st, m, nb2, place_slot, q, nq, nt, persist_ok)) return -1;
            if (m) { a.wave_map = m; blocks = nb2; wpb = W; a.wave_queue = q; a.n_queue = nq; a.n_queue_tail = nt; }
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
    if (flags & PGR_SAMPLE_BLOCKED) {   // (the flags themselves were checked before anything was queued; what is left is the LDS budget)
        const size_t at = (lds + 15) & ~(size_t)15, need = (size_t)wpb * 6144;
        if (at + need > env->max_lds) return fail("pgr_shoot_fan: no LDS left for PGR_SAMPLE_BLOCKED");
        a.blk_lds_off = (int)at; lds = at + need;
    }
    // n_queue_tail = 4 x grid is only right for eight-wave workgroups on a grid of exactly `blocks` workgroups (the kernel reads
    // tail entry n - 1 - (4 blockIdx + wave - 4) and never queues the tail): a change of either would skip or double-integrate packets
    if (a.n_queue_tail && !(threads == 512 && blocks * 4 == (int64_t)a.n_queue_tail))
        return fail("pgr_shoot_fan: internal error: the packet queue's pre-assigned tail does not match the launch shape");
#define PGR_LAUNCH2(LT, ZMV, SV, PV)                                                                 \
    do {                                                                                             \
        { const int li_[8] = {(int)(LT), (ZMV), (SV), (int)(PV), (int)blocks, threads, (int)lds, a.n_queue_tail};              \
          for (int q_ = 0; q_ < 8; q_++) env->last_instance[q_].store(li_[q_], std::memory_order_relaxed); } \
        if (lds > 64 * 1024)                                                                         \
            HIPCHK(hipFuncSetAttribute((const void*)pgr_fan_kernel<LT, ZMV, SV, PV>,                 \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));       \
        hipLaunchKernelGGL((pgr_fan_kernel<LT, ZMV, SV, PV>), dim3((unsigned)blocks), dim3(threads), lds, \
                           st, env->d_dev, a);                                                       \
    } while (0)
#define PGR_LAUNCH1(LT, ZMV, SV)                                                                     \
    do {                                                                                             \
        if (a.wave_queue && (SV) != 2) PGR_LAUNCH2(LT, ZMV, SV, ((SV) != 2));                        \
        else PGR_LAUNCH2(LT, ZMV, SV, false);                                                        \
    } while (0)
#define PGR_LAUNCH(LT, ZMV)                                                                          \
    do {                                                                                             \
        if (sv == 0) PGR_LAUNCH1(LT, ZMV, 0);                                                        \
        else if (sv == 1) PGR_LAUNCH1(LT, ZMV, 1);                                                   \
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
#undef PGR_LAUNCH2
    const hipError_t launch_err = hipGetLastError();
    // (the placement map is this launch's until its fan kernel has run: `guard` records the slot's event on `st` here
    // and on every error return between the slot's pick and this point)
    guard.release();
    if (launch_err != hipSuccess) return fail(std::string("fan kernel launch: ") + hipGetErrorString(launch_err));
    return 0;
}

extern "C" int pgr_debug_last_instance(const pgr_env* env, int32_t out[8])
{
    if (!env || !out) return fail("pgr_debug_last_instance: null argument");
    for (int q = 0; q < 8; q++) out[q] = env->last_instance[q].load(std::memory_order_relaxed);
    return 0;
}

namespace {
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 8) == hipSuccess ? 0 : -1; }
};
}  // namespace

#endif  // PGR_LAUNCH_H
