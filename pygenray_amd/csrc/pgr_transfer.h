// pgr_transfer.h -- results to the host: compaction of dropped rays on the device, the D2H copy pipelined with the page faults of the
// caller's buffers, and the host-pointer entry pgr_shoot_fan built on both.
// (Part of the ONE translation unit pgr_hip.hip, included there in this order; not a stand-alone header.)
#ifndef PGR_TRANSFER_H
#define PGR_TRANSFER_H

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

// The same squeeze out of a sample-BLOCKED array [ceil(S/4)][N][4] (PGR_SAMPLE_BLOCKED): dst[4 b + q][m] = src[b][idx[m]][q].
// One thread per (block b, surviving ray m): it reads the ray's four samples as one 32-byte piece (adjacent rays: adjacent
// pieces) and writes them into four rows (a wave: four 512-byte runs).  idx == nullptr: every ray is kept (m = k).  This
// is how the host-pointer entry and the fan handles hand API callers of HBM-table environments plain [S][M] arrays while
// the fan kernel writes full 32-byte pieces (REF/launch_rays.py:166-186 is what the caller gets either way).
__global__ void pgr_unblock_cols(const double* __restrict__ src, double* __restrict__ dst,
                                 const int* __restrict__ idx, int64_t M, int64_t N, int S)
{
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    const int64_t b = blockIdx.y;
    const int64_t k = idx ? (int64_t)idx[m] : m;
    typedef double d2 __attribute__((ext_vector_type(2)));
    const d2* p = (const d2*)(src + (b * N + k) * 4);
    const d2 lo = __builtin_nontemporal_load(p), hi = __builtin_nontemporal_load(p + 1);
    const int64_t j = 4 * b;
    dst[j * M + m] = lo.x;
    if (j + 1 < S) dst[(j + 1) * M + m] = lo.y;
    if (j + 2 < S) dst[(j + 2) * M + m] = hi.x;
    if (j + 3 < S) dst[(j + 3) * M + m] = hi.y;
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
    // is r_save exactly np.linspace(source_range, receiver_range, S)?  then the kernel recomputes it per index instead
    // of loading it
    bool lin = save;
    if (save) {
        const double step = (S > 1) ? (receiver_range - source_range) / (double)(S - 1) : 0.0;
        for (int32_t j = 0; j < S && lin; j++) {
            volatile double m = (double)j * step;
            volatile double v = m + source_range;
            double want = (j == S - 1 && S > 1) ? receiver_range : (double)v;
            lin = (r_save[j] == want);
        }
        if (lin) flags |= PGR_SAVE_LINSPACE; else flags &= ~PGR_SAVE_LINSPACE;
    }
    // Environments whose tables stay in HBM / L2: the trajectories are integrated by the sample-blocked kernel (device
    // buffers [ceil(S/4)][N][4]) and un-blocked to the caller's [S][N] / [S][M] by the pass that squeezes dropped rays out
    bool blocked = save && lin && (flags & PGR_SAMPLE_MAJOR) && !(flags & PGR_EXACT_SAMPLES) && blocked_layout_fits(env) &&
                   N <= 0x7fffffff;
    if (blocked) {
        // ... which needs a second workspace of 3 S N doubles beside the blocked buffers.  It is obtained HERE, before
        // anything is launched: if the device cannot give it, the fan runs the plain row kernel straight into the
        // caller's layout instead (same bits, 2.3 x the store traffic) -- the blocked path never fails a call that the row
        // path would have served
        const size_t piece = ((size_t)S * (size_t)N * sizeof(double) + 255) & ~(size_t)255;
        const size_t need2 = 3 * piece + (((size_t)N * 4 + 255) & ~(size_t)255) + 256;
        if (need2 > env->ws2_bytes) {
            if (env->ws2) (void)hipFree(env->ws2);
            env->ws2 = nullptr; env->ws2_bytes = 0;
            if (hipMalloc(&env->ws2, need2) == hipSuccess) env->ws2_bytes = need2;
            else { env->ws2 = nullptr; (void)hipGetLastError(); blocked = false; }
        }
    }
    const size_t dev_ns_bytes = blocked ? (size_t)N * (size_t)(4 * ((S + 3) / 4)) * sizeof(double) : ns_bytes;
    // carve one workspace: y0, r_save, T, Z, P, end, 5 int arrays (256-byte aligned pieces)
    const size_t sizes[11] = {(size_t)N * 24, (size_t)(save ? S : 1) * 8, dev_ns_bytes, dev_ns_bytes, dev_ns_bytes,
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
    int rc = pgr_shoot_fan_device(env, (const double*)dy0.p, N, source_range, receiver_range,
                                  (const double*)dr.p, S, rtol, atol, flags | (blocked ? PGR_SAMPLE_BLOCKED : 0u), max_steps,
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
        const bool compact = save && (flags & PGR_COMPACT);
        if (!compact && !blocked) return 0;
        if (compact && !(flags & PGR_SAMPLE_MAJOR)) return fail("pgr_shoot_fan: PGR_COMPACT needs PGR_SAMPLE_MAJOR");
        if (N > 0x7fffffff) return fail("pgr_shoot_fan: PGR_COMPACT supports at most 2^31 - 1 rays per call");   // (compact only: the blocked path was decided with N in range)
        int64_t M = N;
        if (compact) {
            keep.reserve((size_t)N);
            for (int64_t k = 0; k < N; k++) if (status[k] == 0) keep.push_back((int)k);
            M = (int64_t)keep.size();
        }
        if (M == N && !blocked) return 0;
        const size_t mbytes = (size_t)S * (size_t)M * sizeof(double), piece = (mbytes + 255) & ~(size_t)255;
        const size_t need2 = 3 * piece + (((size_t)M * 4 + 255) & ~(size_t)255) + 256;
        if (need2 > env->ws2_bytes) {
            if (env->ws2) (void)hipFree(env->ws2);
            env->ws2 = nullptr; env->ws2_bytes = 0;
            if (hipMalloc(&env->ws2, need2) != hipSuccess) { env->ws2 = nullptr; return fail("pgr_shoot_fan: device allocation of the PGR_COMPACT workspace failed"); }
            env->ws2_bytes = need2;
        }
        int* didx = (int*)((char*)env->ws2 + 3 * piece);
        if (M > 0) {
            const bool all = (M == N);     // (blocked and nothing dropped, or no compaction asked for: un-block every ray)
            if (!all) HIPCHK(hipMemcpyAsync(didx, keep.data(), (size_t)M * sizeof(int), hipMemcpyHostToDevice, st));
            for (int a3 = 0; a3 < 3; a3++) {
                double* tmp = (double*)((char*)env->ws2 + (size_t)a3 * piece);
                if (blocked)
                    hipLaunchKernelGGL(pgr_unblock_cols, dim3((unsigned)((M + 255) / 256), (unsigned)((S + 3) / 4)), dim3(256), 0, st,
                                       (const double*)jb[a3].src, tmp, all ? (const int*)nullptr : (const int*)didx, M, N, (int)S);
                else
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

#endif  // PGR_TRANSFER_H
