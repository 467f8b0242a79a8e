// pgr_host.h -- host side, common part: error string, PGR_TRACE marks, HIPCHK, the environment object (struct pgr_env) and what the
// library says about its own build (pgr_build_info).
// (Part of the ONE translation unit pgr_hip.hip, included there in this order; not a stand-alone header.)
#ifndef PGR_HOST_H
#define PGR_HOST_H

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
    int api_blocked = 1;              // host-pointer entry / fan handles: HBM-table trajectory fans run the sample-blocked kernel (un-blocked on the way out)
    int persistent = 1;               // fans of several rounds: persistent waves claiming packets from the cost-sorted list (0: whole workgroups, static)
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
    // the fan-kernel instance and launch shape of the LAST pgr_shoot_fan_device on this environment (pgr_debug_last_instance:
    // the test that walks every instance asserts that it launched the one it meant to)
    // (atomics: two host threads may launch on one environment; the record is a diagnostic, each field is whole)
    std::atomic<int> last_instance[8];
    pgr_env() { for (int q = 0; q < 8; q++) last_instance[q].store(q < 4 ? -1 : 0, std::memory_order_relaxed); }
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
        info += "FMA contraction allowed (the PGR_ARITH=contracted opt-in: NOT the reference's bits, no bit-parity claim)";
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

#endif  // PGR_HOST_H
