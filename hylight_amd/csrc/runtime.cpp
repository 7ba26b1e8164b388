// runtime.cpp - process-level state: device selection, the compute stream, last error, stats.
#include <mutex>

#include "common.h"
#include <sched.h>

#include <iterator>

namespace hlmi {

namespace {
thread_local std::string g_last_error;
bool g_ready = false;
int g_device = 0;                        // the device the library runs on (worker lanes bind their threads to it)
int g_threads = 0;                       // 0: not chosen yet (host_threads)
std::map<std::string, double> g_stats;
std::mutex g_mu;                         // init / shutdown
std::mutex g_stat_mu;                    // g_stats (the lanes add to it)

// A lane: one host thread's view of the device - its compute stream, its side stream, its pending kernel timers, its pinned
// read-back slot, its share of the allocator's cache.  Lane 0 is the calling thread's (everything outside the query-batch loop
// of ava_device runs there); lanes 1.. belong to the worker threads that loop starts (LaneScope).  A thread that never entered
// a LaneScope is on lane 0.
struct KRec { std::string name; hipEvent_t a, b; hipStream_t s; };
struct Lane {
    hipStream_t stream = nullptr, side = nullptr;
    std::vector<KRec> krecs;
    void *pinned = nullptr;
};
Lane g_lanes[MAX_LANES];
thread_local int t_lane = 0;
inline Lane &lane() { return g_lanes[t_lane]; }
}  // namespace

void set_last_error(const std::string &m) { g_last_error = m; }
const std::string &last_error() { return g_last_error; }

// host threads of the text passes: what hlmi_init was given, else the CPUs this process may run on, 16 at most (a rank per
// GPU on a 256-thread host: 8 x 16 leaves half the machine to the caller)
int host_threads() {
    if (g_threads <= 0) {
        cpu_set_t set;
        int n = 8;
        if (sched_getaffinity(0, sizeof set, &set) == 0) n = CPU_COUNT(&set);
        g_threads = std::max(1, std::min(n, 16));
    }
    return g_threads;
}

void *pinned_scratch() {
    void *&p = lane().pinned;
    if (!p) HIP_CHECK(hipHostMalloc(&p, PINNED_SCRATCH_BYTES, hipHostMallocDefault));
    return p;
}

void init_device(int device, int threads) {
    std::lock_guard<std::mutex> lk(g_mu);
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        fail(HLMI_ENODEV, "no HIP device available (%s); libhylight_mi has no CPU fallback",
             e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
    if (device >= n) fail(HLMI_EINVAL, "device %d out of range (have %d)", device, n);
    if (device >= 0) HIP_CHECK(hipSetDevice(device));
    HIP_CHECK(hipGetDevice(&g_device));
    if (!g_lanes[0].stream) HIP_CHECK(hipStreamCreateWithFlags(&g_lanes[0].stream, hipStreamNonBlocking));
    if (threads > 0) g_threads = threads;
    g_ready = true;
}

void shutdown_device() {
    std::lock_guard<std::mutex> lk(g_mu);
    dev_pool_trim();
    for (Lane &l : g_lanes) {
        for (hipStream_t *st : {&l.side, &l.stream}) {
            if (!*st) continue;
            (void)hipStreamSynchronize(*st);
            (void)hipStreamDestroy(*st);
            *st = nullptr;
        }
    }
    g_ready = false;
}

void require_device() {
    if (!g_ready) init_device(-1, 0);
}

hipStream_t stream() { return lane().stream; }
// second stream: the few LONG alignment tasks of a batch (serial rows, a tail of minutes of wave time on a handful of CUs) run
// here beside the batch's other DP kernels; the caller joins it with an event before anything reads their results
hipStream_t side_stream_if_created() { return lane().side; }
hipStream_t side_stream() {
    hipStream_t &g_side = lane().side;
    if (!g_side) {
        // highest priority: its few waves are the tail of the batch - they should never wait for an issue slot behind the
        // thousands of waves of the kernels they run beside
        int least = 0, greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) { (void)hipGetLastError(); greatest = 0; }
        HIP_CHECK(hipStreamCreateWithPriority(&g_side, hipStreamNonBlocking, greatest));
    }
    return g_side;
}

// (the timers of a lane are its own: started, ended and read by its thread)
static void lane_drain() {
    Lane &l = lane();
    if (l.stream) (void)hipStreamSynchronize(l.stream);
    if (l.side) (void)hipStreamSynchronize(l.side);
}

KTimer::KTimer(const char *name, hipStream_t on) {
    KRec r;
    r.name = name;
    r.s = on ? on : lane().stream;
    HIP_CHECK(hipEventCreate(&r.a));
    HIP_CHECK(hipEventCreate(&r.b));
    HIP_CHECK(hipEventRecord(r.a, r.s));
    slot = lane().krecs.size();
    lane().krecs.push_back(r);
}
KTimer::~KTimer() { (void)hipEventRecord(lane().krecs[slot].b, lane().krecs[slot].s); }

void ktimer_flush() {
    auto &krecs = lane().krecs;
    if (krecs.empty()) return;
    lane_drain();
    std::lock_guard<std::mutex> lk(g_stat_mu);
    for (auto &r : krecs) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            g_stats["kernel_ms." + r.name] += ms;
            g_stats["kernel_launches." + r.name] += 1;
        }
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    krecs.clear();
}

size_t ktimer_mark() { return lane().krecs.size(); }
void ktimer_rollback(size_t mark) {            // forget the timers started after `mark` (a run that was given up)
    auto &krecs = lane().krecs;
    if (mark >= krecs.size()) return;
    lane_drain();
    for (size_t i = mark; i < krecs.size(); ++i) { (void)hipEventDestroy(krecs[i].a); (void)hipEventDestroy(krecs[i].b); }
    krecs.resize(mark);
}

void ktimer_discard() {
    auto &krecs = lane().krecs;
    if (krecs.empty()) return;
    lane_drain();
    for (auto &r : krecs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    krecs.clear();
}

// ---- worker lanes ------------------------------------------------------------------------------------
// Query batches in flight at the same time: HLMI_LANES (default 2, 1 = one batch after the other as before round 5).
int lane_count() {
    int n = 2;
    if (const char *e = hook("HLMI_LANES")) n = atoi(e);
    return std::max(1, std::min(n, MAX_LANES));
}
LaneScope::LaneScope(int id) : prev(t_lane) {
    if (id < 0 || id >= MAX_LANES) fail(HLMI_EINVAL, "lane %d out of range", id);
    t_lane = id;
    HIP_CHECK(hipSetDevice(g_device));          // a new thread starts on device 0
    if (!lane().stream) HIP_CHECK(hipStreamCreateWithFlags(&lane().stream, hipStreamNonBlocking));
}
LaneScope::~LaneScope() {
    lane_drain();
    try { ktimer_flush(); } catch (...) {}
    t_lane = prev;
}

// ---- test hooks and tuning switches ------------------------------------------
// The names the library knows, in ONE table.  hooks_refresh() - called at the start of every C-ABI call (capi.cpp:guarded) -
// is the only place that reads the environment; the code paths ask hook(name).
namespace {
const char *const HOOK_NAMES[] = {
    "HLMI_ALIGN_SPAN_TASKS",
    "HLMI_ANCHOR_BATCH_M",
    "HLMI_ANCHOR_PAIRS",
    "HLMI_ANCHOR_SPLIT",
    "HLMI_ASM_STAGE_CAP",
    "HLMI_ASM_WAVE",
    "HLMI_CHAIN_DP16_CHECK",
    "HLMI_CHAIN_NO_DP16",
    "HLMI_CHAIN_NO_SMALL",
    "HLMI_SEED_GROUP", "HLMI_TEXT_GPU", "HLMI_TEXT_HOST", "HLMI_GRAPH_WINDOW_MB", "HLMI_LANES", "HLMI_SET_ASIDE_CUTS", "HLMI_SEED_NO_GUESS",
    "HLMI_CHAIN_PROF",
    "HLMI_CHAIN_UNPACKED",
    "HLMI_GROUP_HIST",
    "HLMI_HOST_TIMERS",
    "HLMI_LONG_SIDE_STREAM",
    "HLMI_NARROW_LONG_UNPACKED",
    "HLMI_NARROW_UNPACKED",
    "HLMI_NO_EXT_CERT",
    "HLMI_NO_GAP1_CERT",
    "HLMI_NO_GAP2_CERT",
    "HLMI_NO_ONE_PIECE_CERT",
    "HLMI_NO_RANK_WORD",
    "HLMI_NO_SHIFT_CERT",
    "HLMI_NO_STUB",
    "HLMI_NO_SUFFIX_TRIM",
    "HLMI_QCAP_MIN",
    "HLMI_RUN_BUF_CAP",
    "HLMI_SKETCH_PART_MBASES",
    "HLMI_SNP_SORT",
    "HLMI_STUB_FULL_ROWS",
    "HLMI_SUBRUN_MAX_MANCHORS",
    "HLMI_SUBRUN_MAX_OUT_MB",
    "HLMI_SUBRUN_MBASES"};
std::unordered_map<std::string, std::string> g_hooks;
}  // namespace
void hooks_refresh() {
    g_hooks.clear();
    for (const char *n : HOOK_NAMES)
        if (const char *v = getenv(n)) g_hooks[n] = v;
}
const char *hook(const char *name) {
    bool known = false;
    for (const char *n : HOOK_NAMES) known = known || strcmp(n, name) == 0;
    if (!known) fail(HLMI_EINVAL, "hook %s is not in the table of runtime.cpp", name);
    auto it = g_hooks.find(name);
    return it == g_hooks.end() ? nullptr : it->second.c_str();
}

// ---- pooled device allocator -----------------------------------------------------------------------
namespace {
// A released block may still be in use by kernels queued on the releasing lane's stream: fine for the next owner on the SAME
// stream, wrong for one on another.  Every cached block remembers the lane that released it; a lane takes its own blocks first,
// and when it takes another lane's, its stream first waits for everything queued on that lane's stream so far (an event recorded
// at the hand-over: whatever used the block was queued before its release).  The paths that give blocks back to the driver
// drain the whole device first.
struct CachedBlock { void *p; int lane; };
std::multimap<size_t, CachedBlock> g_free_blocks;        // size -> block
std::mutex g_pool_mu;
std::unordered_map<void *, size_t> g_block_size;         // every live or cached block
size_t g_pooled_bytes = 0;
size_t g_owned_bytes = 0;                                // every block in g_block_size (in use or cached)
size_t g_peak_in_use = 0;                                // high-water mark of owned - cached (dev_peak_bytes)
constexpr size_t POOL_CAP = 160ull << 30;                // keep at most this much cached (288 GB HBM)
constexpr size_t GRAN = 2ull << 20;
}  // namespace

// Size classes: multiples of 2 MiB up to 64 MiB, above that multiples of an eighth of the size's power of two (at most 12.5 %
// of slack): the multi-GB buffers of consecutive query batches differ by a few per cent and then fall into the same class, so a
// released block serves the next batch instead of sitting in the cache beside a fresh hipMalloc (full C4: the cache grew until
// the card was full, everything was trimmed and allocated again - the steps of a pass took 9, 12, 25 s).
static size_t size_class(size_t bytes) {
    size_t want = (bytes + GRAN - 1) / GRAN * GRAN;
    if (want > (64ull << 20)) {
        size_t p2 = 1;
        while ((p2 << 1) <= want) p2 <<= 1;
        const size_t step = p2 >> 3;
        want = (want + step - 1) / step * step;
    }
    return want;
}

static void pool_trim_locked() {
    (void)hipDeviceSynchronize();
    for (auto &kv : g_free_blocks) {
        (void)hipFree(kv.second.p);
        g_owned_bytes -= kv.first;
        g_block_size.erase(kv.second.p);
    }
    g_free_blocks.clear();
    g_pooled_bytes = 0;
}

void *dev_alloc(size_t bytes) {
    const size_t want = size_class(bytes);
    std::lock_guard<std::mutex> lk(g_pool_mu);
    auto it = g_free_blocks.lower_bound(want);
    {   // a block of this lane among the ones that fit, else the smallest that fits
        int looked = 0;
        for (auto own = it; own != g_free_blocks.end() && own->first <= want + want / 2 && looked < 16; ++own, ++looked)
            if (own->second.lane == t_lane) { it = own; break; }
    }
    if (it != g_free_blocks.end() && it->first <= want + want / 2) {
        void *p = it->second.p;
        const int from = it->second.lane;
        if (from != t_lane && g_lanes[from].stream && lane().stream) {
            hipEvent_t ev;
            HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            HIP_CHECK(hipEventRecord(ev, g_lanes[from].stream));
            HIP_CHECK(hipStreamWaitEvent(lane().stream, ev, 0));
            if (lane().side) HIP_CHECK(hipStreamWaitEvent(lane().side, ev, 0));
            HIP_CHECK(hipEventDestroy(ev));
        }
        g_pooled_bytes -= it->first;
        g_free_blocks.erase(it);
        g_peak_in_use = std::max(g_peak_in_use, g_owned_bytes - g_pooled_bytes);
        return p;
    }
    // a miss: make room first when the card is nearly full - cached blocks go, largest first, until the request fits (a failed
    // hipMalloc and a trim of everything cost a second per 30 GB that has to be allocated again)
    if (!g_free_blocks.empty()) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            bool drained = false;
            while (free_b < want + (1ull << 30) && !g_free_blocks.empty()) {
                if (!drained) { (void)hipDeviceSynchronize(); drained = true; }   // queued kernels may still use a cached block
                auto big = std::prev(g_free_blocks.end());
                (void)hipFree(big->second.p);
                g_block_size.erase(big->second.p);
                g_pooled_bytes -= big->first;
                g_owned_bytes -= big->first;
                free_b += big->first;
                g_free_blocks.erase(big);
            }
        } else (void)hipGetLastError();
    }
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {                                // out of memory: drop the cache and retry once
        (void)hipGetLastError();                          // (the failed call stays "the last error" until it is read: a later
        pool_trim_locked();                               //  HIP_CHECK(hipGetLastError()) after a kernel launch would report it)
        e = hipMalloc(&p, want);
        if (e != hipSuccess) (void)hipGetLastError();
    }
    if (e != hipSuccess) fail(HLMI_ENOMEM, "hipMalloc of %zu bytes failed: %s", want, hipGetErrorString(e));
    g_block_size[p] = want;
    g_owned_bytes += want;
    g_peak_in_use = std::max(g_peak_in_use, g_owned_bytes - g_pooled_bytes);
    return p;
}

// high-water mark of the device memory in use through dev_alloc since the last reset (the stage reports it per run)
size_t dev_peak_bytes(bool reset) {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    const size_t v = g_peak_in_use;
    if (reset) g_peak_in_use = g_owned_bytes - g_pooled_bytes;
    return v;
}

size_t dev_available_bytes() {           // free on the card + cached here: what a run can still take
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return 0; }
    std::lock_guard<std::mutex> lk(g_pool_mu);
    return free_b + g_pooled_bytes;
}

void dev_free(void *p) {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    auto it = g_block_size.find(p);
    if (it == g_block_size.end()) { (void)hipFree(p); return; }
    if (g_pooled_bytes + it->second > POOL_CAP) {
        (void)hipFree(p);
        g_owned_bytes -= it->second;
        g_block_size.erase(it);
        return;
    }
    g_free_blocks.emplace(it->second, CachedBlock{p, t_lane});
    g_pooled_bytes += it->second;
}

void dev_pool_trim() {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    pool_trim_locked();
}

static double wall_s() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
// Draining the stream at both ends costs a launch bubble each time (eight per query batch): the host_s.* statistics
// are collected only when HLMI_HOST_TIMERS is set.
static bool host_timers_on() {
    const bool on = hook("HLMI_HOST_TIMERS") != nullptr;
    return on;
}
HostTimer::HostTimer(const char *n) : name(n), t0(0) {
    if (!host_timers_on()) return;
    if (lane().stream) (void)hipStreamSynchronize(lane().stream);
    t0 = wall_s();
}
HostTimer::~HostTimer() {
    if (!host_timers_on()) return;
    if (lane().stream) (void)hipStreamSynchronize(lane().stream);
    stat_add(std::string("host_s.") + name, wall_s() - t0);
}

void stat_reset() { std::lock_guard<std::mutex> lk(g_stat_mu); g_stats.clear(); }
void stat_set(const std::string &k, double v) { std::lock_guard<std::mutex> lk(g_stat_mu); g_stats[k] = v; }
void stat_add(const std::string &k, double v) { std::lock_guard<std::mutex> lk(g_stat_mu); g_stats[k] += v; }
std::map<std::string, double> &stats() { return g_stats; }

}  // namespace hlmi
