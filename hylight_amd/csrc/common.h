// common.h - shared host-side plumbing for libhylight_mi.so (errors, device buffers, records).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <algorithm>
#include <vector>

#include "../../include/hylight_mi.h"

namespace hlmi {

// ------------------------------------------------------------------------------------------
// errors: C++ exceptions inside, error codes + thread-local message at the C boundary
// ------------------------------------------------------------------------------------------
struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

[[noreturn]] inline void fail(int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    throw Error(code, buf);
}

#define HIP_CHECK(expr)                                                                       \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            ::hlmi::fail(HLMI_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, \
                         __LINE__);                                                           \
    } while (0)

void set_last_error(const std::string &m);
void require_device();           // throws HLMI_ENODEV when no usable GPU
hipStream_t stream();            // the compute stream of the calling thread's lane (see LaneScope)
int host_threads();

// stats of the last stage (exported through hlmi_last_stats_json)
// Test hooks and tuning switches: the value of a switch the library knows (nullptr: unset).  The table
// is filled from the environment at the start of every C-ABI call (hooks_refresh, capi.cpp:guarded) - nothing else reads it.
const char *hook(const char *name);
void hooks_refresh();
void stat_reset();
void stat_set(const std::string &k, double v);
void stat_add(const std::string &k, double v);
std::map<std::string, double> &stats();

// ------------------------------------------------------------------------------------------
// device memory: size-binned pool in front of hipMalloc / hipFree.  The stage works in query batches
// whose multi-GB scratch buffers have the same sizes batch after batch; hipMalloc + hipFree of those
// cost more than the kernels between them.  A lane's work runs on one stream, so a block can be handed
// out again TO THAT LANE as soon as its owner releases it (the cache is kept per lane: runtime.cpp).
// ------------------------------------------------------------------------------------------
void *dev_alloc(size_t bytes);
void dev_free(void *p);
void dev_pool_trim();            // give every cached block back to the driver
size_t dev_available_bytes();    // free on the card + cached in the pool (0: unknown)
size_t dev_peak_bytes(bool reset);   // high-water mark of the bytes in use through dev_alloc since the last reset

// ------------------------------------------------------------------------------------------
// device buffer (RAII)
// ------------------------------------------------------------------------------------------
template <typename T>
struct DBuf {
    T *p = nullptr;
    size_t n = 0;
    DBuf() = default;
    explicit DBuf(size_t count) { alloc(count); }
    DBuf(const DBuf &) = delete;
    DBuf &operator=(const DBuf &) = delete;
    DBuf(DBuf &&o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
    DBuf &operator=(DBuf &&o) noexcept {
        if (this != &o) { release(); p = o.p; n = o.n; o.p = nullptr; o.n = 0; }
        return *this;
    }
    ~DBuf() { release(); }
    void alloc(size_t count) {
        release();
        n = count;
        if (count) p = (T *)dev_alloc(count * sizeof(T));
    }
    void release() {
        if (p) dev_free(p);
        p = nullptr; n = 0;
    }
    void zero() { if (n) HIP_CHECK(hipMemsetAsync(p, 0, n * sizeof(T), stream())); }
    void fill_ff() { if (n) HIP_CHECK(hipMemsetAsync(p, 0xff, n * sizeof(T), stream())); }
    void upload(const T *h, size_t count) {
        if (count > n) alloc(count);
        if (count) HIP_CHECK(hipMemcpyAsync(p, h, count * sizeof(T), hipMemcpyHostToDevice, stream()));
    }
    void upload(const std::vector<T> &h) { upload(h.data(), h.size()); }
    std::vector<T> download(size_t count) const;
    std::vector<T> download() const { return download(n); }
};

hipStream_t side_stream();       // second stream (runtime.cpp): LONG alignment tasks beside the batch's other DP kernels
hipStream_t side_stream_if_created();      // the same without creating it (nullptr: never used) - for destructors
inline void sync() { HIP_CHECK(hipStreamSynchronize(stream())); }

// Lanes: the query batches of a pass are independent of each other, and a batch alternates between kernels that wait on HBM
// (anchor fill, radix sort) and kernels that wait on the vector units and on dependent loads (chaining, the DP kernels) with
// dozens of host round trips in between.  Two batches in flight - each driven by its own host thread on its own stream - fill
// each other's gaps.  stream(), side_stream(), sync(), the kernel timers, pinned_scratch() and the allocator's cache all refer to
// the lane of the calling thread: lane 0 unless the thread is inside a LaneScope.
constexpr int MAX_LANES = 4;
int lane_count();                // batches in flight (HLMI_LANES, default 2; 1: none beside the caller's)
struct LaneScope {               // the calling thread works on lane `id` while this lives; the lane's stream is drained and its
    explicit LaneScope(int id);  // kernel timers are folded into the stats at the end
    ~LaneScope();
    int prev;
};

// Scoped HIP-event timer on the library stream: per-kernel device time for bench.py's roofline line
// (torch.cuda.Event would only see torch's stream).  Durations are summed per name by ktimer_flush()
// into stats "kernel_ms.<name>" / "kernel_launches.<name>".
struct KTimer {
    explicit KTimer(const char *name, hipStream_t on = nullptr);     // on: the stream the kernel runs on (default: stream())
    ~KTimer();
    size_t slot;
};
void ktimer_flush();
void ktimer_discard();           // drop pending timers without reading them (error paths)
size_t ktimer_mark();            // number of pending timers; ktimer_rollback(mark) forgets the ones started since
void ktimer_rollback(size_t mark);

// Host wall-clock of a scope (stream drained at both ends) -> stats "host_s.<name>"; active only with HLMI_HOST_TIMERS set
struct HostTimer {
    explicit HostTimer(const char *name);
    ~HostTimer();
    const char *name;
    double t0;
};

// 64 KiB of pinned host memory for the small device -> host read-backs (counts, totals): a copy into pageable
// memory takes a staging detour on every one of the ~200 round trips of a stage pass
void *pinned_scratch();
constexpr size_t PINNED_SCRATCH_BYTES = 64 << 10;

template <typename T>
std::vector<T> DBuf<T>::download(size_t count) const {
    std::vector<T> h(count);
    if (!count) return h;
    if (count * sizeof(T) <= PINNED_SCRATCH_BYTES) {
        T *slot = (T *)pinned_scratch();
        HIP_CHECK(hipMemcpyAsync(slot, p, count * sizeof(T), hipMemcpyDeviceToHost, stream()));
        HIP_CHECK(hipStreamSynchronize(stream()));
        std::copy(slot, slot + count, h.begin());
    } else {
        HIP_CHECK(hipMemcpyAsync(h.data(), p, count * sizeof(T), hipMemcpyDeviceToHost, stream()));
        HIP_CHECK(hipStreamSynchronize(stream()));
    }
    return h;
}

template <typename T>
T download_one(const T *dev) {
    T *slot = (T *)pinned_scratch();
    HIP_CHECK(hipMemcpyAsync(slot, dev, sizeof(T), hipMemcpyDeviceToHost, stream()));
    sync();
    return *slot;
}

// blocks of b work-items for a items.  A launch holds fewer than 2^32 work-items per dimension: callers with more than that
// many items launch in slabs (sketch.hip:finish_upload); anything else here is a bug, reported instead of wrapped.
inline unsigned cdiv(size_t a, size_t b) {
    const size_t n = (a + b - 1) / b;
    if (n * b >= (1ull << 32) && b > 1) fail(HLMI_EINVAL, "kernel grid of %zu x %zu work-items (2^32 and more wrap)", n, b);
    return (unsigned)n;
}

// ------------------------------------------------------------------------------------------
// PAF row as the kernels see it (one 64-byte record; rows keep stream order by index)
// ------------------------------------------------------------------------------------------
// CIGAR ops are stored BAM-style in a separate uint32 array: len<<4 | code.
enum : uint32_t { OP_EQ = 7, OP_X = 8, OP_I = 1, OP_D = 2, OP_OTHER = 15 };
enum : uint32_t {
    PF_REV = 1u,       // strand '-'
    PF_STAR = 2u,      // last field is "*" (no CIGAR)
    PF_BAD = 4u,       // row could not be parsed (kept only to preserve window positions)
    PF_GEN = 8u,       // row emitted by the device overlapper: qid/tid are strcmp ranks of the names and the row's text
                       // is a function of its fields, so the whole-line byte order (GNU sort's last resort) is computed
                       // from the fields instead of `tie`
};

struct PafRec {
    uint32_t qid, tid;        // name ids: equal names <=> equal ids (query and target share one space)
    uint32_t qlen, qs, qe;
    uint32_t tlen, ts, te;
    uint32_t nmatch, blen;    // PAF columns 10, 11
    uint32_t flags;
    uint32_t chunk;           // target chunk the row belongs to (per-chunk semantics, utils.py:54)
    uint64_t cig_off;         // first op in the ops array
    uint32_t cig_n;           // number of ops
    uint32_t tie;             // last-resort rank of the row inside its chunk (whole-line byte order)
};
static_assert(sizeof(PafRec) == 64, "PafRec must stay 64 bytes");

}  // namespace hlmi
