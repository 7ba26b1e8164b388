// ava_chain.hip - S2 index, S3 seeding, S4 chaining of the overlapper spec.
//
//   index   : target minimizers radix-sorted by hash; per (hash, chunk) occurrence counts; per-chunk
//             occurrence cut-off from the count histogram (Li 2016 section 2.3: drop the top fraction)
//   seeds   : one thread per query minimizer: binary search of the hash, walk of its occurrence run,
//             pair-once rule strcmp(qname,tname) < 0 (ava "no dual / no diagonal"), anchors packed as
//             key = qlocal | target:tb | strand:1 | tpos:pb | qpos:qpb | qspan:8 in ONE 64-bit word when the
//             widths of the batch fit (they do for read sets); else key = ... | tpos and val = qpos:32 | qspan:8
//   order   : one stable radix sort per query batch on the (target, strand) bits only: generation is query by query,
//             so the (target, strand, query) groups come out contiguous with their anchors in generation order =
//             ascending query position (the chain kernel reads a reverse-strand group back to front)
//   chains  : one wavefront per (query,target,strand) group.  DP in push form: lane l keeps the anchor with
//             index = l mod 64 among the 64 that follow the anchor being finished and tries that anchor as its
//             predecessor (Li 2018 eq. 1-2, integer gap cost from an LDS table) - no cross-lane reduction; every
//             anchor hands its trunk to its best child (64-bit atomicMax); chain ids by pointer jumping inside
//             64-anchor windows, peaks by atomicMax keyed on the chain start, member lists and fixed points by
//             ballots: cut at the peak, split into alignment pieces of fixed points.
// Integer / index work throughout: HBM- and latency-bound, no MFMA.
#include <algorithm>
#include <type_traits>

#include "ava_internal.h"
#include "dev_prims.h"
#include "wave_ops.h"

namespace hlmi {

namespace {
constexpr int WG = 256;
constexpr int HB = 1024;             // occurrence histogram bins per chunk
// anchor key = qlocal | target | strand | tpos, packed with the bit widths the batch actually needs (pb bits of
// target position, tb bits of target id): the anchor radix sort then runs 5 passes instead of 8 on C2
// instrumentation (phase cycle counters, the 16-predecessor self-check, group size histogram) is compiled in only with
// -DHLMI_INSTRUMENT (HLMI_INSTRUMENT=1 python -m hylight_amd.build): the shipped library carries none of it
#ifdef HLMI_INSTRUMENT
constexpr bool INSTR = true;
#else
constexpr bool INSTR = false;
#endif
constexpr int QL_BITS = 16, T_BITS_MAX = 21, TPOS_BITS_MAX = 29;     // target positions: 29 bits (contigs as targets, HyLight.py:149,180); chain scores are
                                                                      // bounded by the QUERY length (22 bits in the packed DP, else 26: 6 tie-break bits in 32)
inline dim3 grid1(size_t n) { return dim3(cdiv(n ? n : 1, WG)); }

__device__ __forceinline__ size_t lower_bound_u64(const uint64_t *a, size_t n, uint64_t v) {
    size_t lo = 0, hi = n;
    while (lo < hi) {
        size_t m = (lo + hi) >> 1;
        if (a[m] < v) lo = m + 1; else hi = m;
    }
    return lo;
}
__device__ __forceinline__ size_t upper_bound_u32(const uint32_t *a, size_t n, uint32_t v) {
    size_t lo = 0, hi = n;
    while (lo < hi) {
        size_t m = (lo + hi) >> 1;
        if (a[m] <= v) lo = m + 1; else hi = m;
    }
    return lo;
}

// ---------------------------------------------------------------------------------------------
// index
// ---------------------------------------------------------------------------------------------
__global__ void split_mz_kernel(const Mz *mz, size_t n, uint64_t *key, uint64_t *y) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    key[i] = mz[i].x >> 8;
    y[i] = mz[i].y;
}
__global__ void index_head_kernel(const uint64_t *key, const uint64_t *y, const uint32_t *chunk_of_t, size_t n,
                                  uint8_t *head) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    head[i] = (i == 0 || key[i] != key[i - 1] || chunk_of_t[y[i] >> 32] != chunk_of_t[y[i - 1] >> 32]) ? 1 : 0;
}
__global__ void index_occ_kernel(const uint32_t *run_start, size_t n_runs, size_t n, uint32_t *occ) {
    size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (e >= n) return;
    size_t r = upper_bound_u32(run_start, n_runs, (uint32_t)e) - 1;
    size_t end = r + 1 < n_runs ? run_start[r + 1] : n;
    occ[e] = (uint32_t)(end - run_start[r]);
}
// Most runs are one to a few entries long: those counts meet in a small LDS table per workgroup first (a few hundred
// hot global counters took the whole kernel otherwise); chunks beyond the table and longer runs go straight to memory.
constexpr int HIST_LDS_CHUNKS = 256, HIST_LDS_LEN = 8;
__global__ __launch_bounds__(WG) void index_hist_kernel(const uint32_t *run_start, size_t n_runs, size_t n, const uint64_t *y,
                                                         const uint32_t *chunk_of_t, uint32_t *hist) {
    __shared__ uint32_t s_h[HIST_LDS_CHUNKS * HIST_LDS_LEN];
    for (int i = threadIdx.x; i < HIST_LDS_CHUNKS * HIST_LDS_LEN; i += WG) s_h[i] = 0;
    __syncthreads();
    size_t r = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (r < n_runs) {
        size_t end = r + 1 < n_runs ? run_start[r + 1] : n;
        uint32_t len = (uint32_t)(end - run_start[r]);
        uint32_t c = chunk_of_t[y[run_start[r]] >> 32];
        if (c < (uint32_t)HIST_LDS_CHUNKS && len < (uint32_t)HIST_LDS_LEN) atomicAdd(&s_h[c * HIST_LDS_LEN + len], 1u);
        else atomicAdd(&hist[(size_t)c * HB + (len < (uint32_t)HB - 1 ? len : (uint32_t)HB - 1)], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < HIST_LDS_CHUNKS * HIST_LDS_LEN; i += WG)
        if (s_h[i]) atomicAdd(&hist[(size_t)(i / HIST_LDS_LEN) * HB + (size_t)(i % HIST_LDS_LEN)], s_h[i]);
}

// chunk << 32 | run length of every run: sorted, the exact occurrence quantile of a chunk can be read off (used only
// for chunks whose quantile lies beyond the histogram: extremely repetitive input)
__global__ void run_len_key_kernel(const uint32_t *run_start, size_t n_runs, size_t n, const uint64_t *y, const uint32_t *chunk_of_t,
                                   uint64_t *key) {
    size_t r = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (r >= n_runs) return;
    const size_t end = r + 1 < n_runs ? run_start[r + 1] : n;
    key[r] = (uint64_t)chunk_of_t[y[run_start[r]] >> 32] << 32 | (uint32_t)(end - run_start[r]);
}
__global__ void index_rank_kernel(const uint64_t *y, const uint32_t *occ, const uint32_t *mid_occ, const uint32_t *chunk_of_t,
                                  const uint32_t *rank_t, size_t n, uint32_t *rk) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t t = (uint32_t)(y[i] >> 32);
    rk[i] = occ[i] > mid_occ[chunk_of_t[t]] ? 0u : rank_t[t] + 1u;      // 0: too frequent inside its chunk
}
__global__ void compose_ck_kernel(const uint64_t *key, const uint32_t *rk, size_t n, int rb, uint64_t *ck) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) ck[i] = key[i] << rb | rk[i];
}
__global__ void iota_u32_kernel(uint32_t *v, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) v[i] = (uint32_t)i;
}
__global__ void gather_key_kernel(const uint64_t *src, const uint32_t *perm, size_t n, uint64_t *dst) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[perm[i]];
}
__global__ void gather_entry_kernel(const uint64_t *y, const uint32_t *rk, const uint32_t *perm, size_t n, uint64_t *y2, uint32_t *rk2) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) { y2[i] = y[perm[i]]; rk2[i] = rk[perm[i]]; }
}
__global__ void index_bucket_kernel(const uint64_t *key, size_t n, int shift, size_t n_buckets, uint32_t *bucket) {
    size_t b = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (b <= n_buckets) bucket[b] = b == n_buckets ? (uint32_t)n : (uint32_t)lower_bound_u64(key, n, (uint64_t)b << shift);
}

// ---------------------------------------------------------------------------------------------
// seeds
// ---------------------------------------------------------------------------------------------
struct SeedArgs {
    const Mz *qmz;              // first minimizer of the batch
    size_t n_mz;
    const uint64_t *ikey, *iy, *ick;    // ick: key << rb | rank word (rb = 0: not built)
    int rb;
    const uint32_t *irk, *bucket, *rank_q, *qlen;
    uint32_t *run_lo, *run_len; // occurrence run of every query minimizer: written by the counting pass, read by the fill
    int bucket_shift;
    size_t n_idx;
    uint32_t q_lo;
    int pair_once;              // 1: only strcmp(qname,tname) < 0; 0: every pair except self
    int pb, tb;                 // key layout: tpos bits, target bits
    int vb;                     // payload bits below tpos: 8 + qpos bits when the anchor is one word, 0 with a value array
    int sk;                     // 0, or the bytes (2 / 4) of a separate (target, strand) key: the word then holds the rest
    void *oskey;                // that key array
    uint32_t n_ranks;           // name ranks run over [0, n_ranks): where in a key's run a rank is expected (0: no guess)
};

// Wave-cooperative: a wave owns 64 consecutive query minimizers.  Every lane binary-searches the
// occurrence run of its own minimizer; the wave then serves the 64 runs one after the other with the
// lanes striding the run, so index reads and anchor writes are coalesced and the anchors of one
// minimizer keep ascending occurrence order (ballot + popcount ranks the survivors).
template <bool FILL>
__global__ __launch_bounds__(WG) void seed_kernel(SeedArgs a, uint32_t *cnt, const uint64_t *aoff, uint64_t *okey,
                                                   uint64_t *oval) {
    const int lane = threadIdx.x & 63;
    const size_t m = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const bool live = m < a.n_mz;
    Mz z{0, 0};
    size_t lo = 0, hi = 0;
    uint32_t rq = 0, ql = 0;
    uint64_t w0 = 0;
    if (live) {
        z = a.qmz[m];
        const uint32_t q = (uint32_t)(z.y >> 32);
        rq = a.rank_q[q];
        if (FILL) {
            lo = a.run_lo[m];
            hi = lo + a.run_len[m];
            ql = a.qlen[q];
            w0 = aoff[m];
        } else {        // the key's bucket bounds the search to a few entries
            const uint64_t key = z.x >> 8;
            const size_t b = (size_t)(key >> a.bucket_shift);
            const size_t b_lo = a.bucket[b], b_hi = a.bucket[b + 1];
            // the entries are ordered by (key, rank word) with rank word 0 = too frequent, else name rank + 1: the ones
            // this query may pair with are a suffix of the key's run - counting is two searches, not a scan
            // (the searches are bound by the 64-byte lines they touch: one word per entry, one read per step)
            auto first_ge = [&](size_t from, uint64_t k, uint32_t r) {       // first entry >= (k, r) in [from, b_hi)
                size_t l = from, h = b_hi;
                if (a.rb) {
                    const uint64_t want = k << a.rb | r;
                    while (l < h) {
                        const size_t mid = (l + h) >> 1;
                        if (a.ick[mid] < want) l = mid + 1; else h = mid;
                    }
                    return l;
                }
                while (l < h) {
                    const size_t mid = (l + h) >> 1;
                    const uint64_t km = a.ikey[mid];
                    if (km < k || (km == k && a.irk[mid] < r)) l = mid + 1; else h = mid;
                }
                return l;
            };
            uint32_t c = 0;
            constexpr int BK = 16;                               // a bucket of up to BK words is fetched whole
            if (a.rb && b_hi - b_lo <= (size_t)BK) {
                // The searches below are chains of dependent 8-byte loads into the same one or two cache lines (the
                // kernel waited on them 96 % of its time).  The bucket's words are requested together instead and
                // counted in registers: entries below (key, r) = words below key << rb | r.
                const int nb = (int)(b_hi - b_lo);
                uint64_t w[BK];
#pragma unroll
                for (int k = 0; k < BK; ++k) w[k] = k < nb ? a.ick[b_lo + (size_t)k] : ~0ull;
                auto below = [&](uint64_t want) {
                    int n = 0;
#pragma unroll
                    for (int k = 0; k < BK; ++k) n += (k < nb && w[k] < want) ? 1 : 0;
                    return (size_t)n;
                };
                auto upto_key = [&]() {                          // entries with hash <= key (no key + 1: it may not fit)
                    int n = 0;
#pragma unroll
                    for (int k = 0; k < BK; ++k) n += (k < nb && (w[k] >> a.rb) <= key) ? 1 : 0;
                    return (size_t)n;
                };
                hi = b_lo + upto_key();
                if (a.pair_once) {
                    lo = b_lo + below(key << a.rb | (uint64_t)(rq + 2));
                    lo = lo < hi ? lo : hi;
                    c = (uint32_t)(hi - lo);
                } else {
                    lo = b_lo + below(key << a.rb | 1u);
                    size_t e0 = b_lo + below(key << a.rb | (uint64_t)(rq + 1)), e1 = b_lo + below(key << a.rb | (uint64_t)(rq + 2));
                    lo = lo < hi ? lo : hi; e0 = e0 < hi ? e0 : hi; e1 = e1 < hi ? e1 : hi;
                    c = (uint32_t)((hi - lo) - (e1 - e0));
                }
            } else if (a.pair_once) {                            // only partners that rank above the query
                // A key's run is ordered by name rank, and the targets of a sub-run are spread evenly over the ranks: the first
                // partner above rank r sits near the r / n_ranks point of the run.  Sixteen words around that point (two
                // adjacent lines, requested together) often hold the answer; otherwise they bound the
                // binary search to one side.  The result is the binary search's: the count pass read 13 dependent lines per
                // query minimizer - 192 GB per C3 step against 39 GB of algorithmic bytes (profiles/r05f_c3_pmc_traffic.txt).
                size_t s_lo = b_lo, s_hi = b_hi;
                bool found = false;
                const uint64_t last_w = a.rb && b_hi > b_lo ? a.ick[b_hi - 1] : 0;     // (requested with the window, used below)
                if (a.rb && a.n_ranks && b_hi - b_lo > (size_t)BK) {
                    const uint64_t want = key << a.rb | (uint64_t)(rq + 2);
                    const size_t len = b_hi - b_lo;
                    size_t g = b_lo + (size_t)((unsigned long long)len * (rq + 2 < a.n_ranks ? rq + 2 : a.n_ranks) / a.n_ranks);
                    size_t w0 = g & ~(size_t)(BK - 1);             // (two whole lines; a window centred on g touched a third and was slower)
                    if (w0 < b_lo) w0 = b_lo;
                    if (w0 + BK > b_hi) w0 = b_hi - BK;
                    uint64_t w[BK];
#pragma unroll
                    for (int k = 0; k < BK; ++k) w[k] = a.ick[w0 + (size_t)k];
                    int nbel = 0;
#pragma unroll
                    for (int k = 0; k < BK; ++k) nbel += w[k] < want ? 1 : 0;
                    if (nbel == 0) s_hi = w0;                    // the answer is at or before the window's first word
                    else if (nbel == BK) s_lo = w0 + BK;         // ... behind its last
                    else { lo = w0 + (size_t)nbel; found = true; }
                }
                if (!found) {
                    if (a.rb) {
                        const uint64_t want = key << a.rb | (uint64_t)(rq + 2);
                        size_t l = s_lo, h = s_hi;
                        while (l < h) {
                            const size_t mid = (l + h) >> 1;
                            if (a.ick[mid] < want) l = mid + 1; else h = mid;
                        }
                        lo = l;
                    } else lo = first_ge(b_lo, key, rq + 2);
                }
                // (at read-set depth a bucket is one key's run: when the bucket's last entry carries the key, the run ends with
                //  the bucket - one word instead of a second search over the same hundred entries)
                if (a.rb && b_hi > b_lo && (last_w >> a.rb) == key) hi = b_hi > lo ? b_hi : lo;
                else hi = first_ge(lo, key + 1, 0u);
                c = (uint32_t)(hi - lo);
            } else {                                             // every partner but the read itself
                lo = first_ge(b_lo, key, 1u);
                const size_t e0 = first_ge(lo, key, rq + 1), e1 = first_ge(e0, key, rq + 2);
                hi = first_ge(e1, key + 1, 0u);
                c = (uint32_t)((hi - lo) - (e1 - e0));
            }
            a.run_lo[m] = (uint32_t)lo;
            a.run_len[m] = (uint32_t)(hi - lo);
            cnt[m] = c;
        }
    }
    if (!FILL) return;
    // one anchor of run b (this lane's entry e of it) - shared by the two loops below
    auto emit = [&](bool ok, unsigned long long at, uint64_t y, unsigned long long zx, unsigned long long zy, uint32_t ql_b) {
        if (!ok) return;
        const uint32_t qspan = (uint32_t)(zx & 0xff), qpos = (uint32_t)zy >> 1, qz = (uint32_t)zy & 1;
        const uint32_t q = (uint32_t)(zy >> 32), t = (uint32_t)(y >> 32);
        const uint32_t tpos = (uint32_t)y >> 1, strand = qz ^ ((uint32_t)y & 1);
        const uint32_t qp = strand ? ql_b - (qpos + 1 - qspan) - 1 : qpos;
        const uint64_t kk = (uint64_t)(q - a.q_lo) << (a.tb + 1 + a.pb) | (uint64_t)t << (1 + a.pb) |
                            (uint64_t)strand << a.pb | tpos;
        if (a.sk) {         // the sorted bits apart: (target, strand) in a small key, the rest in one word
            okey[at] = ((uint64_t)(q - a.q_lo) << a.pb | tpos) << a.vb | (uint64_t)qp << 8 | qspan;
            if (a.sk == 2) ((uint16_t *)a.oskey)[at] = (uint16_t)(t << 1 | strand);
            else ((uint32_t *)a.oskey)[at] = t << 1 | strand;
        } else if (a.vb) okey[at] = kk << a.vb | (uint64_t)qp << 8 | qspan;
        else { okey[at] = kk; oval[at] = (uint64_t)qp << 32 | (uint64_t)qspan << 24; }
    };
    // Runs of more than SHORT_RUN entries: one after the other, all 64 lanes striding the run; the first 64 entries of the
    // next run are already in flight while this one is worked on (two dependent reads per run would otherwise sit on the
    // critical path every time).
    constexpr int SHORT_RUN = 32;
    const unsigned long long any_run = __ballot(hi != lo);
    unsigned long long todo = __ballot(hi - lo > (size_t)SHORT_RUN);
    uint64_t y_p = 0;
    uint32_t rt_p = 0;
    auto prefetch = [&](int bn) {
        const unsigned long long e = __shfl((unsigned long long)lo, bn, 64) + lane;
        if (e < __shfl((unsigned long long)hi, bn, 64)) { y_p = a.iy[e]; if (!a.pair_once) rt_p = a.irk[e]; }
    };
    if (todo) prefetch(__ffsll((long long)todo) - 1);
    while (todo) {
        const int b = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const unsigned long long lo_b = __shfl((unsigned long long)lo, b, 64), hi_b = __shfl((unsigned long long)hi, b, 64);
        const uint32_t rq_b = __shfl(rq, b, 64);
        const unsigned long long zx = __shfl((unsigned long long)z.x, b, 64), zy = __shfl((unsigned long long)z.y, b, 64);
        unsigned long long w = __shfl((unsigned long long)w0, b, 64);
        const uint32_t ql_b = __shfl(ql, b, 64);
        uint64_t y = y_p;
        uint32_t rt = rt_p;
        if (todo) prefetch(__ffsll((long long)todo) - 1);
        for (unsigned long long e0 = lo_b; e0 < hi_b; e0 += 64) {
            const unsigned long long e = e0 + lane;
            if (e0 != lo_b && e < hi_b) { y = a.iy[e]; if (!a.pair_once) rt = a.irk[e]; }
            // the run holds partners only (counting pass), except the read itself when pairs are taken both ways
            const bool ok = e < hi_b && (a.pair_once || rt != rq_b + 1u);
            const unsigned long long mask = __ballot(ok);
            emit(ok, w + __popcll(mask & ((1ull << lane) - 1)), y, zx, zy, ql_b);
            w += (uint32_t)__popcll(mask);
        }
    }
    // Short runs, FOUR at a time: a quarter of the wave each (sub-runs of a few thousand targets against a million
    // queries - the full C4 - have ~5 partners per query minimizer: 64 lanes on one run left nine tenths of them idle).
    unsigned long long todo_s = any_run & ~__ballot(hi - lo > (size_t)SHORT_RUN);
    const int g = lane >> 4, sl = lane & 15;
    while (todo_s) {
        int bsel[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { bsel[k] = todo_s ? __ffsll((long long)todo_s) - 1 : -1; todo_s &= todo_s ? todo_s - 1 : 0; }
        const int b = g == 0 ? bsel[0] : (g == 1 ? bsel[1] : (g == 2 ? bsel[2] : bsel[3]));
        const int bs = b >= 0 ? b : 0;
        const unsigned long long lo_b = __shfl((unsigned long long)lo, bs, 64);
        const unsigned long long hi_all = __shfl((unsigned long long)hi, bs, 64);      // (every lane takes part in the shuffle)
        const unsigned long long hi_b = b >= 0 ? hi_all : lo_b;
        const uint32_t rq_b = __shfl(rq, bs, 64);
        const unsigned long long zx = __shfl((unsigned long long)z.x, bs, 64), zy = __shfl((unsigned long long)z.y, bs, 64);
        unsigned long long w = __shfl((unsigned long long)w0, bs, 64);
        const uint32_t ql_b = __shfl(ql, bs, 64);
#pragma unroll
        for (int it = 0; it < SHORT_RUN / 16; ++it) {
            const unsigned long long e = lo_b + (unsigned long long)(16 * it + sl);
            uint64_t y = 0;
            uint32_t rt = 0;
            if (e < hi_b) { y = a.iy[e]; if (!a.pair_once) rt = a.irk[e]; }
            const bool ok = e < hi_b && (a.pair_once || rt != rq_b + 1u);
            const unsigned long long mask = __ballot(ok);
            const uint32_t mg = (uint32_t)(mask >> (16 * g)) & 0xffffu;
            emit(ok, w + __popc(mg & ((1u << sl) - 1u)), y, zx, zy, ql_b);
            w += (uint32_t)__popc(mg);
            if (!__any(lo_b + 16ull * (it + 1) < hi_b)) break;
        }
    }
}

__global__ void gather_u32_at_kernel(const uint32_t *src, const uint32_t *idx, uint32_t *dst, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
}
__global__ void mark_proven_kernel(const uint32_t *gorder, const uint8_t *verdict, size_t n, uint8_t *flag) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n && verdict[i] == 1) flag[gorder[i]] = 1;
}
__global__ void plus_one_u8_kernel(const uint8_t *in, uint8_t *out, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i] + 1;
}
__global__ void gather_u64_at_kernel(const uint64_t *src, const uint64_t *idx, uint64_t *dst, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
}
// sort key that puts the big groups first: the hardware hands workgroups out in order, so the long groups start
// early and the tail of a launch is made of small ones
// groups of more than SMALL_N anchors go through chain_kernel (a wave per group), the others through chain_small_kernel (a lane
// per group): groups above SMALL_N and anchors of the others, one atomic pair per wave into one of TALLY_SLOTS cache lines (a short-
// read call has a million waves of small groups a step: on one address their atomics took longer than the kernel's real work)
constexpr int SMALL_N = 32;
constexpr int TALLY_SLOTS = 64, TALLY_STRIDE = 16;       // 16 x 8 B = one 128-byte line per slot
__global__ void group_size_key_kernel(const uint32_t *gstart, const uint32_t *gsize, size_t n_groups, size_t n_anchors, uint32_t *key,
                                      uint32_t *order, unsigned long long *tally) {
    size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    uint32_t sz = 0;
    if (g < n_groups) {
        const size_t e = g + 1 < n_groups ? gstart[g + 1] : n_anchors;
        sz = gsize ? gsize[g] : (uint32_t)(e - gstart[g]);
        key[g] = 0xffffu - (sz < 0xffffu ? sz : 0xffffu);
        order[g] = (uint32_t)g;
    }
    const unsigned long long big = __ballot(sz > (uint32_t)SMALL_N);
    uint32_t small_anchors = sz > (uint32_t)SMALL_N ? 0u : sz;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) small_anchors += __shfl_xor(small_anchors, o, 64);
    if ((threadIdx.x & 63) == 0) {
        unsigned long long *slot = tally + (size_t)(blockIdx.x % TALLY_SLOTS) * TALLY_STRIDE;
        if (big) atomicAdd(&slot[0], (unsigned long long)__popcll(big));
        if (small_anchors) atomicAdd(&slot[1], (unsigned long long)small_anchors);
    }
}

// HLMI_GROUP_HIST=1: groups and anchors per power-of-two size class (tuning aid, statistics group_hist_*)
__global__ void group_hist_kernel(const uint32_t *gstart, const uint32_t *gsize, size_t n_groups, size_t n_anchors, unsigned long long *hist) {
    __shared__ unsigned long long h[64];
    if (threadIdx.x < 64) h[threadIdx.x] = 0;
    __syncthreads();
    size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (g < n_groups) {
        const size_t e = g + 1 < n_groups ? gstart[g + 1] : n_anchors;
        const uint32_t sz = gsize ? gsize[g] : (uint32_t)(e - gstart[g]);
        const int c = 31 - __clz((int)sz);
        atomicAdd(&h[c], 1ull);
        atomicAdd(&h[32 + c], (unsigned long long)sz);
    }
    __syncthreads();
    if (threadIdx.x < 64 && h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}

// ---------------------------------------------------------------------------------------------
// chains
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int ilog2_u32(uint32_t v) { return 31 - __clz((int)v); }

__device__ __forceinline__ bool block_ok(int q0, int t0, int q1, int t1, int shift_max) {
    // any length (blocks above BLOCK_MAX rows / columns are LONG blocks of the alignment pass); only the
    // diagonal shift is bounded - the band has BAND_W diagonals
    int m = q1 - q0, n = t1 - t0, d = n - m;
    if (m < 0 || n < 0) return false;
    return (d < 0 ? -d : d) <= shift_max;       // (SHIFT_MAX; 0 with bandwidth 0: ungapped alignment)
}

struct ChainArgs {
    const uint64_t *key, *val;
    const void *skey;                 // (target << 1 | strand) of every anchor when the key word does not hold them
    int sk;                           //   its width in bytes (2 / 4), 0: the bits sit in `key` above the target position
    const uint32_t *gstart;
    const uint32_t *gsize, *gq, *gts; // group records of seed_group.hip (nullptr: groups are the runs of the sorted batch, their
                                      // query / target / strand sit in the anchors)
    const uint32_t *gorder;           // groups, largest first (the order the workgroups take them in)
    size_t n_list;                    // groups in gorder (this launch's share of the n_groups groups)
    size_t n_groups, n_anchors;
    uint32_t *fp;                     // MODE 2: score << 5 | distance to the predecessor (0: none) of every anchor, from chain_dp16_kernel
    uint32_t *sbase;                  // per chain start: f(parent of the start) | has a child << 31
    uint32_t *starts;                 // per group, in its range [first anchor, ...): the chain starts that can become a chain
                                      // (a child, or min_cnt <= 1), ascending - the candidate scan reads these few instead of
                                      // the chain id of every anchor
    int *root;                        // chain id of every anchor (index of the chain's start inside the group)
    unsigned long long *peak;         // per chain, at its root: best f << 32 | ~(first index reaching it)
    int k, max_gap, bw, min_score, min_cnt;
    int shift_max;                    // largest diagonal shift of one alignment block
    int pb, tb, vb;                   // key layout (SeedArgs)
    uint64_t pmask;                   // (1 << pb) - 1
    uint32_t qmask;                   // (1 << qpos bits) - 1
    uint32_t q_lo;
    Piece *pieces;
    FixPt *fps;
    uint32_t cap_pieces, cap_fps;
    uint32_t *counters;               // [0] pieces, [1] fixed points written (a 64-bit pair for chain_small_kernel), [2] overflow flag, [3] groups through the full DP
    unsigned long long *prof;         // HLMI_CHAIN_PROF: wave cycles per phase [0] block loop [1] candidates + member lists [2] fixed points
    const uint8_t *check_ok;          // self-check (HLMI_CHAIN_DP16_CHECK): per group, 1 = chain_dp16_kernel claims equality;
    unsigned long long *check_bad;    //   the full DP then compares its scores / predecessors with fp and counts differences
};

// query in the batch || target || strand of group g, whose first anchor is b
__device__ __forceinline__ uint64_t group_word(const ChainArgs &a, size_t g, size_t b) {
    if (a.gsize) return (uint64_t)a.gq[g] << (a.tb + 1) | a.gts[g];
    const uint64_t top = a.key[b] >> (a.vb + a.pb);
    if (!a.sk) return top;
    const uint32_t ts = a.sk == 2 ? (uint32_t)((const uint16_t *)a.skey)[b] : ((const uint32_t *)a.skey)[b];
    return top << (a.tb + 1) | ts;
}
__device__ __forceinline__ size_t group_end(const ChainArgs &a, size_t g, size_t b) {
    if (a.gsize) return b + a.gsize[g];
    return g + 1 < a.n_groups ? (size_t)a.gstart[g + 1] : a.n_anchors;
}
// target position, query position, span of anchor idx
__device__ __forceinline__ void anchor_fields(const ChainArgs &a, size_t idx, int &t, int &q, int &sp) {
    const uint64_t key = a.key[idx];
    if (a.vb) {
        sp = (int)(key & 0xff); q = (int)((uint32_t)(key >> 8) & a.qmask); t = (int)((key >> a.vb) & a.pmask);
    } else {
        const uint64_t val = a.val[idx];
        t = (int)(key & a.pmask); q = (int)(val >> 32); sp = (int)((val >> 24) & 0xff);
    }
}
// Fixed-point selection over the members of one chain (oracle/ava_oracle.c:align_chain).
// 64 anchors at a time sit in registers; the next fixed point = first later member that is >= BLOCK_MIN away
// in both sequences (or the last member) comes from one ballot instead of a scan over ~16 anchors.
// One pass: the fixed points go into a range reserved for the worst case (a chain of len members has at most
// 2 * len of them: one per member plus one more per piece), every piece takes its slot when it closes.
// Finished pieces wait in a small LDS buffer of the wave and go out PIECE_BUF at a time: one returning atomic on
// the piece counter per flush instead of one per piece (2 M pieces a step on one address cost a third of the
// kernel), and the records leave as one coalesced store.
constexpr int PIECE_BUF = 32;
__device__ __forceinline__ void flush_pieces(const ChainArgs &a, int lane, const Piece *buf, int &n_buf) {
    if (!n_buf) return;
    uint32_t slot = 0;
    if (lane == 0) slot = atomicAdd(&a.counters[0], (uint32_t)n_buf);
    slot = (uint32_t)__builtin_amdgcn_readfirstlane((int)slot);
    if ((unsigned long long)slot + (unsigned)n_buf > a.cap_pieces) { if (lane == 0) a.counters[2] = 1; }
    else if (lane < n_buf) a.pieces[slot + lane] = buf[lane];
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    n_buf = 0;
}
// The members of chain s are the anchors s <= i <= peak whose chain id (root) is s.  There is no member list: the
// windows below are 64 consecutive ANCHORS of the group (coalesced reads of the anchors and their chain ids), lanes
// that hold another chain's anchor are masked and carry the coordinates of the member before them, which keeps the
// coordinates monotone along the window for the successor search.  Returns false (nothing written) when the chain has
// fewer than min_cnt members.
__device__ bool emit_chain(const ChainArgs &a, size_t b, size_t g_first, long long g_step, int lane, int s, int peak_i,
                           uint32_t q, uint32_t t, uint32_t strand, uint32_t &n_pieces, uint32_t &n_fps, uint32_t fp_base,
                           Piece *buf, int &n_buf) {
    bool open = false;
    int cq = 0, ct = 0;
    uint32_t np = 0, nf = 0, piece_fp0 = 0;
    const bool wr = lane == 0;
    n_pieces = n_fps = 0;
    auto close_piece = [&]() {
        if (wr) buf[n_buf] = Piece{q, t, strand, (uint32_t)s, np, fp_base + piece_fp0, nf - piece_fp0, 0};
        ++np;
        if (++n_buf == PIECE_BUF) { __builtin_amdgcn_s_waitcnt(0); __builtin_amdgcn_wave_barrier(); flush_pieces(a, lane, buf, n_buf); }
    };
    const int end = peak_i + 1;                                // the peak is the chain's last member
    int base = s, te_l = 0, qe_l = 0, sp_l = 0;
    unsigned long long memb = 0;                               // member lanes of the window
    // Successor links of the window (long chains): nxt = the member that becomes the next fixed point when this one
    // is the current one (64: none inside the window), okm = lanes whose block to that successor passes block_ok.
    // Member coordinates grow strictly, so "far enough in both sequences, or the last member" is monotone along
    // the window: a 6-round binary search per lane, once per window; the serial part of a run of fixed points is
    // then one readlane per fixed point, and the run is stored by all its lanes at once.
    int nxt = 64, links_base = -(1 << 30);
    unsigned long long okm = 0;
    // raw window: end coordinates + span of the anchors x0 .. x0 + 63, chain id in rt (-1 past the peak)
    auto load_window = [&](int x0, int &te, int &qe, int &sp, int &rt) {
        te = qe = sp = 0; rt = -1;
        if (x0 + lane < end) {
            rt = a.root[b + (size_t)(x0 + lane)];
            anchor_fields(a, (size_t)((long long)g_first + g_step * (x0 + lane)), te, qe, sp);
            ++te; ++qe;
        }
    };
    // masks the other chains' anchors: they take the coordinates of the last member at or before them
    auto settle = [&](int te, int qe, int sp, int rt) {
        memb = __ballot(rt == s);
        const unsigned long long upto = memb & ((2ull << lane) - 1ull);
        const int src = upto ? 63 - __clzll((long long)upto) : lane;
        te_l = __shfl(te, src, 64); qe_l = __shfl(qe, src, 64); sp_l = sp;
        if (!upto) { te_l = 0; qe_l = 0; }
    };
    int te_n = 0, qe_n = 0, sp_n = 0, rt_n = -1;
    {
        int te0, qe0, sp0, rt0;
        load_window(s, te0, qe0, sp0, rt0);
        if (s + 64 < end) load_window(s + 64, te_n, qe_n, sp_n, rt_n);
        settle(te0, qe0, sp0, rt0);
        int cnt = __popcll(memb);
        if (cnt < a.min_cnt) {                                 // (rare) the first window does not settle it: count on
            if (s + 64 < end) cnt += __popcll(__ballot(rt_n == s));
            for (int w0 = s + 128; w0 < end && cnt < a.min_cnt; w0 += 64)
                cnt += __popcll(__ballot(w0 + lane < end && a.root[b + (size_t)(w0 + lane)] == s));
            if (cnt < a.min_cnt) return false;
        }
    }
    for (int x = s; x < end;) {
        if (x >= base + 64) {                                  // (always the window that follows: x never skips one)
            base += 64;
            settle(te_n, qe_n, sp_n, rt_n);
            if (base + 64 < end) load_window(base + 64, te_n, qe_n, sp_n, rt_n);
        }
        if (!open) {                                           // a piece starts at the START of the next member from x on
            const unsigned long long rem = memb & ~((1ull << (x - base)) - 1ull);
            if (!rem) { x = base + 64; continue; }
            const int l = __ffsll((long long)rem) - 1;
            const int te = __builtin_amdgcn_readlane(te_l, l), qe = __builtin_amdgcn_readlane(qe_l, l),
                      sp = __builtin_amdgcn_readlane(sp_l, l);
            int q0 = qe - sp, t0 = te - sp;
            if (q0 < 0 || t0 < 0) { int sh = q0 < t0 ? -q0 : -t0; q0 += sh; t0 += sh; }
            if (q0 < 0) q0 = 0;
            if (t0 < 0) t0 = 0;
            if (block_ok(q0, t0, qe, te, a.shift_max)) {
                piece_fp0 = nf;
                if (wr) { a.fps[fp_base + nf] = FixPt{(uint32_t)q0, (uint32_t)t0}; a.fps[fp_base + nf + 1] = FixPt{(uint32_t)qe, (uint32_t)te}; }
                nf += 2;
                cq = qe; ct = te; open = true;
            }
            x = base + l + 1;
            continue;
        }
        // open piece: first member j >= x in the window that qualifies as the next fixed point
        const int j_l = base + lane;
        const bool ok = ((memb >> lane) & 1ull) && j_l >= x && qe_l > cq && te_l > ct &&
                        ((qe_l - cq >= BLOCK_MIN && te_l - ct >= BLOCK_MIN) || j_l == peak_i);
        const unsigned long long m = __ballot(ok);
        if (!m) { x = base + 64; continue; }                   // nothing in this window
        const int l = __ffsll((long long)m) - 1;
        const int te = __builtin_amdgcn_readlane(te_l, l), qe = __builtin_amdgcn_readlane(qe_l, l);
        if (block_ok(cq, ct, qe, te, a.shift_max)) {
            const int wend = end - base < 64 ? end - base : 64;
            if (wend - l < 8) {                                // few anchors left in the window: one at a time
                if (wr) a.fps[fp_base + nf] = FixPt{(uint32_t)qe, (uint32_t)te};
                ++nf;
                cq = qe; ct = te;
                x = base + l + 1;
                continue;
            }
            if (links_base != base) {
                links_base = base;
                int lo = lane + 1, hi = wend;
                for (int r = 0; r < 6; ++r) {
                    const int mid = (lo + hi) >> 1;
                    const int qm = __shfl(qe_l, mid & 63, 64), tm = __shfl(te_l, mid & 63, 64);
                    const bool far = (qm - qe_l >= BLOCK_MIN && tm - te_l >= BLOCK_MIN) || base + mid == peak_i;
                    if (lo < hi) { if (far) hi = mid; else lo = mid + 1; }
                }
                nxt = lo < wend ? lo : 64;
                const int qn = __shfl(qe_l, nxt & 63, 64), tn = __shfl(te_l, nxt & 63, 64);
                okm = __ballot(nxt < 64 && block_ok(qe_l, te_l, qn, tn, a.shift_max));
            }
            unsigned long long vis = 0;                        // l and the fixed points that follow it in this window
            int e = l;
            for (;;) {
                vis |= 1ull << e;
                if (!((okm >> e) & 1ull)) break;               // successor outside the window, or a split: generic path
                e = __builtin_amdgcn_readlane(nxt, e);
            }
            if ((vis >> lane) & 1ull)
                a.fps[fp_base + nf + (uint32_t)__popcll(vis & ((1ull << lane) - 1ull))] = FixPt{(uint32_t)qe_l, (uint32_t)te_l};
            nf += (uint32_t)__popcll(vis);
            cq = __builtin_amdgcn_readlane(qe_l, e); ct = __builtin_amdgcn_readlane(te_l, e);
            x = base + e + 1;
        } else {                                               // split: close here, reopen at this member
            close_piece();
            open = false;
            x = base + l;
        }
    }
    if (open) close_piece();
    n_pieces = np;
    n_fps = nf;
    return true;
}

constexpr int PEN_TAB = 2048;         // gap-cost table entries (bandwidth + 2 must fit; else the VALU form runs)

// The gap cost is an LDS table look-up (dd*k/100 would be two quarter-rate multiplies per lane).
// CHAIN_GROUPS consecutive groups per wave, CHAIN_WAVES independent waves per workgroup: group sizes span three
// orders of magnitude, so the balancing is left to the hardware dispatcher (a fixed grid-stride split of the
// groups left the SIMDs at 3 of 8 resident waves on average).
constexpr int BC_RING = 256;         // best-child ring: this block, the two before (still read), one spare
constexpr int CHAIN_GROUPS = 4;
constexpr int CHAIN_WAVES = 4;      // waves per workgroup sharing one gap-cost table: LDS per wave sets the occupancy (1: 3.4 waves/SIMD, 74 ms; 2: 64 ms; 4: 58 ms; 8: 60 ms on C2)
// TAB: 0 = gap cost computed, 1 = byte table (the index is the LDS address), 2 = 16-bit table,
//      3 = packed DP state (score << 8 | predecessor stamp, see below) with a 32-bit table of -(cost << 8)
constexpr int PK_NONE = 255;         // stamp of "no predecessor"
constexpr int PK_NEG = -(1 << 30);   // candidate that loses against every state (scores stay below 2^22)
// ---------------------------------------------------------------------------------------------
// chains, first pass: the DP over the 16 nearest predecessors, four groups per wave
// ---------------------------------------------------------------------------------------------
// On read sets the anchors of a (query, target, strand) group lie along one diagonal and f grows by about a k-mer per
// anchor: the best predecessor is then one of the nearest few, and most of the 64 candidates the specification asks for
// only cost instructions.  This kernel runs the same push-form recurrence with a window of 16 - a DPP row of 16 lanes
// holds one group, a wave four of them, the sender's state reaches its row through row_newbcast - and PROVES per
// anchor that the other 48 predecessors cannot change the result: a predecessor j scores at most f(j) + span_i, so
//     max { f(j) : i - 64 <= j <= i - 17 } + span_i <= f16(i)
// means no far predecessor beats the near maximum (on a tie the spec takes the later predecessor, which is the near
// one).  By induction over the group f16 = f and the predecessors agree.  A group with one anchor that fails the test
// is handed to the 64-predecessor kernel instead (repeats, off-diagonal seeds).  Output: score << 5 | distance to the
// predecessor per anchor (4 bytes; chain_kernel<3, true> reads them), one verdict per group.
constexpr int D16_WAVES = 4;
// lane J of the own row of 16, to every lane of it: v_mov_b32_dpp row_newbcast.  The empty asm keeps the move a move:
// folded into an arithmetic instruction as its DPP operand (v_subrev_u32_dpp ... row_newbcast:J) the result was wrong on
// gfx950 (measured; the same code with ds_bpermute or with the unfolded move agrees with the CPU).
template <int J> __device__ __forceinline__ int row_lane(int x) {
    int r = __builtin_amdgcn_update_dpp(0, x, 0x150 + J, 0xf, 0xf, true);
    asm volatile("" : "+v"(r));
    return r;
}
__device__ __forceinline__ int row_max_u_incl_prefix(int x) {               // unsigned max over lanes <= l of the row (identity 0)
    auto mx = [](int a, int b) { return (int)((uint32_t)a > (uint32_t)b ? (uint32_t)a : (uint32_t)b); };
    x = mx(x, dpp_i32<0x111>(0, x)); x = mx(x, dpp_i32<0x112>(0, x)); x = mx(x, dpp_i32<0x114>(0, x)); x = mx(x, dpp_i32<0x118>(0, x));
    return x;
}
__device__ __forceinline__ int row_max_u_incl_suffix(int x) {               // ... over lanes >= l (row_shl)
    auto mx = [](int a, int b) { return (int)((uint32_t)a > (uint32_t)b ? (uint32_t)a : (uint32_t)b); };
    x = mx(x, dpp_i32<0x101>(0, x)); x = mx(x, dpp_i32<0x102>(0, x)); x = mx(x, dpp_i32<0x104>(0, x)); x = mx(x, dpp_i32<0x108>(0, x));
    return x;
}

struct Dp16State {
    int M_t, M_q, M_s, M_pk, N_t, N_q, N_s, N_pk, O_pk, w;
};
// the anchor held by lane J of every row becomes the sender: its position goes to the row, the lane itself moves on to
// its next anchor (16 further), and every lane gets the part of the candidate that does not depend on the sender's score
template <int J>
__device__ __forceinline__ void dp16_prepare(Dp16State &z, const int *pen_tab, uint32_t lim4, uint32_t bw4) {
    constexpr unsigned long long me = 0x0001000100010001ull << J;
    const int tj = row_lane<J>(z.M_t), qj = row_lane<J>(z.M_q);
    z.M_t = select_by_mask(me, z.N_t, z.M_t); z.M_q = select_by_mask(me, z.N_q, z.M_q); z.M_s = select_by_mask(me, z.N_s, z.M_s);
    const uint32_t dr = (uint32_t)(z.M_t - tj), dq = (uint32_t)(z.M_q - qj);          // 4 x gap; a negative gap is huge
    const uint32_t hi = dr < dq ? dq : dr;
    uint32_t di = sad_u32(dr, dq, 0u);
    di = di < bw4 ? di : bw4;
    const int pen = *(const int *)((const char *)pen_tab + di);
    // the smaller gap, capped by the span (>= k for an anchor, 0 in the lanes past the group's end): one v_min3, and
    // "both gaps >= 1" is a test of that minimum
    const uint32_t mn = min(min(dr, dq), (uint32_t)z.M_s);
    z.w = ((mn >= 4u) & (hi <= lim4)) ? (int)(mn << 6) + pen : PK_NEG;
}
// steps J .. 15 of a block: finish the sender, push it, prepare the next one
template <int J>
__device__ __forceinline__ void dp16_steps(Dp16State &z, const int *pen_tab, uint32_t lim4, uint32_t bw4) {
    constexpr unsigned long long me = 0x0001000100010001ull << J;
    const int sb = row_lane<J>(z.M_pk);
    z.O_pk = select_by_mask(me, sb, z.O_pk);
    z.M_pk = select_by_mask(me, z.N_pk, z.M_pk);
    const int cand = ((sb & ~255) + (16 + J - 256)) + z.w;    // the table holds (1 - cost) << 8
    z.M_pk = cand > z.M_pk ? cand : z.M_pk;
    if constexpr (J < 15) {
        dp16_prepare<J + 1>(z, pen_tab, lim4, bw4);
        dp16_steps<J + 1>(z, pen_tab, lim4, bw4);
    }
}

// the wave's four groups gorder[gi0 .. gi0 + 3], one per row of 16 lanes; pen_tab = the packed DP's table ((1 - cost) << 8,
// entry bw + 1 rejecting).  Returns the row's verdict (the same in its 16 lanes).
// (Round 3, tried and dropped: a group in which every anchor's predecessor is the anchor before it is one chain and needs no
// bookkeeping at all; with the scores kept out of memory for such groups the pass had to run twice for the others - and on C3
// only 10 % of the groups are linear, a single stray anchor among ~390 breaks it: chain 190 -> 256 ms.)
__device__ int dp16_groups(const ChainArgs &a, const int *pen_tab, size_t gi0, uint32_t *fp) {
    const int lane = threadIdx.x & 63, rl = lane & 15, row = lane >> 4;
    const size_t gi = gi0 + (size_t)row;
    // geometry of the row's group (uniform inside the row)
    size_t g = 0, g_first = 0;
    long long g_step = 1;
    int n = 0;
    if (gi < a.n_list) {
        g = a.gorder[gi];
        const size_t b = a.gstart[g], e = group_end(a, g, b);
        n = (int)(e - b);
        if (n < a.min_cnt) n = 0;                                  // nobody chains it
        const uint32_t strand = (uint32_t)group_word(a, g, b) & 1u;
        g_first = strand ? b + (size_t)(e - b) - 1 : b;
        g_step = strand ? -1 : 1;
    }
    int n_max = __builtin_amdgcn_readlane(n, 0);
    n_max = max(n_max, __builtin_amdgcn_readlane(n, 16));
    n_max = max(n_max, __builtin_amdgcn_readlane(n, 32));
    n_max = max(n_max, __builtin_amdgcn_readlane(n, 48));
    if (!n_max) return 2;
    constexpr int DEAD_Q = -(1 << 30);
    auto load_block = [&](int i0, int &t, int &q, int &sp) {       // positions x 4 (byte offsets into the table), span x 4
        t = 0; q = DEAD_Q; sp = 0;
        if (i0 + rl < n) {
            anchor_fields(a, (size_t)((long long)g_first + g_step * (i0 + rl)), t, q, sp);
            t <<= 2; q <<= 2; sp <<= 2;
        }
    };
    Dp16State z;
    int P_t, P_q, P_s;
    load_block(0, z.M_t, z.M_q, z.M_s);
    load_block(16, z.N_t, z.N_q, z.N_s);
    z.M_pk = (z.M_s << 6) | PK_NONE;
    const uint32_t lim4 = 4u * (uint32_t)a.max_gap, bw4 = 4u * (uint32_t)(a.bw + 1);
    int F1 = 0, F2 = 0, F3 = 0, F4 = 0, R1 = 0, R2 = 0, R3 = 0;    // scores of the four blocks before / their row maxima
    int gmax = 0;                                                  // largest score of the group
    unsigned long long bad = 0;
    for (int i0 = 0; i0 < n_max; i0 += 16) {
        // every lane l holds anchor i0 + l here; N = the block after it, P = the one after that
        load_block(i0 + 32, P_t, P_q, P_s);                        // in flight during the block
        const int cur_span = z.M_s >> 2;                           // the bound below needs the spans of this block's anchors
        z.N_pk = (z.N_s << 6) | PK_NONE;
        z.O_pk = PK_NONE;
        dp16_prepare<0>(z, pen_tab, lim4, bw4);
        dp16_steps<0>(z, pen_tab, lim4, bw4);
        // every lane holds an anchor of the next block now: stamps of this block's senders become "previous block"
        z.M_pk -= (z.M_pk & 255) != PK_NONE ? 16 : 0;
        const bool live = i0 + rl < n;
        const int O_f = live ? z.O_pk >> 8 : 0, O_st = z.O_pk & 255;
        if (live) {
            const uint32_t off = O_st == PK_NONE ? 0u : (uint32_t)(rl + 16 - O_st);
            fp[(size_t)((long long)g_first + g_step * (i0 + rl))] = (uint32_t)O_f << 5 | off;
        }
        // far predecessors of lane l: block -4 lanes >= l, blocks -3 and -2, block -1 lanes < l
        {
            const int far4 = row_max_u_incl_suffix(F4);
            const int pre1 = dpp_i32<0x111>(0, row_max_u_incl_prefix(F1));       // exclusive
            int far = far4 > R3 ? far4 : R3;
            far = far > R2 ? far : R2;
            far = far > pre1 ? far : pre1;
            bad |= __ballot(live && far + cur_span > O_f);
            F4 = F3; F3 = F2; F2 = F1; F1 = O_f;
            R3 = R2; R2 = R1; R1 = row_lane<15>(row_max_u_incl_prefix(O_f));
            gmax = gmax > R1 ? gmax : R1;
        }
        // the block after next moves up
        z.N_t = P_t; z.N_q = P_q; z.N_s = P_s;
    }
    const unsigned long long mine = (bad >> (16 * row)) & 0xffffull;
    // 1: proven.  2: proven, and no chain of the group can reach the minimum score (a chain scores at most its largest
    // f): nobody needs to look at it again.  0: the full DP decides.
    return mine ? 0 : (gmax < a.min_score || n == 0 ? 2 : 1);
}


// the pass on its own (self-check mode: its output is compared with the 64-predecessor DP of every group)
__global__ __launch_bounds__(64 * D16_WAVES) void chain_dp16_kernel(ChainArgs a, uint32_t *fp, uint8_t *gok) {
    __shared__ int pen_tab[PEN_TAB];
    for (int d = threadIdx.x; d < a.bw + 2; d += 64 * D16_WAVES) {
        const int pen = d && d <= a.bw ? (d * a.k) / 100 + (ilog2_u32((uint32_t)d) >> 1) : 0;
        pen_tab[d] = d <= a.bw ? (1 - pen) * 256 : PK_NEG;
    }
    __syncthreads();
    const size_t gi0 = ((size_t)blockIdx.x * D16_WAVES + (threadIdx.x >> 6)) * 4;
    const int verdict = dp16_groups(a, pen_tab, gi0, fp);
    const int lane = threadIdx.x & 63;
    if ((lane & 15) == 0 && gi0 + (size_t)(lane >> 4) < a.n_list) gok[gi0 + (size_t)(lane >> 4)] = (uint8_t)verdict;
}

// MODE 0: the 64-predecessor DP for every group.  MODE 2 (packed form only): the wave first runs the 16-predecessor DP
// over its four groups (dp16_groups); a group with the proof only needs the chain bookkeeping, read from fp, a group
// without it takes the 64-predecessor DP as before, a group that cannot reach the minimum score is dropped.
template <int TAB, int MODE = 0>
__global__ __launch_bounds__(64 * CHAIN_WAVES) __attribute__((amdgpu_waves_per_eu(MODE == 2 ? 8 : 4, 8))) void chain_kernel(ChainArgs a) {
    static_assert(MODE == 0 || TAB == 3, "the 16-predecessor pass uses the packed table");
    typedef typename std::conditional<TAB == 1, uint8_t, typename std::conditional<TAB == 3, int, uint16_t>::type>::type pen_t;
    __shared__ pen_t pen_tab[TAB ? PEN_TAB : 1];
    __shared__ unsigned long long s_bc[CHAIN_WAVES][BC_RING];      // best child of the anchors of the last few blocks
    unsigned long long *bc = s_bc[threadIdx.x >> 6];
    __shared__ Piece s_pieces[CHAIN_WAVES][PIECE_BUF];
    Piece *pbuf = s_pieces[threadIdx.x >> 6];
    int n_pbuf = 0;
    const int lane = threadIdx.x & 63;
    if (TAB) {
        for (int d = threadIdx.x; d < a.bw + 2; d += 64 * CHAIN_WAVES) {
            const int pen = d && d <= a.bw ? (d * a.k) / 100 + (ilog2_u32((uint32_t)d) >> 1) : 0;
            // packed form: (1 - cost) << 8 (the 1 undoes the -1 the gap terms carry), entry bw + 1 rejects
            pen_tab[d] = TAB == 3 ? (pen_t)(d <= a.bw ? (1 - pen) * 256 : PK_NEG) : (pen_t)pen;
        }
        __syncthreads();
    }
    const size_t g_lo = ((size_t)blockIdx.x * CHAIN_WAVES + (threadIdx.x >> 6)) * CHAIN_GROUPS;
    const size_t g_hi = g_lo + CHAIN_GROUPS < a.n_list ? g_lo + CHAIN_GROUPS : a.n_list;
    uint32_t wave_fps = 0;                                 // fixed points actually written by this wave (statistics)
    int verdict16 = 0;
    if constexpr (MODE == 2) {
        static_assert(CHAIN_GROUPS == 4, "one group per row of 16 lanes");
        const long long td0 = (INSTR && a.prof) ? (long long)__builtin_readcyclecounter() : 0;
        verdict16 = g_lo < a.n_list ? dp16_groups(a, (const int *)pen_tab, g_lo, a.fp) : 2;
        if (INSTR && a.prof && lane == 0) atomicAdd(&a.prof[4], (unsigned long long)((long long)__builtin_readcyclecounter() - td0));
        __builtin_amdgcn_s_waitcnt(0);                     // fp is read back by this wave below
        __threadfence_block();
        if (lane == 0) {
            uint32_t n_full = 0;
            for (int r = 0; r < 4; ++r) n_full += g_lo + (size_t)r < g_hi && __builtin_amdgcn_readlane(verdict16, 16 * r) == 0 ? 1u : 0u;
            if (n_full) atomicAdd(&a.counters[3], n_full);
        }
    }
    for (size_t gi = g_lo; gi < g_hi; ++gi) {
        bool use_pre = false;
        if constexpr (MODE == 2) {
            const int vd = __builtin_amdgcn_readlane(verdict16, 16 * (int)(gi - g_lo));
            if (vd == 2) continue;
            use_pre = vd == 1;
        }
        const size_t g = (uint32_t)__builtin_amdgcn_readfirstlane((int)a.gorder[gi]);
        const size_t b = (uint32_t)__builtin_amdgcn_readfirstlane((int)a.gstart[g]);
        const size_t e = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)group_end(a, g, b));
        const int n = (int)(e - b);
        if (n < a.min_cnt) continue;
        // ---- DP ---------------------------------------------------------------------------------------------------
        // "Push" form of the recurrence: lane l holds the anchor with index = l mod 64 among the 64 that FOLLOW the
        // anchor j being finished (M_*: position, span, best score so far + 1 / best predecessor).  Step j reads the
        // finished f(j) from lane j % 64, hands that lane to anchor j + 64 (N_*: the next 64 anchors, loaded a block
        // ahead) and lets every lane try j as its predecessor: no cross-lane reduction, and the part that does not
        // depend on f (gap geometry, gap cost look-up) is computed one step ahead.  Predecessors arrive in ascending
        // order, so "candidate >= best" gives ties to the closest one; M_best starts at span + 1 so that the first
        // predecessor needs candidate > span.
        const uint64_t key0 = group_word(a, g, b);
        const uint32_t qg = a.q_lo + (uint32_t)(key0 >> (a.tb + 1));
        const uint32_t tg = (uint32_t)(key0 >> 1) & ((1u << a.tb) - 1);
        const uint32_t strand = (uint32_t)key0 & 1u;
        // anchor i of the group in chaining order (ascending aligned query position) sits at b + i, or at
        // b + n - 1 - i on the reverse strand
        const size_t g_first = strand ? b + (size_t)n - 1 : b;
        const long long g_step = strand ? -1 : 1;
        constexpr int DEAD_Q = -(1 << 30);                    // lanes past the end of the group: dq < 0 fails the gap test
        auto load_block = [&](int i0, int &t, int &q, int &sp) {
            t = 0; q = DEAD_Q; sp = 0;
            if (i0 + lane < n) anchor_fields(a, (size_t)((long long)g_first + g_step * (i0 + lane)), t, q, sp);
        };
        int M_t = 0, M_q = 0, M_s = 0, N_t = 0, N_q = 0, N_s = 0, P_t, P_q, P_s;
        if (!use_pre) {
            load_block(0, M_t, M_q, M_s);
            load_block(64, N_t, N_q, N_s);
        }
        int M_best = M_s + 1, M_bp = -1;
        auto prepare = [&](int jl, int &w) {
            const int tj = __builtin_amdgcn_readlane(M_t, jl), qj = __builtin_amdgcn_readlane(M_q, jl);
            const unsigned long long me = 1ull << jl;         // this lane now receives for anchor j + 64
            M_t = select_by_mask(me, N_t, M_t); M_q = select_by_mask(me, N_q, M_q); M_s = select_by_mask(me, N_s, M_s);
            const int dr = M_t - tj, dq = M_q - qj;           // dq >= 0: the group is in query order
            const int dg = dr < dq ? dr : dq, mx = dr < dq ? dq : dr, dd = mx - dg;
            const bool ok = (dg >= 1) & (mx <= a.max_gap) & (dd <= a.bw);
            int pen;
            if (TAB) {
                const uint32_t di = (uint32_t)dd < (uint32_t)(a.bw + 1) ? (uint32_t)dd : (uint32_t)(a.bw + 1);
                pen = pen_tab[di];
            } else {
                pen = dd ? (dd * a.k) / 100 + (ilog2_u32((uint32_t)dd) >> 1) : 0;
            }
            w = ok ? (dg < M_s ? dg : M_s) - pen : -(1 << 30);
        };
        // ---- chains without walking them --------------------------------------------------------------------------
        // An anchor belongs to the chain of its parent when it is that parent's best child, else it starts a chain.
        // Parents are at most 64 back, so once block B has voted the best children of block B-1's parents are final
        // and the chain id ("root" = index of the start) of every anchor of block B-1 follows from the roots of block
        // B-2 by pointer jumping inside the block (<= 6 rounds of one ds_bpermute).  Per anchor only the root goes to
        // memory; a start also stores f(parent) and whether it has a child; the peak of every chain (max f, first
        // index) is a 64-bit atomicMax keyed by the root - the chain owning the block's best f, the long one, sends
        // one atomic for all its lanes.  f, p and the best-child ring never leave the wave.
        for (int k = lane; k < BC_RING; k += 64) bc[k] = 0;
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
        long long tp0 = (INSTR && a.prof) ? (long long)__builtin_readcyclecounter() : 0, tpc = 0;
        int P1_f = 0, P1_p = -1, P2_f = 0;                 // f / p of the block before, f of the one before that
        int prev_root = 0;                                 // roots of the last resolved block (lane = index mod 64)
        int n_starts = 0;                                  // chain starts listed so far (a.starts[b ..])
        auto resolve = [&](int w0, int Rf, int Rp, int Bf) {
            const int i = w0 + lane;
            const bool live = i < n;
            const int pi = live ? Rp : -1;
            bool child = false, has_child = false;
            int val = 0;                                   // val >= 0: root; val < 0: ~lane of the parent (same block)
            if (live) {
                child = pi >= 0 && (0xffffffffu - (uint32_t)(bc[pi & (BC_RING - 1)] & 0xffffffffull)) == (uint32_t)i;
                has_child = bc[i & (BC_RING - 1)] != 0;
                val = !child ? i : ~(pi - w0);
            }
            // every lane takes part in the shuffles (ds_bpermute reads 0 from idle lanes)
            const int from_prev = __shfl(prev_root, (pi - w0 + 64) & 63, 64);
            const int f_same = __shfl(Rf, pi & 63, 64), f_before = __shfl(Bf, pi & 63, 64);
            if (child && pi < w0) val = from_prev;
            const unsigned long long open_m = __ballot(val < 0);
            if (open_m && !__ballot(val < 0 && ~val != lane - 1)) {
                // every unresolved anchor hangs on the anchor right before it (the usual block: one chain, no stray anchor in
                // between): its root is the value of the nearest resolved lane below - one shuffle instead of six rounds
                const unsigned long long res = ~open_m & ((2ull << lane) - 1ull);          // (lane 0 is never unresolved here:
                val = __shfl(val, 63 - __clzll((long long)res), 64);                       //  its parent lies in the block before)
            } else
            while (__ballot(val < 0)) {
                const int up = __shfl(val, val < 0 ? ~val : lane, 64);
                if (val < 0) val = up;                     // parent resolved: its root; else jump to the parent's parent
            }
            if (live) {
                a.root[b + i] = val;
                if (val == i) {
                    a.sbase[b + i] = (uint32_t)(pi < 0 ? 0 : (pi >= w0 ? f_same : f_before)) | (has_child ? 0x80000000u : 0u);
                    a.peak[b + i] = 0;                     // a chain's peak word starts here: every vote for it comes later,
                }                                          // from this wave (no memset of the whole array per batch)
            }
            {   // a childless start is a one-anchor chain: it can only survive when min_cnt <= 1
                const bool st = live && val == i && (a.min_cnt <= 1 || has_child);
                const unsigned long long sm = __ballot(st);
                if (st) a.starts[b + (size_t)n_starts + (size_t)__popcll(sm & ((1ull << lane) - 1ull))] = (uint32_t)i;
                n_starts += __popcll(sm);
            }
            __builtin_amdgcn_s_waitcnt(0);                 // the zeros are out before the first votes (same wave, same address)
            prev_root = val;
            // peaks: f << 32 | ~index, max = highest f, first index among equals
            const uint32_t k32 = live ? (uint32_t)Rf << 6 | (uint32_t)(63 - lane) : 0u;
            const uint32_t kmax = wave_max_u32_dpp(k32);
            const int lmax = 63 - (int)(kmax & 63u);
            const int rmax = __builtin_amdgcn_readlane(val, lmax);
            if (live && (val != rmax || lane == lmax))
                atomicMax(&a.peak[b + val], (unsigned long long)(uint32_t)Rf << 32 | (0xffffffffu - (uint32_t)i));
            // the votes for the block before this one have served their purpose: its ring slots are free for the block
            // three ahead (this block's own are still read when the next one is resolved)
            bc[(i - 64) & (BC_RING - 1)] = 0;
        };
        if (MODE == 2 && use_pre) {
            // scores and predecessors come from dp16_groups (the 16-predecessor DP, verified equal to the
            // 64-predecessor one for this group): only the chain bookkeeping runs here
            auto load_fp = [&](int i0) { return i0 + lane < n ? a.fp[(size_t)((long long)g_first + g_step * (i0 + lane))] : 0u; };
            uint32_t v_next = load_fp(0);
            for (int i0 = 0; i0 < n; i0 += 64) {
                const uint32_t v = v_next;
                v_next = load_fp(i0 + 64);                        // in flight while this block is resolved
                int O_f = 0, O_p = -1;
                if (i0 + lane < n) {
                    O_f = (int)(v >> 5);
                    O_p = (v & 31u) ? i0 + lane - (int)(v & 31u) : -1;
                }
                if (i0 + lane < n && O_p >= 0)
                    atomicMax(&bc[O_p & (BC_RING - 1)], (unsigned long long)(uint32_t)O_f << 32 | (0xffffffffu - (uint32_t)(i0 + lane)));
                __builtin_amdgcn_s_waitcnt(0xc07f);               // LDS only (lgkmcnt 0): the prefetch stays in flight
                __builtin_amdgcn_wave_barrier();
                if (i0) resolve(i0 - 64, P1_f, P1_p, P2_f);
                P2_f = P1_f; P1_f = O_f; P1_p = O_p;
            }
        } else if constexpr (TAB == 3) {
            // Packed state: one word per lane = best score << 8 | stamp of the predecessor that gave it (PK_NONE: none,
            // the score is then the anchor's own span).  Step jl of a block stamps 64 + jl; when a block ends every
            // lane holds an anchor of the next block and real stamps drop by 64, so that a finished anchor of block B
            // reads predecessor = 64 (B - 1) + stamp.  "candidate >= best, later predecessor wins ties, the first
            // predecessor must beat span" is then ONE signed max: stamps grow with the step, PK_NONE beats every
            // stamp at equal score.  Positions are kept times 4 (the gap difference is the byte offset into the
            // table) and minus one anchor step (0 <= 4 (gap - 1) < 4 max_gap is one unsigned compare per sequence).
            // 19 vector instructions per step instead of 30.
            const int lim4 = 4 * a.max_gap, bw4 = 4 * (a.bw + 1);
            auto scale = [&](int &t, int &q, int &sp) { t <<= 2; q = q == DEAD_Q ? DEAD_Q : q << 2; sp = (sp << 2) - 4; };
            scale(M_t, M_q, M_s);
            scale(N_t, N_q, N_s);
            int M_pk = ((M_s + 4) << 6) | PK_NONE;
            auto prepare4 = [&](int jl, int &w) {
                const int tj = scalar_add(__builtin_amdgcn_readlane(M_t, jl), 4), qj = scalar_add(__builtin_amdgcn_readlane(M_q, jl), 4);
                const unsigned long long me = 1ull << jl;
                M_t = select_by_mask(me, N_t, M_t); M_q = select_by_mask(me, N_q, M_q); M_s = select_by_mask(me, N_s, M_s);
                const int dr = M_t - tj, dq = M_q - qj;          // 4 (gap - 1)
                const bool ok = ((uint32_t)dr < (uint32_t)lim4) & ((uint32_t)dq < (uint32_t)lim4);
                uint32_t di = sad_u32((uint32_t)dr, (uint32_t)dq, 0u);
                di = di < (uint32_t)bw4 ? di : (uint32_t)bw4;
                const int pen = *(const int *)((const char *)pen_tab + di);
                int mn = dr < dq ? dr : dq;
                mn = mn < M_s ? mn : M_s;
                w = ok ? mn * 64 + pen : PK_NEG;
            };
            for (int i0 = 0; i0 < n; i0 += 64) {
                if (i0) { N_t = P_t; N_q = P_q; N_s = P_s; }
                load_block(i0 + 128, P_t, P_q, P_s);          // in flight during this block
                scale(P_t, P_q, P_s);
                const int N_pk = ((N_s + 4) << 6) | PK_NONE;
                const int nb = __builtin_amdgcn_readfirstlane(n - i0 < 64 ? n - i0 : 64);
                int O_pk = PK_NONE;
                int w_cur;
                prepare4(0, w_cur);
                for (int jl = 0; jl < nb; ++jl) {
                    const int sb = __builtin_amdgcn_readlane(M_pk, jl);
                    O_pk = writelane_i32(O_pk, sb, jl);
                    M_pk = select_by_mask(1ull << jl, N_pk, M_pk);
                    const int cand = ((sb & ~255) + 64 + jl) + w_cur;
                    M_pk = cand > M_pk ? cand : M_pk;
                    int w_nxt;
                    prepare4((jl + 1) & 63, w_nxt);           // (past the block end: hands a lane over twice, w unused)
                    w_cur = w_nxt;
                }
                if (nb == 64) M_pk -= (M_pk & 255) != PK_NONE ? 64 : 0;
                const int O_f = O_pk >> 8, O_st = O_pk & 255;
                const int O_p = O_st == PK_NONE ? -1 : i0 - 64 + O_st;
                if (INSTR && a.check_ok && a.check_ok[g] && i0 + lane < n) {
                    const uint32_t v = a.fp[(size_t)((long long)g_first + g_step * (i0 + lane))];
                    const int f16 = (int)(v >> 5), p16 = (v & 31u) ? i0 + lane - (int)(v & 31u) : -1;
                    if (f16 != O_f || p16 != O_p) {
                        if (atomicAdd(a.check_bad, 1ull) == 0) {
                            a.check_bad[1] = g; a.check_bad[2] = (unsigned long long)(i0 + lane); a.check_bad[3] = (unsigned long long)n;
                            a.check_bad[4] = (unsigned long long)(uint32_t)f16 << 32 | (uint32_t)p16;
                            a.check_bad[5] = (unsigned long long)(uint32_t)O_f << 32 | (uint32_t)O_p;
                        }
                    }
                }
                if (i0 + lane < n && O_p >= 0)
                    atomicMax(&bc[O_p & (BC_RING - 1)], (unsigned long long)(uint32_t)O_f << 32 | (0xffffffffu - (uint32_t)(i0 + lane)));
                __builtin_amdgcn_s_waitcnt(0);
                __builtin_amdgcn_wave_barrier();
                if (i0) resolve(i0 - 64, P1_f, P1_p, P2_f);
                P2_f = P1_f; P1_f = O_f; P1_p = O_p;
            }
        } else
        for (int i0 = 0; i0 < n; i0 += 64) {
            if (i0) { N_t = P_t; N_q = P_q; N_s = P_s; }
            load_block(i0 + 128, P_t, P_q, P_s);              // in flight during this block
            const int N_b = N_s + 1;
            const int nb = __builtin_amdgcn_readfirstlane(n - i0 < 64 ? n - i0 : 64);
            int O_f = 0, O_p = -1;
            int w_cur;
            prepare(0, w_cur);
            for (int jl = 0; jl < nb; ++jl) {
                const int sb = __builtin_amdgcn_readlane(M_best, jl), sp = __builtin_amdgcn_readlane(M_bp, jl);
                const int fj = sp < 0 ? sb - 1 : sb;
                O_f = writelane_i32(O_f, fj, jl);
                O_p = writelane_i32(O_p, sp, jl);
                M_best = select_by_mask(1ull << jl, N_b, M_best);
                M_bp = writelane_i32(M_bp, -1, jl);
                const int cand = fj + w_cur;
                const bool take = cand >= M_best;
                M_best = take ? cand : M_best;
                M_bp = take ? i0 + jl : M_bp;
                int w_nxt;
                prepare(jl + 1 < nb ? jl + 1 : jl, w_nxt);    // (the repeat at the block end is idempotent, its w unused)
                w_cur = w_nxt;
            }
            // best child of the parents (they sit in this block or the one before): 64-bit max of f << 32 | ~index in LDS
            if (i0 + lane < n && O_p >= 0)
                atomicMax(&bc[O_p & (BC_RING - 1)], (unsigned long long)(uint32_t)O_f << 32 | (0xffffffffu - (uint32_t)(i0 + lane)));
            __builtin_amdgcn_s_waitcnt(0);
            __builtin_amdgcn_wave_barrier();
            if (i0) resolve(i0 - 64, P1_f, P1_p, P2_f);       // every child of the block before has voted now
            P2_f = P1_f; P1_f = O_f; P1_p = O_p;
        }
        resolve((n - 1) & ~63, P1_f, P1_p, P2_f);
        __threadfence_block();
        if (INSTR && a.prof) { const long long t = (long long)__builtin_readcyclecounter(); if (lane == 0) atomicAdd(&a.prof[0], (unsigned long long)(t - tp0)); tp0 = t; }
        uint32_t fcur = 2u * (uint32_t)b;                  // next free fixed-point slot of the group
        for (int k0 = 0; k0 < n_starts; k0 += 64) {
            int pk_i = 0, s_mine = 0;
            bool cand = false;
            if (k0 + lane < n_starts) {
                s_mine = (int)a.starts[b + (size_t)(k0 + lane)];
                const uint32_t sb = a.sbase[b + s_mine];    // f(parent of the start) | has a child << 31
                const unsigned long long pk = a.peak[b + s_mine];
                pk_i = (int)(0xffffffffu - (uint32_t)(pk & 0xffffffffull));
                cand = (int)(pk >> 32) - (int)(sb & 0x7fffffffu) >= a.min_score;
            }
            unsigned long long cm = __ballot(cand);
            while (cm) {
                const int l = __ffsll((long long)cm) - 1;
                cm &= cm - 1;
                const int s = __builtin_amdgcn_readlane(s_mine, l);
                const int peak_i = __builtin_amdgcn_readlane(pk_i, l);
                // fixed points of this chain: at most 2 per member, and the chains of a group have disjoint members -
                // the group's range of the array (twice its anchors) is handed out chain after chain, no counter needed
                uint32_t np = 0, nf = 0;
                const long long te = (INSTR && a.prof) ? (long long)__builtin_readcyclecounter() : 0;
                emit_chain(a, b, g_first, g_step, lane, s, peak_i, qg, tg, strand, np, nf, fcur, pbuf, n_pbuf);
                fcur += nf;
                if (INSTR && a.prof) tpc += (long long)__builtin_readcyclecounter() - te;
                wave_fps += nf;
            }
        }
        if (INSTR && a.prof && lane == 0) {
            const long long t = (long long)__builtin_readcyclecounter();
            atomicAdd(&a.prof[1], (unsigned long long)(t - tp0 - tpc));
            atomicAdd(&a.prof[2], (unsigned long long)tpc);
            atomicAdd(&a.prof[3], 1ull);
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    flush_pieces(a, lane, pbuf, n_pbuf);
    if (lane == 0 && wave_fps) atomicAdd(&a.counters[1], wave_fps);
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// chains of the small groups: one LANE per group
// ---------------------------------------------------------------------------------------------
// chain_kernel spends a wave on a group: right for the groups of long reads (hundreds of anchors), 6 000 wave instructions for
// the 14 anchors a short read leaves on a contig (C4s: 66 M groups a step, 360 ms).  A group of at most SMALL_N anchors is
// chained by one lane instead, with the loops of the specification as they stand in oracle/ava_oracle.c (chain_group: every
// predecessor of the group, closest first; best children; chains from their starts, cut at the peak; align_chain: fixed
// points) over arrays in LDS ([anchor][lane]: conflict-free for the common index).  Groups arrive sorted by size, so the 64
// groups of a wave run about equally long.  Pieces: the logic runs twice - first counting, then, after ONE slot request per
// wave, writing.
__global__ __launch_bounds__(64) void chain_small_kernel(ChainArgs a, size_t first, size_t count) {
    __shared__ int s_t[SMALL_N][64], s_q[SMALL_N][64], s_f[SMALL_N][64];
    __shared__ uint8_t s_sp[SMALL_N][64], s_p[SMALL_N][64], s_bc[SMALL_N][64], s_path[SMALL_N][64];
    const int lane = threadIdx.x;
    const size_t u = (size_t)blockIdx.x * 64 + (size_t)lane;
    int n = 0;
    size_t b = 0, g = 0;
    uint32_t qg = 0, tg = 0, strand = 0;
    if (u < count) {
        g = a.gorder[first + u];
        b = a.gstart[g];
        const size_t e = group_end(a, g, b);
        n = (int)(e - b);
        if (n > SMALL_N) { a.counters[2] = 1; n = 0; }         // (the host splits the list by size: cannot happen)
        if (n < a.min_cnt) n = 0;
    }
    if (n) {
        const uint64_t key0 = group_word(a, g, b);
        qg = a.q_lo + (uint32_t)(key0 >> (a.tb + 1));
        tg = (uint32_t)(key0 >> 1) & ((1u << a.tb) - 1);
        strand = (uint32_t)key0 & 1u;
        const size_t g_first = strand ? b + (size_t)n - 1 : b;   // chaining order (chain_kernel)
        const long long g_step = strand ? -1 : 1;
        for (int i0 = 0; i0 < n; i0 += 4) {                   // four anchors' loads in flight together (each is a line of its own)
            int t[4], q[4], sp[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u < n ? i0 + u : n - 1;
                anchor_fields(a, (size_t)((long long)g_first + g_step * i), t[u], q[u], sp[u]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (i0 + u < n) { s_t[i0 + u][lane] = t[u]; s_q[i0 + u][lane] = q[u]; s_sp[i0 + u][lane] = (uint8_t)sp[u]; }
        }
    }
    // ---- DP + best children (oracle/ava_oracle.c:chain_group) ----
    for (int i = 0; i < n; ++i) {
        const int ti = s_t[i][lane], qi = s_q[i][lane], si = s_sp[i][lane];
        int best = si, bp = -1;
        // predecessors closest first, four at a time: their twelve LDS reads are issued together (the kernel runs at five
        // waves per CU: a read per dependent step was most of its time), then judged in the specification's order
        for (int j0 = i - 1; j0 >= 0; j0 -= 4) {
            int tj[4], qj[4], fj[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 - u >= 0 ? j0 - u : 0;
                tj[u] = s_t[j][lane]; qj[u] = s_q[j][lane]; fj[u] = s_f[j][lane];
            }
            bool stop = false;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 - u;
                if (j < 0 || stop) continue;
                const int dr = ti - tj[u], dq = qi - qj[u];
                if (dq > a.max_gap) { stop = true; continue; }  // query positions only grow going back
                if (dr <= 0 || dr > a.max_gap || dq == 0) continue;
                const int dd = dr > dq ? dr - dq : dq - dr;
                if (dd > a.bw) continue;
                const int dg = dr < dq ? dr : dq;
                const int sc = dg < si ? dg : si;
                const int pen = dd ? (dd * a.k) / 100 + (ilog2_u32((uint32_t)dd) >> 1) : 0;
                const int cand = fj[u] + sc - pen;
                if (cand > best) { best = cand; bp = j; }
            }
            if (stop) break;
        }
        s_f[i][lane] = best;
        s_p[i][lane] = (uint8_t)(bp + 1);
        s_bc[i][lane] = 0;
    }
    for (int i = 0; i < n; ++i) {                               // ascending: a later child wins only with a strictly larger f
        const int p = (int)s_p[i][lane] - 1;
        if (p < 0) continue;
        const int c = (int)s_bc[p][lane] - 1;
        if (c < 0 || s_f[i][lane] > s_f[c][lane]) s_bc[p][lane] = (uint8_t)(i + 1);
    }
    // ---- chains -> pieces + fixed points (chain_group's tail, align_chain) ----
    auto run = [&](bool write, uint32_t slot0, uint32_t &np_all, uint32_t &nf_all) {
        np_all = nf_all = 0;
        uint32_t fcur = 2u * (uint32_t)b;                       // the group's range of the fixed-point array (chain_kernel)
        for (int s = 0; s < n; ++s) {
            const int ps = (int)s_p[s][lane] - 1;
            if (ps >= 0 && (int)s_bc[ps][lane] == s + 1) continue;          // continues its parent's chain
            int m = 0, best_len = 1, cur = s, best_f = s_f[s][lane];
            s_path[m++][lane] = (uint8_t)s;
            while (s_bc[cur][lane]) {
                cur = (int)s_bc[cur][lane] - 1;
                s_path[m++][lane] = (uint8_t)cur;
                if (s_f[cur][lane] > best_f) { best_f = s_f[cur][lane]; best_len = m; }
            }
            const int score = best_f - (ps >= 0 ? s_f[ps][lane] : 0);
            if (score < a.min_score || best_len < a.min_cnt) continue;
            bool open = false;
            int cq = 0, ct = 0;
            uint32_t np = 0, nf = 0, piece_fp0 = 0;
            auto close_piece = [&]() {
                if (write) a.pieces[slot0 + np_all + np] = Piece{qg, tg, strand, (uint32_t)s, np, fcur + piece_fp0, nf - piece_fp0, 0};
                ++np;
            };
            for (int x = 0; x < best_len; ++x) {
                const int an = s_path[x][lane];
                const int qe = s_q[an][lane] + 1, te = s_t[an][lane] + 1;
                if (!open) {                                    // a piece starts at the start of this member
                    const int sp = s_sp[an][lane];
                    int q0 = qe - sp, t0 = te - sp;
                    if (q0 < 0 || t0 < 0) { const int sh = q0 < t0 ? -q0 : -t0; q0 += sh; t0 += sh; }
                    if (q0 < 0) q0 = 0;
                    if (t0 < 0) t0 = 0;
                    if (!block_ok(q0, t0, qe, te, a.shift_max)) continue;
                    piece_fp0 = nf;
                    if (write) { a.fps[fcur + nf] = FixPt{(uint32_t)q0, (uint32_t)t0}; a.fps[fcur + nf + 1] = FixPt{(uint32_t)qe, (uint32_t)te}; }
                    nf += 2;
                    cq = qe; ct = te; open = true;
                    continue;
                }
                if (!((qe - cq >= BLOCK_MIN && te - ct >= BLOCK_MIN) || x == best_len - 1)) continue;
                if (qe <= cq || te <= ct) continue;
                if (block_ok(cq, ct, qe, te, a.shift_max)) {
                    if (write) a.fps[fcur + nf] = FixPt{(uint32_t)qe, (uint32_t)te};
                    ++nf;
                    cq = qe; ct = te;
                } else {                                        // split: close here, reopen at this member
                    close_piece();
                    open = false;
                    --x;
                }
            }
            if (open) close_piece();
            fcur += nf;
            np_all += np; nf_all += nf;
        }
    };
    uint32_t np_mine = 0, nf_mine = 0;
    run(false, 0, np_mine, nf_mine);
    uint32_t incl = np_mine;                                    // inclusive prefix sum over the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = (uint32_t)__shfl_up((int)incl, o, 64);
        if (lane >= o) incl += v;
    }
    const uint32_t total = (uint32_t)__shfl((int)incl, 63, 64);
    uint32_t nf_wave = nf_mine;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) nf_wave += (uint32_t)__shfl_xor((int)nf_wave, o, 64);
    if (!total) return;                                         // (uniform)
    uint32_t slot = 0;
    if (lane == 0) {
        // pieces and fixed points in ONE 64-bit add on the counter pair: a short-read call has a million of these waves a step,
        // and returning atomics on one address are handed out one after the other (two per wave were most of the kernel)
        const unsigned long long old = atomicAdd((unsigned long long *)a.counters, (unsigned long long)total | (unsigned long long)nf_wave << 32);
        slot = (uint32_t)old;
    }
    slot = (uint32_t)__builtin_amdgcn_readfirstlane((int)slot);
    if ((unsigned long long)slot + total > a.cap_pieces) { if (lane == 0) a.counters[2] = 1; return; }
    if (np_mine) run(true, slot + incl - np_mine, np_mine, nf_mine);
}

// ---------------------------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------------------------
void build_index(const DevSketch &tsk, const uint32_t *d_chunk_of_t, const uint32_t *d_rank_t, uint32_t n_chunks,
                 uint64_t n_names, const hlmi_ava_opts &o, DevIndex &ix) {
    const size_t n = tsk.n;
    ix.n = n;
    ix.pair_once = o.pair_once;
    ix.key.alloc(n ? n : 1);
    ix.y.alloc(n ? n : 1);
    ix.occ.alloc(n ? n : 1);
    std::vector<uint32_t> mid(n_chunks, (uint32_t)o.min_mid_occ);
    if (n) {
        hipLaunchKernelGGL(split_mz_kernel, grid1(n), dim3(WG), 0, stream(), tsk.mz.p, n, ix.key.p, ix.y.p);
        sort_pairs_u64_u64(ix.key.p, ix.y.p, n, 0, 2 * o.k);
        DBuf<uint8_t> head(n);
        hipLaunchKernelGGL(index_head_kernel, grid1(n), dim3(WG), 0, stream(), ix.key.p, ix.y.p, d_chunk_of_t, n, head.p);
        DBuf<uint32_t> run_start(n);
        const size_t n_runs = select_flagged_indices(head.p, run_start.p, n);
        hipLaunchKernelGGL(index_occ_kernel, grid1(n), dim3(WG), 0, stream(), run_start.p, n_runs, n, ix.occ.p);
        DBuf<uint32_t> hist((size_t)n_chunks * HB);
        hist.zero();
        hipLaunchKernelGGL(index_hist_kernel, grid1(n_runs), dim3(WG), 0, stream(), run_start.p, n_runs, n, ix.y.p,
                           d_chunk_of_t, hist.p);
        HIP_CHECK(hipGetLastError());
        std::vector<uint32_t> h = hist.download();
        DBuf<uint64_t> exact_keys;
        for (uint32_t c = 0; c < n_chunks && o.mid_occ_frac > 0; ++c) {   // frac <= 0: fixed cut-off (-f INT)
            uint64_t nd = 0;
            for (int v = 0; v < HB; ++v) nd += h[(size_t)c * HB + v];
            if (!nd) continue;
            uint64_t q = (uint64_t)(uint32_t)((1.0 - o.mid_occ_frac) * (double)nd);
            if (q >= nd) q = nd - 1;
            uint64_t cum = 0;
            int v = 0;
            for (; v < HB; ++v) { cum += h[(size_t)c * HB + v]; if (cum > q) break; }
            if (v >= HB - 1) {     // beyond the histogram (minimap2 only clamps its cut-off, it never gives up): exact quantile
                if (!exact_keys.n) {
                    exact_keys.alloc(n_runs);
                    hipLaunchKernelGGL(run_len_key_kernel, grid1(n_runs), dim3(WG), 0, stream(), run_start.p, n_runs, n, ix.y.p,
                                       d_chunk_of_t, exact_keys.p);
                    sort_keys_u64(exact_keys, n_runs, 0, 64);
                    HIP_CHECK(hipGetLastError());
                }
                uint64_t before = 0;               // runs of the chunks ahead of c
                for (uint32_t c2 = 0; c2 < c; ++c2) for (int b = 0; b < HB; ++b) before += h[(size_t)c2 * HB + b];
                v = (int)std::min<uint64_t>((uint32_t)download_one(exact_keys.p + before + q), (uint64_t)MAX_MID_OCC);
                stat_add("index_exact_quantiles", 1);
            }
            int t = v + 1;
            if (t > (int)mid[c]) mid[c] = (uint32_t)t;
            if (mid[c] > (uint32_t)MAX_MID_OCC) mid[c] = MAX_MID_OCC;
        }
    }
    ix.mid_occ.upload(mid);
    ix.rk.alloc(n ? n : 1);
    // buckets: about one per 16 entries, never more than the key has bits
    ix.bucket_bits = std::min(2 * o.k, std::max(1, bits_for(n >> 4)));
    ix.bucket_shift = 2 * o.k - ix.bucket_bits;
    const size_t nb = (size_t)1 << ix.bucket_bits;
    ix.bucket.alloc(nb + 1);
    if (n) hipLaunchKernelGGL(index_rank_kernel, grid1(n), dim3(WG), 0, stream(), ix.y.p, ix.occ.p, ix.mid_occ.p, d_chunk_of_t,
                              d_rank_t, n, ix.rk.p);
    if (n > 1) {
        // order the entries of every key by their rank word (too frequent first, then ascending name rank), entries of
        // one target stay in position order: two stable sorts, by rank word and then by key again.  A query's partners
        // are then a suffix of the key's run (seed_kernel<count>).
        if (n >= (1ull << 32)) fail(HLMI_EINVAL, "index larger than 2^32 entries");
        DBuf<uint32_t> perm(n), rkey(n);
        DBuf<uint64_t> key2(n);
        hipLaunchKernelGGL(iota_u32_kernel, grid1(n), dim3(WG), 0, stream(), perm.p, n);
        HIP_CHECK(hipMemcpyAsync(rkey.p, ix.rk.p, n * 4, hipMemcpyDeviceToDevice, stream()));
        sort_pairs_u32_u32(rkey.p, perm.p, n, 0, 32);
        hipLaunchKernelGGL(gather_key_kernel, grid1(n), dim3(WG), 0, stream(), ix.key.p, perm.p, n, key2.p);
        sort_pairs_u64_u32(key2, perm, n, 0, 2 * o.k);
        DBuf<uint64_t> y2(n);
        hipLaunchKernelGGL(gather_entry_kernel, grid1(n), dim3(WG), 0, stream(), ix.y.p, ix.rk.p, perm.p, n, y2.p, rkey.p);
        std::swap(ix.y, y2);
        std::swap(ix.rk, rkey);
    }
    ix.rank_bits = bits_for(n_names + 1);
    if (2 * o.k + ix.rank_bits > 64 || hook("HLMI_NO_RANK_WORD")) ix.rank_bits = 0;      // (the variable: test hook)
    ix.ck.alloc(n && ix.rank_bits ? n : 1);
    if (n && ix.rank_bits)
        hipLaunchKernelGGL(compose_ck_kernel, grid1(n), dim3(WG), 0, stream(), ix.key.p, ix.rk.p, n, ix.rank_bits, ix.ck.p);
    hipLaunchKernelGGL(index_bucket_kernel, grid1(nb + 1), dim3(WG), 0, stream(), ix.key.p, n, ix.bucket_shift, nb, ix.bucket.p);
    HIP_CHECK(hipGetLastError());
    sync();
}

static SeedArgs make_seed_args(const AvaInput &in, const DevIndex &ix, const SeedPlan &plan, const uint32_t *d_qlen,
                               size_t q_lo, size_t q_hi) {
    SeedArgs sa{};
    sa.pair_once = ix.pair_once;
    sa.qmz = in.d_qmz + in.qmz_off[q_lo];
    sa.n_mz = in.qmz_off[q_hi] - in.qmz_off[q_lo];
    sa.ikey = ix.key.p; sa.iy = ix.y.p; sa.irk = ix.rk.p; sa.ick = ix.ck.p; sa.rb = ix.rank_bits; sa.bucket = ix.bucket.p; sa.bucket_shift = ix.bucket_shift;
    sa.run_lo = plan.lo.p + in.qmz_off[q_lo]; sa.run_len = plan.len.p + in.qmz_off[q_lo];
    sa.rank_q = in.d_rank_q; sa.qlen = d_qlen;
    sa.n_idx = ix.n;
    sa.q_lo = (uint32_t)q_lo;
    sa.n_ranks = hook("HLMI_SEED_NO_GUESS") ? 0u : (uint32_t)std::min<size_t>(in.n_ranks, 0xffffffffu);
    return sa;
}

void plan_seeds(const AvaInput &in, const DevIndex &ix, SeedPlan &plan) {
    const size_t nQ = in.Q->n;
    plan.per_query.assign(nQ, 0);
    const size_t n_all = in.qmz_off[nQ];
    plan.cnt.alloc(n_all ? n_all : 1);
    plan.lo.alloc(n_all ? n_all : 1);
    plan.len.alloc(n_all ? n_all : 1);
    SeedArgs sa = make_seed_args(in, ix, plan, nullptr, 0, nQ);
    if (!sa.n_mz || !ix.n) { plan.cnt.zero(); return; }
    {
        KTimer kt("seed_count");
        hipLaunchKernelGGL(seed_kernel<false>, grid1(sa.n_mz), dim3(WG), 0, stream(), sa, plan.cnt.p, nullptr, nullptr, nullptr);
    }
    HIP_CHECK(hipGetLastError());
    DBuf<uint64_t> aoff(sa.n_mz);
    exclusive_scan_u32_to_u64(plan.cnt.p, aoff.p, sa.n_mz);
    const uint64_t total = (uint64_t)download_one(aoff.p + (sa.n_mz - 1)) + download_one(plan.cnt.p + (sa.n_mz - 1));
    // cumulative anchors at the query boundaries (queries without minimizers at the very end read the total)
    std::vector<uint64_t> safe(nQ);
    for (size_t q = 0; q < nQ; ++q) safe[q] = std::min<uint64_t>(in.qmz_off[q], sa.n_mz - 1);
    DBuf<uint64_t> d_bidx, d_b(nQ);
    d_bidx.upload(safe);
    hipLaunchKernelGGL(gather_u64_at_kernel, grid1(nQ), dim3(WG), 0, stream(), aoff.p, d_bidx.p, d_b.p, nQ);
    std::vector<uint64_t> cum = d_b.download(nQ);
    cum.push_back(total);
    for (size_t q = 0; q < nQ; ++q) if (in.qmz_off[q] >= sa.n_mz) cum[q] = total;
    for (size_t q = 0; q < nQ; ++q) plan.per_query[q] = cum[q + 1] - cum[q];
}

void seed_and_chain(const AvaInput &in, const DevIndex &ix, const hlmi_ava_opts &o, const SeedPlan &plan,
                    const uint32_t *d_qlen, const uint32_t *d_tlen, size_t q_lo, size_t q_hi, ChainOut &out,
                    SeedStats &st) {
    (void)d_tlen;
    out = ChainOut();
    if (q_hi - q_lo > (1u << QL_BITS)) fail(HLMI_EINVAL, "query batch larger than %d", 1 << QL_BITS);
    if (in.T->n > (1u << T_BITS_MAX)) fail(HLMI_EINVAL, "more than %d targets in one run", 1 << T_BITS_MAX);
    uint64_t max_tlen = 1;
    for (size_t t = 0; t < in.T->n; ++t) max_tlen = std::max<uint64_t>(max_tlen, in.T->h_off[t + 1] - in.T->h_off[t]);
    if (max_tlen >= (1ull << TPOS_BITS_MAX)) fail(HLMI_EINVAL, "target longer than 2^%d bases", TPOS_BITS_MAX);
    const int pb = bits_for(max_tlen), tb = bits_for(in.T->n > 1 ? in.T->n - 1 : 1);
    uint64_t max_qlen = 1;
    for (size_t q = q_lo; q < q_hi; ++q) max_qlen = std::max<uint64_t>(max_qlen, in.Q->h_off[q + 1] - in.Q->h_off[q]);
    const int qbits = bits_for((uint64_t)(q_hi - q_lo - 1 ? q_hi - q_lo - 1 : 1)), qpb = bits_for(max_qlen);
    if (qbits + tb + 1 + pb > 64)       // (ava_device sizes its query batches so that this holds)
        fail(HLMI_EINVAL, "anchor key: %d query + %d target + %d position bits do not fit 64", qbits, tb, pb);
    if (max_qlen >= (1ull << 26)) fail(HLMI_EINVAL, "query longer than 2^26 bases (chain scores carry 6 tie-break bits in 32)");
    // one 64-bit word per anchor when everything fits: the sort moves half the bytes and needs no value array
    // (HLMI_ANCHOR_PAIRS=1 forces the key + value form: test hook for the path wide inputs take)
    // else, when the word holds everything but the (target, strand) bits, those go to a 2- or 4-byte key of their own: the
    // sort moves 10 or 12 bytes per anchor instead of 16 (HLMI_ANCHOR_SPLIT=1 forces this form, test hook)
    const bool pairs = hook("HLMI_ANCHOR_PAIRS") != nullptr;
    // (the word's query position is read back through 32 bits: 24 position bits above the 8 of the span)
    const bool fits = qbits + tb + 1 + pb + qpb + 8 <= 64 && qpb <= 24 && !pairs && !hook("HLMI_ANCHOR_SPLIT");
    const bool split = !fits && !pairs && qbits + pb + qpb + 8 <= 64 && qpb <= 24;
    int vb = fits || split ? qpb + 8 : 0;
    const char *force = hook("HLMI_ANCHOR_SPLIT");         // "4": the wide key also where two bytes would do
    int sk = split ? (tb + 1 <= 16 && !(force && force[0] == '4') ? 2 : 4) : 0;
    SeedArgs sa = make_seed_args(in, ix, plan, d_qlen, q_lo, q_hi);
    sa.pb = pb; sa.tb = tb; sa.vb = vb; sa.sk = sk;
    if (!sa.n_mz || !ix.n) return;
    const uint32_t *cnt = plan.cnt.p + in.qmz_off[q_lo];
    size_t A = 0;                               // the plan knows the anchors of every query: no round trip to the device
    for (size_t q = q_lo; q < q_hi; ++q) A += plan.per_query[q];
    st.anchors += A;
    if (!A) return;
    if (A >= (1ull << 31) - 1024) fail(HLMI_EINVAL, "anchor batch too large");      // fixed points sit at 2 x anchor offsets
    HostTimer *ht_s = new HostTimer("seed_sort_phase");
    // HLMI_SEED_GROUP: the batch's anchors grouped straight out of the index by seed_group.hip (pieces of queries, partner tables in
    // LDS, merge by target) - no anchor batch in generation order, no device-wide sort, no head selection.  Measured on C3
    // (profiles/r05c_*): 108 ms per step for the 68 % of the anchors whose pieces fit their tables, i.e. about the 169 ms of
    // fill + sort + heads for all of them - so the sort path stays the default and the kernels stay behind the switch.
    GroupedAnchors ga;
    bool grouped = hook("HLMI_SEED_GROUP") && ix.pair_once && !pairs && !hook("HLMI_ANCHOR_SPLIT") && qpb <= 24 && pb + qpb + 8 <= 64 &&
                   seed_group_supported(ix, A, pb, qpb);
    if (grouped) {
        grouped = seed_group(in, ix, plan, d_qlen, q_lo, q_hi, qpb + 8, o.min_cnt, A, ga);
        if (!grouped) stat_add("seed_group_gave_up", 1);
    }
    if (grouped) stat_add("anchors_grouped_in_lds", (double)A);
    DBuf<uint64_t> akey, aval;
    DBuf<uint16_t> sk16;
    DBuf<uint32_t> sk32;
    DBuf<uint32_t> gstart;
    const void *skey = nullptr;
    const uint32_t *gsize_p = nullptr;
    size_t G = 0;
    stat_add("anchor_bytes", (double)A * (grouped ? 8.0 : sk ? 8.0 + sk : vb ? 8.0 : 16.0));
    if (grouped) {
        vb = qpb + 8; sk = 0;
        akey = std::move(ga.key);
        gstart = std::move(ga.gstart);
        gsize_p = ga.gsize.p;
        G = ga.G;
        st.groups += ga.G_all;
    } else {
    DBuf<uint64_t> aoff(sa.n_mz);
    exclusive_scan_u32_to_u64(cnt, aoff.p, sa.n_mz);
    akey.alloc(A); aval.alloc(vb ? 1 : A);
    sk16.alloc(sk == 2 ? A : 0);
    sk32.alloc(sk == 4 ? A : 0);
    sa.oskey = sk == 2 ? (void *)sk16.p : (void *)sk32.p;
    {
        KTimer kt("seed_fill");
        hipLaunchKernelGGL(seed_kernel<true>, grid1(sa.n_mz), dim3(WG), 0, stream(), sa, nullptr, aoff.p, akey.p, aval.p);
    }
    HIP_CHECK(hipGetLastError());
    aoff.release();
    {
        KTimer kt("anchor_sort");
        // grouping only: anchors are generated query by query, so a stable sort on the (target, strand) bits alone
        // leaves every (target, strand, query) group contiguous and in generation order = ascending query position
        // (descending on the reverse strand); neither the position bits nor the query bits are sorted
        if (sk == 2) sort_pairs_u16_u64(sk16, akey, A, 0, 1 + tb);
        else if (sk == 4) sort_pairs_u32_u64(sk32, akey, A, 0, 1 + tb);
        else if (vb) sort_keys_u64(akey, A, vb + pb, vb + pb + 1 + tb);
        else sort_pairs_u64_u64(akey.p, aval.p, A, pb, pb + 1 + tb);
    }
    skey = sk == 2 ? (const void *)sk16.p : (const void *)sk32.p;
    gstart.alloc(A);
    G = sk ? select_run_heads_split(skey, sk, akey.p, A, vb + pb, gstart.p)
           : select_run_heads_u64(akey.p, A, vb + pb, gstart.p);
    st.groups += G;
    }
    delete ht_s;

    HostTimer ht_c("chain_phase");

    DBuf<uint32_t> counters(8);
    counters.zero();
    ChainArgs ca{};
    DBuf<uint32_t> gkey(G ? G : 1), gorder(G ? G : 1);
    size_t G_big = 0;                            // groups of more than SMALL_N anchors: the head of the size-ordered list
    if (G) {
        DBuf<unsigned long long> tally((size_t)TALLY_SLOTS * TALLY_STRIDE);
        tally.zero();
        hipLaunchKernelGGL(group_size_key_kernel, grid1(G), dim3(WG), 0, stream(), gstart.p, gsize_p, G, A, gkey.p, gorder.p, tally.p);
        sort_pairs_u32_u32(gkey.p, gorder.p, G, 0, 16);
        const std::vector<unsigned long long> ht = tally.download((size_t)TALLY_SLOTS * TALLY_STRIDE);
        unsigned long long a_small = 0;
        for (int k = 0; k < TALLY_SLOTS; ++k) { G_big += (size_t)ht[(size_t)k * TALLY_STRIDE]; a_small += ht[(size_t)k * TALLY_STRIDE + 1]; }
        if (hook("HLMI_CHAIN_NO_SMALL")) { G_big = G; a_small = 0; }      // test hook: every group through chain_kernel
        stat_add("chain_groups_small", (double)(G - G_big));
        stat_add("anchors_small_groups", (double)a_small);
        if (INSTR && hook("HLMI_GROUP_HIST")) {
            DBuf<unsigned long long> hist(64);
            hist.zero();
            hipLaunchKernelGGL(group_hist_kernel, grid1(G), dim3(WG), 0, stream(), gstart.p, gsize_p, G, A, hist.p);
            const std::vector<unsigned long long> h = hist.download(64);
            for (int c = 0; c < 32; ++c) if (h[c]) {
                char nm[48];
                snprintf(nm, sizeof nm, "group_hist_n_%02d", c); stat_add(nm, (double)h[c]);
                snprintf(nm, sizeof nm, "group_hist_a_%02d", c); stat_add(nm, (double)h[32 + c]);
            }
        }
    }
    ca.key = akey.p; ca.val = aval.p; ca.skey = skey; ca.sk = sk; ca.gstart = gstart.p; ca.gorder = gorder.p; ca.n_groups = G; ca.n_anchors = A;
    if (grouped) { ca.gsize = ga.gsize.p; ca.gq = ga.gq.p; ca.gts = ga.gts.p; }
    DBuf<int> root(A);
    DBuf<unsigned long long> peak(A);             // written by the kernel at every chain start before it is voted on
    DBuf<uint32_t> sbase(A), starts(A);
    ca.sbase = sbase.p; ca.root = root.p; ca.peak = peak.p; ca.starts = starts.p;
    ca.k = o.k; ca.max_gap = o.max_gap; ca.bw = o.bandwidth; ca.min_score = o.min_chain_score; ca.min_cnt = o.min_cnt;
    ca.shift_max = o.bandwidth == 0 ? 0 : SHIFT_MAX;
    ca.q_lo = (uint32_t)q_lo;
    ca.pb = pb; ca.tb = tb; ca.vb = vb; ca.pmask = (1ull << pb) - 1; ca.qmask = (uint32_t)((1ull << qpb) - 1);
    ca.cap_pieces = (uint32_t)std::min<size_t>(A / 2 + 1024, 0xfffffff0u);
    ca.cap_fps = (uint32_t)(2 * A + 1024);
    out.pieces.alloc(ca.cap_pieces);
    out.fps.alloc(ca.cap_fps);
    ca.pieces = out.pieces.p; ca.fps = out.fps.p; ca.counters = counters.p;
    // packed DP state: scores (< longest query + one span) must stay below 2^22, positions * 4 inside 31 bits
    const bool packed = o.bandwidth + 2 <= PEN_TAB && max_qlen + 256 < (1ull << 22) && pb <= 28 && o.max_gap < (1 << 24) &&
                        !hook("HLMI_CHAIN_UNPACKED");
    ca.n_list = G_big;
    if (G > G_big) {                              // the small groups: a lane each
        KTimer kt("chain_small");
        hipLaunchKernelGGL(chain_small_kernel, dim3((unsigned)cdiv(G - G_big, (size_t)64)), dim3(64), 0, stream(), ca, G_big, G - G_big);
        HIP_CHECK(hipGetLastError());
    }
    const dim3 block(64 * CHAIN_WAVES);
    auto grid_for = [](size_t n_list) { return dim3((unsigned)std::max<size_t>(1, cdiv(n_list, (size_t)CHAIN_GROUPS * CHAIN_WAVES))); };
    if (packed && G_big && !hook("HLMI_CHAIN_NO_DP16")) {
        // every wave first runs the 16-predecessor DP over its four groups, which also proves per group whether the
        // 64-predecessor DP of the specification would have given the same scores and predecessors (dp16_groups); the
        // chain bookkeeping of a proven group reads those, only the groups without the proof run the full DP
        DBuf<uint32_t> fp(A);
        ca.fp = fp.p;
        if (INSTR && hook("HLMI_CHAIN_DP16_CHECK")) {     // self-check: every group through the full DP, compared with the proven ones
            DBuf<uint8_t> verdict(G), ok_of_group(G);      // (indexed by list position / group: the first G_big entries are used)
            DBuf<unsigned long long> n_bad(8);
            n_bad.zero();
            ok_of_group.zero();
            hipLaunchKernelGGL(chain_dp16_kernel, dim3((unsigned)cdiv(G_big, (size_t)D16_WAVES * 4)), dim3(64 * D16_WAVES), 0, stream(), ca, fp.p,
                               verdict.p);
            hipLaunchKernelGGL(mark_proven_kernel, grid1(G_big), dim3(WG), 0, stream(), gorder.p, verdict.p, G_big, ok_of_group.p);
            ca.check_ok = ok_of_group.p; ca.check_bad = n_bad.p;
            hipLaunchKernelGGL((chain_kernel<3, 0>), grid_for(G_big), block, 0, stream(), ca);
            HIP_CHECK(hipGetLastError());
            const std::vector<unsigned long long> hb = n_bad.download(8);
            stat_add("chain_dp16_mismatches", (double)hb[0]);
            if (hb[0]) fprintf(stderr, "dp16 check: group %llu anchor %llu of %llu: dp16 f %llu p %d, full f %llu p %d\n", hb[1], hb[2], hb[3],
                               hb[4] >> 32, (int)(uint32_t)hb[4], hb[5] >> 32, (int)(uint32_t)hb[5]);
        } else {
            DBuf<unsigned long long> prof(8);
            if (INSTR && hook("HLMI_CHAIN_PROF")) { prof.zero(); ca.prof = prof.p; }
            {
                KTimer kt("chain");
                hipLaunchKernelGGL((chain_kernel<3, 2>), grid_for(G_big), block, 0, stream(), ca);
            }
            HIP_CHECK(hipGetLastError());
            sync();                                   // fp goes out of scope
            if (INSTR && ca.prof) {
                const std::vector<unsigned long long> hp = prof.download(8);
                stat_add("chain_prof_dp16_cyc", (double)hp[4]);
                stat_add("chain_prof_blocks_cyc", (double)hp[0]); stat_add("chain_prof_members_cyc", (double)hp[1]);
                stat_add("chain_prof_fixed_cyc", (double)hp[2]); stat_add("chain_prof_groups", (double)hp[3]);
            }
        }
    } else if (G_big) {
        KTimer kt("chain");
        const dim3 grid = grid_for(G_big);
        if (packed) hipLaunchKernelGGL(chain_kernel<3>, grid, block, 0, stream(), ca);
        else if (o.bandwidth + 2 > PEN_TAB) hipLaunchKernelGGL(chain_kernel<0>, grid, block, 0, stream(), ca);
        else if ((o.bandwidth * o.k) / 100 + 16 <= 255) hipLaunchKernelGGL(chain_kernel<1>, grid, block, 0, stream(), ca);
        else hipLaunchKernelGGL(chain_kernel<2>, grid, block, 0, stream(), ca);
    }
    HIP_CHECK(hipGetLastError());
    std::vector<uint32_t> hc = counters.download(8);
    if (hc[2]) fail(HLMI_ENOMEM, "chain output buffer overflowed (pieces %u/%u)", hc[0], ca.cap_pieces);
    out.n_pieces = hc[0];
    out.n_fp = hc[1];
    stat_add("chain_groups_full_dp", (double)hc[3]);

}

}  // namespace hlmi
