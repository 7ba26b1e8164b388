// seed_group.hip - S3 without a device-wide sort: the anchors of a query batch leave these kernels already grouped by
// (query, target, strand), every group contiguous and in generation order (ascending forward query position; the chain
// kernels read a reverse-strand group back to front).
//
// The anchors of ONE query are generated minimizer by minimizer (oracle/ava_oracle.c:oracle_ava, the loop over qm[x]), and
// the occurrences of one minimizer sit in the index ordered by (name rank of the target, target, position): a query's
// anchors are a matrix [minimizer][target], written row by row, and grouping them by target is its transpose.
// A workgroup takes a PIECE = (query, range of targets); a long query has several (a launch lasts as long as its longest
// workgroup).  Both kernels walk the query's minimizers a chunk at a time: up to 64 runs of index entries (<= 2 048 entries)
// copied into LDS by LDS-DMA (global_load_lds: no register, no wait - the next chunk's copy is in flight while this one is
// worked on), the records of the chunk after that already requested.
//   seed_count_kernel   lane = minimizer, every wave a quarter of each run: the entries of the piece's targets find - or
//                       make - their target's slot in an open-addressing table in LDS and bump the counter of their strand;
//                       then the counters become the groups' places inside the piece, every group of at least min_cnt
//                       anchors gets a record (start, size, query, target << 1 | strand), the table goes to memory;
//   seed_place_kernel   the table comes back (places now counted from the query's first anchor: the pieces before this
//                       one are known), then a 64-way merge by target, lane = minimizer with a pointer into its run: the
//                       smallest target any lane points at (one DPP min over the wave) is the step's target, the lanes
//                       that hold it are two ballots (one per strand), their ranks inside the ballots are generation
//                       order, the anchors go out as a run of consecutive 8-byte words per (step, strand).  Every wave
//                       merges its quarter of the piece's targets: a group belongs to one wave, the chunks follow each
//                       other, so the group's counter is simply its next free place.  Consecutive minimizers of a read
//                       meet the same partners: a step serves about half the wave.
// Stable by construction (chunks ascend, lanes ascend inside a chunk; a minimizer that meets a target more than once takes
// the step's slow path: its entries stay together, in index order).
// The index entries are read as 32-bit words dense target << (pb + 1) | position << 1 | strand (dense = place of the target
// in the order (name rank, target): one word decides the merge), built once per index (seed_group_prepare).
// A piece with more partners than its table holds gives the batch back to the sort path (ava_chain.hip).
// Bytes: 16 per non-empty query minimizer and piece + 2 x 4 per anchor and piece of its query (index entries, both kernels)
// + 8 per anchor out.
// Replaces seed_kernel<true> + the rocPRIM radix sort + the group-head selection (ava_chain.hip) for batches of long
// queries with the pair-once rule (filter_overlap_slr2.py:51: -X of the ava-pb preset).
#include <algorithm>
#include <numeric>

#include "ava_internal.h"
#include "dev_prims.h"
#include "wave_ops.h"

namespace hlmi {

namespace {
constexpr int WG = 256;
constexpr int SG_WAVES = 4, SG_WG = 64 * SG_WAVES;
constexpr int SG_PBITS = 10, SG_PAIRS = 1 << SG_PBITS;      // slots of the partner table (one per target: both strands)
constexpr uint32_t SG_MAXD = 640;                           // targets per sub-pass
constexpr int SG_CAP_BITS = 11, SG_CAP = 1 << SG_CAP_BITS;  // index entries of a chunk (its stage in LDS)
constexpr uint64_t SG_PIECE_ANCHORS = 96 << 10;             // anchors per piece (a query of average length: one piece)
constexpr int SG_QBITS = 12, SG_QPAIRS = 1 << SG_QBITS;     // slots of a query's table in seed_scan_kernel
constexpr uint32_t SG_QMAXD = 2800;                         // targets per query
constexpr uint32_t SG_EMPTY = 0xffffffffu, SG_SENT = 0xffffffffu;

struct NzRec { uint32_t off, lo, zq, cnt; };       // a query minimizer with partners: first anchor (batch numbering), first
                                                   // index entry, qpos << 9 | span << 1 | strand, partners
static_assert(sizeof(NzRec) == 16, "NzRec is read as one 16-byte word");
static_assert(SG_MAXD + SG_WG < SG_PAIRS, "the partner table keeps a free slot while insertions are in flight");

inline dim3 grid1(size_t n) { return dim3(cdiv(n ? n : 1, WG)); }

// ---- once per index: 32-bit entries ------------------------------------------------------------------------------------
__global__ __launch_bounds__(WG) void y32_kernel(const uint64_t *y, const uint32_t *dense_of_t, size_t n, int sh, uint32_t *y32) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) y32[i] = dense_of_t[y[i] >> 32] << sh | ((uint32_t)y[i] & ((1u << sh) - 1u));
}

// ---- the batch's non-empty minimizers, compacted ---------------------------------------------------------------------
__global__ __launch_bounds__(WG) void nz_count_kernel(const uint32_t *cnt, size_t n, uint32_t *wcnt) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const unsigned long long m = __ballot(i < n && cnt[i] != 0);
    if ((threadIdx.x & 63) == 0 && (i >> 6) < (n + 63) / 64) wcnt[i >> 6] = (uint32_t)__popcll(m);
}
// q_k0 / q_k1: per query of the batch the range of its records (written by its first / last minimizer; a query without
// minimizers keeps 0, 0 and has no anchors)
__global__ __launch_bounds__(WG) void nz_fill_kernel(const Mz *qmz, const uint32_t *cnt, const uint32_t *lo, const uint32_t *aoff,
                                                      size_t n, const uint32_t *woff, uint32_t q_lo, NzRec *nz, uint32_t *q_k0,
                                                      uint32_t *q_k1) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    uint32_t c = 0;
    Mz z{0, 0};
    if (i < n) { c = cnt[i]; z = qmz[i]; }
    const unsigned long long m = __ballot(c != 0);
    if (i >= n) return;
    const uint32_t k = woff[i >> 6] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    if (c) nz[k] = NzRec{aoff[i], lo[i], ((uint32_t)z.y >> 1) << 9 | (uint32_t)(z.x & 0xff) << 1 | ((uint32_t)z.y & 1u), c};
    const uint32_t q = (uint32_t)(z.y >> 32);
    if (i == 0 || (uint32_t)(qmz[i - 1].y >> 32) != q) q_k0[q - q_lo] = k;
    if (i + 1 == n || (uint32_t)(qmz[i + 1].y >> 32) != q) q_k1[q - q_lo] = k + (c ? 1u : 0u);
}

struct SgArgs {
    const NzRec *nz;
    const uint32_t *q_k0, *q_k1;
    const uint32_t *ye;                 // index entries: dense target << sh | position << 1 | strand
    const uint32_t *t_of_dense;
    uint32_t n_dense;                   // targets of the index
    const uint32_t *q_a0;               // first anchor of every query of the batch (batch numbering), n_q + 1 entries
    const uint32_t *q_dense;            // per query of the batch: targets that do not rank behind it (its partners' dense numbers start there)
    const uint32_t *piece_q, *piece_ir; // pieces, query-major: query inside the batch, piece of the query << 16 | pieces of it
    const uint32_t *q_list, *q_pid0;    // the queries that have anchors, the first piece of each (n_live + 1 entries)
    const uint32_t *qlen;               // by global query
    uint32_t q_lo;
    int sh, vb;                         // sh = position bits + 1; anchor word: tpos << vb | qpos << 8 | span
    uint64_t *okey;
    uint32_t *tab_key, *tab_pos;        // per piece: the table (SG_PAIRS keys; 2 SG_PAIRS anchor counts, after seed_scan_kernel places)
    uint32_t *gstart, *gsize, *gq, *gts;      // group records
    uint32_t gcap;
    uint32_t *counters;                 // [0] records asked for, [1] groups of any size, [2] give-up flag
    uint32_t min_cnt;
};

__device__ __forceinline__ uint32_t sg_hash(uint32_t d) { return (d * 0x9E3779B1u) >> (32 - SG_PBITS); }

struct SgShared {
    uint32_t key[SG_PAIRS];             // dense target of the slot
    uint32_t pos[2 * SG_PAIRS];         // per slot and strand: anchors counted / next free place of the group
    uint32_t stage[2][SG_CAP];          // two chunks: the runs of their minimizers one after the other
    uint32_t r_lo[2][64], r_cnt[2][64], r_off[2][64], r_zq[2][64];    // the chunks' minimizers: first entry in the index, entries, place in the stage
    uint32_t n_rec[2];
    uint32_t ws[2][SG_WAVES];
    uint32_t ctl[4];                    // [0] targets in the table, [1] too many, [2] first record of the piece, [3] groups
};

// The chunks of a query, one after the other: process(buffer, minimizers in it) is called by the whole workgroup for each.
template <typename F>
__device__ __forceinline__ void sg_chunks(const SgArgs &a, SgShared &S, int tid, uint32_t k0, uint32_t k1, F &&process) {
    const int lane = tid & 63, w = tid >> 6;
    uint32_t kc = k0, part = 0;                                       // wave 0: next record, entries of it already taken
    uint4 rec = make_uint4(0, 0, 0, 0);                               // wave 0: record kc + lane, requested a chunk ahead
    auto fetch = [&]() {
        rec = make_uint4(0, 0, 0, 0);
        if (kc + (uint32_t)lane < k1) rec = ((const uint4 *)a.nz)[kc + (uint32_t)lane];
    };
    // as many of the next 64 minimizers as fit the stage (a run longer than the stage goes in pieces)
    auto describe = [&](int b) {
        uint32_t lo = rec.y, cnt = rec.w;
        if (lane == 0 && cnt) { lo += part; cnt -= part; }
        const uint32_t incl = wave_prefix_sum_incl_dpp(cnt);
        const uint32_t n_take = (uint32_t)__popcll(__ballot(cnt != 0 && incl <= (uint32_t)SG_CAP));      // (a prefix of the lanes: incl grows)
        const bool any = __ballot(cnt != 0) != 0;
        if (!any) { if (lane == 0) S.n_rec[b] = 0; return; }
        if (!n_take) {                                                // the first run alone is longer than the stage: a piece of it
            if (lane == 0) { S.r_lo[b][0] = lo; S.r_cnt[b][0] = SG_CAP; S.r_off[b][0] = 0; S.r_zq[b][0] = rec.z; S.n_rec[b] = 1; }
            part += SG_CAP;
            return;                                                   // (the same records again, lane 0 further into its run)
        }
        if ((uint32_t)lane < n_take) { S.r_lo[b][lane] = lo; S.r_cnt[b][lane] = cnt; S.r_off[b][lane] = incl - cnt; S.r_zq[b][lane] = rec.z; }
        if (lane == 0) S.n_rec[b] = n_take;
        kc += n_take; part = 0;
        fetch();
    };
    auto issue = [&](int b) {                                         // the chunk's runs into its stage, a wave per run, by LDS-DMA
        const uint32_t n = S.n_rec[b];
        for (uint32_t r = (uint32_t)w; r < n; r += SG_WAVES) {
            const uint32_t lo = S.r_lo[b][r], cnt = S.r_cnt[b][r], off = S.r_off[b][r];
            for (uint32_t i0 = 0; i0 < cnt; i0 += 64)
                if (i0 + (uint32_t)lane < cnt)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(a.ye + lo + i0 + (uint32_t)lane),
                                                     (__attribute__((address_space(3))) void *)(&S.stage[b][off + i0]), 4, 0, 0);
        }
    };
    if (w == 0) { fetch(); describe(0); }
    __syncthreads();
    issue(0);
    for (int b = 0, round = 0; round < (1 << 26); b ^= 1, ++round) {      // (a piece has far fewer chunks: every loop of these kernels is bounded)
        if (w == 0) describe(b ^ 1);
        __syncthreads();                                              // the chunk's copy has landed (every wave waited for its own), the next one is described
        const uint32_t n = S.n_rec[b];
        if (!n) break;
        issue(b ^ 1);
        process(b, n);
        // stage b and its description are free again.  A raw barrier behind the LDS traffic only: __syncthreads() would wait for
        // the copy of the next chunk, which has the whole of the next round to land (the barrier at the top of the loop waits for it)
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();
    }
}

// A piece = a range of the query's minimizers (records) that holds about SG_PIECE_ANCHORS anchors: records k0 .. k1.
struct PieceGeom { uint32_t qi, k0, k1, pid; };
__device__ __forceinline__ PieceGeom sg_piece(const SgArgs &a, uint32_t *s_kb, int tid) {
    PieceGeom g;
    g.pid = blockIdx.x;
    g.qi = a.piece_q[g.pid];
    const uint32_t ir = a.piece_ir[g.pid], i = ir >> 16, R = ir & 0xffffu;
    const uint32_t a0 = a.q_a0[g.qi], a1 = a.q_a0[g.qi + 1];
    if (tid < 2) {                                                    // the first record at or behind the i-th (i+1-th) R-th of the anchors
        const uint32_t want = a0 + (uint32_t)((uint64_t)(a1 - a0) * (i + (uint32_t)tid) / R);
        uint32_t lo = a.q_k0[g.qi], hi = a.q_k1[g.qi];
        for (int r = 0; r < 32 && lo < hi; ++r) {
            const uint32_t mid = (lo + hi) >> 1;
            if (a.nz[mid].off < want) lo = mid + 1u; else hi = mid;
        }
        s_kb[tid] = lo;
    }
    __syncthreads();
    g.k0 = s_kb[0]; g.k1 = s_kb[1];
    return g;
}

// The merge of one chunk by one wave: lane = minimizer with a pointer into its run, targets d_lo .. d_hi.  A step takes the
// smallest target any lane points at; the lanes that hold it are two ballots (one per strand); the target's two counters are
// the next free places of its two groups, the lanes store their anchors there in lane order = generation order.
__device__ __forceinline__ void sg_merge(const SgArgs &a, SgShared &S, int b, uint32_t n_rec, int lane, uint32_t d_lo, uint32_t d_hi, uint32_t ql) {
    const int sh = a.sh;
    const uint32_t pmask = (1u << (sh - 1)) - 1u;
    const unsigned long long lt = (1ull << lane) - 1ull;
    const uint32_t *stage = S.stage[b];
    uint32_t ptr = 0, end = 0, zq = 0;
    if ((uint32_t)lane < n_rec) {
        const uint32_t off = S.r_off[b][lane], cnt = S.r_cnt[b][lane];
        zq = S.r_zq[b][lane];
        end = off + cnt;
        uint32_t lo = off, hi = end;                                  // first entry of the run with target >= d_lo
        for (int r = 0; r < SG_CAP_BITS + 1 && lo < hi; ++r) {
            const uint32_t mid = (lo + hi) >> 1;
            if ((stage[mid] >> sh) < d_lo) lo = mid + 1u; else hi = mid;
        }
        ptr = lo;
    }
    auto entry = [&](uint32_t p) {
        uint32_t v = SG_SENT;
        if (p < end) { v = stage[p]; if ((v >> sh) >= d_hi) v = SG_SENT; }
        return v;
    };
    const uint32_t qpos = zq >> 9, span = (zq >> 1) & 0xffu;
    const uint32_t q_fwd = qpos << 8 | span, q_rev = (ql - (qpos + 1u - span) - 1u) << 8 | span;
    auto store = [&](uint32_t dest, uint32_t v) {
        a.okey[dest] = (uint64_t)((v >> 1) & pmask) << a.vb | (((v ^ zq) & 1u) ? q_rev : q_fwd);
    };
    uint32_t cur = entry(ptr), nxt = entry(ptr + 1u);
    // (every step takes at least one entry of the chunk: the bound is never met - and an explicit bound is what keeps this loop
    //  honest: with `for (;;)` here the build of ROCm 7.2 produced a kernel that never returned, with the bound it is correct)
    for (int steps = 0; steps <= SG_CAP; ++steps) {
        const uint32_t m = wave_min_u32_dpp(cur);
        if (m == SG_SENT) break;
        const uint32_t d = m >> sh;                                   // the step's target
        const bool match = (cur >> sh) == d;
        const unsigned long long again = __ballot(match && (nxt >> sh) == d);
        // the target's slot and its two counters, read together (uniform: every lane the same words)
        uint32_t h = sg_hash(d), at0 = 0, at1 = 0;
        for (int probes = 0;; ++probes) {
            const uint32_t k = S.key[h];
            at0 = S.pos[2u * h]; at1 = S.pos[2u * h + 1u];
            if (k == d) break;
            h = (h + 1u) & (uint32_t)(SG_PAIRS - 1);
            // (cannot happen: seed_count_kernel saw every entry and gave every target its slot; a wave never spins)
            if (probes > SG_PAIRS) { if (lane == 0) a.counters[2] = 2; return; }
        }
        uint32_t n0, n1, c_mine = 1;
        if (!again) {
            const bool s1 = (cur ^ zq) & 1u;
            const unsigned long long b0 = __ballot(match && !s1), b1 = __ballot(match && s1);
            n0 = (uint32_t)__popcll(b0); n1 = (uint32_t)__popcll(b1);
            if (match) store((s1 ? at1 : at0) + (uint32_t)__popcll((s1 ? b1 : b0) & lt), cur);
        } else {
            // a minimizer meets the target more than once: its entries stay together - per lane the entries of the target by
            // strand, places from prefix sums over the lanes
            uint32_t c0 = 0, c1 = 0;
            if (match) {
                uint32_t p = ptr, v = cur;
                do { if ((v ^ zq) & 1u) ++c1; else ++c0; ++p; v = entry(p); } while ((v >> sh) == d && p <= end);
            }
            const uint32_t x0 = wave_prefix_sum_incl_dpp(c0), x1 = wave_prefix_sum_incl_dpp(c1);
            n0 = (uint32_t)__builtin_amdgcn_readlane((int)x0, 63); n1 = (uint32_t)__builtin_amdgcn_readlane((int)x1, 63);
            c_mine = c0 + c1;
            if (match) {
                uint32_t p = ptr, v = cur, o0 = at0 + x0 - c0, o1 = at1 + x1 - c1;
                do { store((v ^ zq) & 1u ? o1++ : o0++, v); ++p; v = entry(p); } while ((v >> sh) == d && p <= end);
            }
        }
        // (the slot belongs to this wave - its share of the targets -, every lane has read the counters: lane 0 moves them on)
        if (lane == 0) { if (n0) S.pos[2u * h] = at0 + n0; if (n1) S.pos[2u * h + 1u] = at1 + n1; }
        if (match) {
            if (!again) { ++ptr; cur = nxt; }
            else { ptr += c_mine; cur = entry(ptr); }
            nxt = entry(ptr + 1u);
        }
    }
}

// the wave's share of the query's partners: they rank behind the query (pair once: strcmp(qname, tname) < 0), i.e. their dense
// numbers start at q_dense[query]; names and places on the genome have nothing to do with each other, so equal shares of the
// numbers are about equal shares of the partners
__device__ __forceinline__ void sg_share(const SgArgs &a, uint32_t qi, int w, uint32_t &d_lo, uint32_t &d_hi) {
    const uint32_t first = a.q_dense[qi], n = a.n_dense - first;
    d_lo = first + (uint32_t)((uint64_t)n * (uint32_t)w / SG_WAVES);
    d_hi = w + 1 == SG_WAVES ? a.n_dense : first + (uint32_t)((uint64_t)n * (uint32_t)(w + 1) / SG_WAVES);
    if (w == 0) d_lo = 0;
}

__global__ __launch_bounds__(SG_WG) void seed_count_kernel(SgArgs a) {
    __shared__ SgShared S;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const PieceGeom g = sg_piece(a, S.ws[0], tid);
    for (int s = tid; s < SG_PAIRS; s += SG_WG) { S.key[s] = SG_EMPTY; S.pos[2 * s] = 0; S.pos[2 * s + 1] = 0; }
    if (tid < 4) S.ctl[tid] = 0;
    __syncthreads();
    // ---- count: lane = minimizer, the wave takes every fourth entry of its run; the entry finds - or makes - the slot of its
    //      target and bumps the counter of its strand.  (Measured against counting by the merge of seed_place_kernel, two ballots
    //      and one counter update per step: 46 ms against 90 ms per C3 step - the merge is a chain of dependent LDS round trips
    //      per step, this loop is not.) ----
    const int sh = a.sh;
    sg_chunks(a, S, tid, g.k0, g.k1, [&](int b, uint32_t n) {
        if (__hip_atomic_load(&S.ctl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) return;
        if ((uint32_t)lane >= n) return;
        const uint32_t off = S.r_off[b][lane], cnt = S.r_cnt[b][lane], zq = S.r_zq[b][lane];
        for (uint32_t i = (uint32_t)w; i < cnt; i += SG_WAVES) {
            const uint32_t v = S.stage[b][off + i], d = v >> sh;
            uint32_t h = sg_hash(d);
            for (int probes = 0; probes <= SG_PAIRS; ++probes) {
                uint32_t k = S.key[h];
                if (k == SG_EMPTY) {
                    k = atomicCAS(&S.key[h], SG_EMPTY, d);
                    if (k == SG_EMPTY) { if (atomicAdd(&S.ctl[0], 1u) >= SG_MAXD) S.ctl[1] = 1; break; }
                }
                if (k == d) break;
                // (the table never fills up: SG_MAXD targets + one insertion in flight per lane of the workgroup stay
                //  below its slots, and nobody inserts once the give-up flag is seen)
                if (__hip_atomic_load(&S.ctl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
                h = (h + 1u) & (uint32_t)(SG_PAIRS - 1);
            }
            atomicAdd(&S.pos[2u * h + ((v ^ zq) & 1u)], 1u);
        }
    });
    __syncthreads();
    if (S.ctl[1]) { if (tid == 0 && !a.counters[2]) a.counters[2] = 1; return; }      // more partners than the table holds: the batch is sorted instead
    // ---- the table to memory: seed_scan_kernel puts the pieces of a query together ----
    uint32_t *tk = a.tab_key + (size_t)g.pid * SG_PAIRS, *tp = a.tab_pos + (size_t)g.pid * (2 * SG_PAIRS);
    for (int s = tid; s < SG_PAIRS; s += SG_WG) tk[s] = S.key[s];
    for (int s = tid; s < 2 * SG_PAIRS; s += SG_WG) tp[s] = S.pos[s];
}

// One workgroup per query: the tables of its pieces become ONE table (open addressing again, SG_QPAIRS slots), the totals per
// (target, strand) become the groups' places inside the query's range of the anchor array, every group of at least min_cnt
// anchors gets a record (start, size, query, target << 1 | strand), and piece after piece the counts in the pieces' tables are
// replaced by places: the group's start + what the pieces before this one put there.
struct SgScanShared {
    uint32_t key[SG_QPAIRS];
    uint32_t tot[2 * SG_QPAIRS];        // anchors, then first anchor of the group
    uint32_t run[2 * SG_QPAIRS];        // anchors of the pieces gone by
    uint32_t ws[2][SG_WAVES];
    uint32_t ctl[4];
};
__device__ __forceinline__ uint32_t sg_qhash(uint32_t d) { return (d * 0x9E3779B1u) >> (32 - SG_QBITS); }

__global__ __launch_bounds__(SG_WG) void seed_scan_kernel(SgArgs a) {
    __shared__ SgScanShared S;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (a.counters[2]) return;                                        // a piece gave up: its table never reached memory
    const uint32_t qi = a.q_list[blockIdx.x], p0 = a.q_pid0[blockIdx.x], p1 = a.q_pid0[blockIdx.x + 1];
    const uint32_t a0 = a.q_a0[qi];
    for (int s = tid; s < SG_QPAIRS; s += SG_WG) { S.key[s] = SG_EMPTY; S.tot[2 * s] = 0; S.tot[2 * s + 1] = 0; S.run[2 * s] = 0; S.run[2 * s + 1] = 0; }
    if (tid < 4) S.ctl[tid] = 0;
    __syncthreads();
    auto slot_of = [&](uint32_t d, bool insert) {
        uint32_t h = sg_qhash(d);
        for (int probes = 0; probes <= SG_QPAIRS; ++probes) {
            uint32_t k = S.key[h];
            if (insert && k == SG_EMPTY) {
                k = atomicCAS(&S.key[h], SG_EMPTY, d);
                if (k == SG_EMPTY) { if (atomicAdd(&S.ctl[0], 1u) >= SG_QMAXD) S.ctl[1] = 1; return h; }
            }
            if (k == d) return h;
            if (__hip_atomic_load(&S.ctl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
            h = (h + 1u) & (uint32_t)(SG_QPAIRS - 1);
        }
        S.ctl[1] = 1;
        return h;
    };
    for (uint32_t p = p0; p < p1; ++p) {
        const uint32_t *tk = a.tab_key + (size_t)p * SG_PAIRS, *tp = a.tab_pos + (size_t)p * (2 * SG_PAIRS);
        for (int s = tid; s < SG_PAIRS; s += SG_WG) {
            const uint32_t d = tk[s];
            if (d == SG_EMPTY) continue;
            const uint32_t h = slot_of(d, true);
            const uint32_t c0 = tp[2 * s], c1 = tp[2 * s + 1];
            if (c0) atomicAdd(&S.tot[2u * h], c0);
            if (c1) atomicAdd(&S.tot[2u * h + 1u], c1);
        }
    }
    __syncthreads();
    if (S.ctl[1]) { if (tid == 0) a.counters[2] = 3; return; }        // more partners than the query's table holds: the batch is sorted instead
    constexpr int PER = 2 * SG_QPAIRS / SG_WG;                        // (slot, strand) counters per thread
    uint32_t tot[PER];
    uint32_t mine = 0, recs = 0, any = 0;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const uint32_t n = S.tot[tid * PER + i];
        tot[i] = n;
        mine += n;
        recs += n >= a.min_cnt && n ? 1u : 0u;
        any += n ? 1u : 0u;
    }
    const uint32_t incl = wave_prefix_sum_incl_dpp(mine), rincl = wave_prefix_sum_incl_dpp(recs);
    uint32_t any_w = any;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) any_w += (uint32_t)__shfl_xor((int)any_w, o, 64);
    if (lane == 63) { S.ws[0][w] = incl; S.ws[1][w] = rincl; }
    if (lane == 0 && any_w) atomicAdd(&S.ctl[3], any_w);
    __syncthreads();
    uint32_t before = 0, rbefore = 0, rtotal = 0;
#pragma unroll
    for (int p = 0; p < SG_WAVES; ++p) {
        const uint32_t v = S.ws[0][p], r = S.ws[1][p];
        if (p < w) { before += v; rbefore += r; }
        rtotal += r;
    }
    if (tid == 0) {
        S.ctl[2] = rtotal ? atomicAdd(&a.counters[0], rtotal) : 0u;
        atomicAdd(&a.counters[1], S.ctl[3]);
    }
    __syncthreads();
    {
        uint32_t at = a0 + before + incl - mine;
        uint32_t gi = S.ctl[2] + rbefore + rincl - recs;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int c = tid * PER + i;
            S.tot[c] = at;
            if (tot[i] >= a.min_cnt && tot[i]) {
                if (gi < a.gcap) { a.gstart[gi] = at; a.gsize[gi] = tot[i]; a.gq[gi] = qi; a.gts[gi] = a.t_of_dense[S.key[c >> 1]] << 1 | (uint32_t)(c & 1); }
                ++gi;
            }
            at += tot[i];
        }
    }
    __syncthreads();
    // the pieces in order: counts -> places (a target has one slot in a piece's table: nobody else touches its running count)
    for (uint32_t p = p0; p < p1; ++p) {
        const uint32_t *tk = a.tab_key + (size_t)p * SG_PAIRS;
        uint32_t *tp = a.tab_pos + (size_t)p * (2 * SG_PAIRS);
        for (int s = tid; s < SG_PAIRS; s += SG_WG) {
            const uint32_t d = tk[s];
            if (d == SG_EMPTY) continue;
            const uint32_t h = slot_of(d, false);
            const uint32_t c0 = tp[2 * s], c1 = tp[2 * s + 1], r0 = S.run[2u * h], r1 = S.run[2u * h + 1u];
            tp[2 * s] = S.tot[2u * h] + r0; tp[2 * s + 1] = S.tot[2u * h + 1u] + r1;
            S.run[2u * h] = r0 + c0; S.run[2u * h + 1u] = r1 + c1;
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(SG_WG) void seed_place_kernel(SgArgs a) {
    __shared__ SgShared S;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const PieceGeom g = sg_piece(a, S.ws[0], tid);
    {   // ---- the piece's table back, its counters now the next free places of its groups ----
        const uint32_t *tk = a.tab_key + (size_t)g.pid * SG_PAIRS, *tp = a.tab_pos + (size_t)g.pid * (2 * SG_PAIRS);
        for (int s = tid; s < SG_PAIRS; s += SG_WG) S.key[s] = tk[s];
        for (int s = tid; s < 2 * SG_PAIRS; s += SG_WG) S.pos[s] = tp[s];
    }
    __syncthreads();
    const uint32_t ql = a.qlen[a.q_lo + g.qi];
    uint32_t d_lo, d_hi;
    sg_share(a, g.qi, w, d_lo, d_hi);
    sg_chunks(a, S, tid, g.k0, g.k1, [&](int b, uint32_t n) { sg_merge(a, S, b, n, lane, d_lo, d_hi, ql); });
}

}  // namespace

// 32-bit index entries for the merge (dense target << (pb + 1) | position << 1 | strand): built once per index, when the
// targets' count and lengths leave the top bit of the word free
void seed_group_prepare(const AvaInput &in, DevIndex &ix) {
    ix.y32_sh = 0;
    const size_t nT = in.T->n;
    if (!ix.pair_once || !ix.n || !nT) return;
    uint64_t max_tlen = 1;
    for (size_t t = 0; t < nT; ++t) max_tlen = std::max<uint64_t>(max_tlen, in.T->h_off[t + 1] - in.T->h_off[t]);
    const int pb = bits_for(max_tlen), db = bits_for(nT > 1 ? nT - 1 : 1);
    if (db + pb + 1 > 31) return;
    // dense = place of the target in the order (name rank, target): the order of a hash's entries in the index (build_index)
    DBuf<uint32_t> tmp;
    std::vector<uint32_t> rank(nT);
    HIP_CHECK(hipMemcpyAsync(rank.data(), in.d_rank_t, nT * 4, hipMemcpyDeviceToHost, stream()));
    sync();
    std::vector<uint32_t> t_of_dense(nT), dense_of_t(nT);
    std::iota(t_of_dense.begin(), t_of_dense.end(), 0u);
    std::stable_sort(t_of_dense.begin(), t_of_dense.end(), [&](uint32_t x, uint32_t y) { return rank[x] < rank[y]; });
    ix.rank_of_dense.resize(nT);
    for (size_t d = 0; d < nT; ++d) { dense_of_t[t_of_dense[d]] = (uint32_t)d; ix.rank_of_dense[d] = rank[t_of_dense[d]]; }
    ix.t_of_dense.upload(t_of_dense);
    tmp.upload(dense_of_t);
    ix.y32.alloc(ix.n);
    hipLaunchKernelGGL(y32_kernel, grid1(ix.n), dim3(WG), 0, stream(), ix.y.p, tmp.p, ix.n, pb + 1, ix.y32.p);
    HIP_CHECK(hipGetLastError());
    sync();                                                  // (tmp goes out of scope)
    ix.y32_sh = pb + 1;
}

bool seed_group_supported(const DevIndex &ix, uint64_t anchors, int pb, int qpb) {
    if (!ix.y32_sh || ix.y32_sh != pb + 1) return false;     // 32-bit index entries (seed_group_prepare)
    if (qpb > 22) return false;                              // qpos << 9 | span << 1 | strand in 32 bits
    if (anchors >= (1ull << 31)) return false;
    return true;
}

bool seed_group(const AvaInput &in, const DevIndex &ix, const SeedPlan &plan, const uint32_t *d_qlen, size_t q_lo, size_t q_hi,
                int vb, int min_cnt, size_t A, GroupedAnchors &out) {
    const size_t m0 = in.qmz_off[q_lo], n_mz = in.qmz_off[q_hi] - m0, nq = q_hi - q_lo;
    const uint32_t *cnt = plan.cnt.p + m0, *lo = plan.lo.p + m0;
    // anchors before every minimizer / query of the batch
    DBuf<uint32_t> aoff(n_mz);
    exclusive_scan_u32(cnt, aoff.p, n_mz);
    std::vector<uint32_t> h_a0(nq + 1, 0);
    for (size_t q = 0; q < nq; ++q) h_a0[q + 1] = h_a0[q] + (uint32_t)plan.per_query[q_lo + q];
    // pieces: a query's minimizers in ranges of about SG_PIECE_ANCHORS anchors (a launch lasts as long as its longest workgroup)
    std::vector<uint32_t> piece_q, piece_ir, q_list, q_pid0;
    for (size_t q = 0; q < nq; ++q) {
        const uint64_t aq = plan.per_query[q_lo + q];
        if (!aq) continue;
        const uint32_t R = (uint32_t)std::min<uint64_t>(0xffffu, (aq + SG_PIECE_ANCHORS - 1) / SG_PIECE_ANCHORS);
        q_list.push_back((uint32_t)q);
        q_pid0.push_back((uint32_t)piece_q.size());
        for (uint32_t i = 0; i < R; ++i) { piece_q.push_back((uint32_t)q); piece_ir.push_back(i << 16 | R); }
    }
    // where the partners of every query start in the dense numbering of the targets (they rank behind it)
    std::vector<uint32_t> h_rq(nq), h_qd(nq);
    HIP_CHECK(hipMemcpyAsync(h_rq.data(), in.d_rank_q + q_lo, nq * 4, hipMemcpyDeviceToHost, stream()));
    sync();
    for (size_t q = 0; q < nq; ++q)
        h_qd[q] = (uint32_t)(std::upper_bound(ix.rank_of_dense.begin(), ix.rank_of_dense.end(), h_rq[q]) - ix.rank_of_dense.begin());
    DBuf<uint32_t> d_qd;
    d_qd.upload(h_qd);
    const size_t n_pieces = piece_q.size(), n_live = q_list.size();
    q_pid0.push_back((uint32_t)n_pieces);
    DBuf<uint32_t> q_a0, d_pq, d_pir, d_ql, d_qp0;
    q_a0.upload(h_a0); d_pq.upload(piece_q); d_pir.upload(piece_ir); d_ql.upload(q_list); d_qp0.upload(q_pid0);
    // compacted records of the minimizers that have partners
    const size_t nw = (n_mz + 63) / 64;
    DBuf<uint32_t> wcnt(nw), woff(nw), q_k0(nq), q_k1(nq);
    q_k0.zero(); q_k1.zero();
    DBuf<NzRec> nz(n_mz);
    {
        KTimer kt("seed_group_prep");
        hipLaunchKernelGGL(nz_count_kernel, grid1(n_mz), dim3(WG), 0, stream(), cnt, n_mz, wcnt.p);
        exclusive_scan_u32(wcnt.p, woff.p, nw);
        hipLaunchKernelGGL(nz_fill_kernel, grid1(n_mz), dim3(WG), 0, stream(), in.d_qmz + m0, cnt, lo, aoff.p, n_mz, woff.p, (uint32_t)q_lo,
                           nz.p, q_k0.p, q_k1.p);
    }
    HIP_CHECK(hipGetLastError());
    out.key.alloc(A);
    DBuf<uint32_t> tab_key(n_pieces * SG_PAIRS), tab_pos(n_pieces * 2 * SG_PAIRS);
    const size_t gcap = std::max<size_t>((size_t)1 << 20, A / 16);      // (read sets: a record per several hundred anchors)
    DBuf<uint32_t> counters(4);
    {
        out.gstart.alloc(gcap); out.gsize.alloc(gcap); out.gq.alloc(gcap); out.gts.alloc(gcap);
        counters.zero();
        SgArgs sa{};
        sa.nz = nz.p; sa.q_k0 = q_k0.p; sa.q_k1 = q_k1.p;
        sa.ye = ix.y32.p; sa.t_of_dense = ix.t_of_dense.p; sa.n_dense = (uint32_t)in.T->n; sa.sh = ix.y32_sh;
        sa.q_a0 = q_a0.p; sa.q_dense = d_qd.p; sa.piece_q = d_pq.p; sa.piece_ir = d_pir.p; sa.q_list = d_ql.p; sa.q_pid0 = d_qp0.p; sa.qlen = d_qlen; sa.q_lo = (uint32_t)q_lo;
        sa.vb = vb; sa.okey = out.key.p;
        sa.tab_key = tab_key.p; sa.tab_pos = tab_pos.p;
        sa.gstart = out.gstart.p; sa.gsize = out.gsize.p; sa.gq = out.gq.p; sa.gts = out.gts.p; sa.gcap = (uint32_t)gcap;
        sa.counters = counters.p; sa.min_cnt = (uint32_t)min_cnt;
        {
            KTimer kt("seed_group_count");
            hipLaunchKernelGGL(seed_count_kernel, dim3((unsigned)n_pieces), dim3(SG_WG), 0, stream(), sa);
        }
        {
            KTimer kt("seed_group_scan");
            hipLaunchKernelGGL(seed_scan_kernel, dim3((unsigned)n_live), dim3(SG_WG), 0, stream(), sa);
        }
        HIP_CHECK(hipGetLastError());
        const std::vector<uint32_t> hc = counters.download(4);
        if (hc[2]) { stat_add(hc[2] == 3 ? "seed_group_query_table_full" : "seed_group_piece_table_full", 1); return false; }      // the sort path takes the batch
        out.G = hc[0]; out.G_all = hc[1];
        if (out.G > gcap) return false;                      // more records than provided for (groups of a few anchors each): likewise
        {
            KTimer kt("seed_group_place");
            hipLaunchKernelGGL(seed_place_kernel, dim3((unsigned)n_pieces), dim3(SG_WG), 0, stream(), sa);
        }
        HIP_CHECK(hipGetLastError());
        stat_add("seed_group_pieces", (double)n_pieces);
        if (counters.download(4)[2]) fail(HLMI_EHIP, "seed_place_kernel: a target without a slot");      // (and the piece tables may go out of scope)
        return true;
    }
}

}  // namespace hlmi
