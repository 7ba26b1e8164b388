// seed_group.hip - S3 without a device-wide sort: the anchors of a query batch leave this kernel already grouped by
// (query, target, strand), every group contiguous and in generation order (ascending forward query position; the chain
// kernels read a reverse-strand group back to front).
//
// The anchors of ONE query are generated minimizer by minimizer (oracle/ava_oracle.c:oracle_ava, the loop over qm[x]), so a
// query's share of the batch is a contiguous range of the anchor numbering, and grouping it is a stable partition of that
// range by (target, strand).  One workgroup takes one query:
//   pass A  every wave walks its eighth of the range, 64 anchors at a time (lane = anchor: the occurrence is found by a
//           6-step search over a register window of the query's non-empty minimizers), reads the index entry, and counts it
//           under its (target, strand) key in an open-addressing table in LDS (insert by compare-and-swap; a query meets a few
//           hundred partners) - one 16-bit counter per (wave, slot);
//   scan    per slot the counters become the waves' offsets inside the group, the slot totals become the groups' places
//           inside the query's range, and every group of at least min_cnt anchors gets a record (start, size, query,
//           target << 1 | strand);
//   pass B  the same walk again; the lanes of a wave that hold the same slot are found by a match over its 11 index bits
//           (ballots), ranked by lane = generation order, and the anchor goes to its final place.
// Stable by construction: waves own ascending eighths, tiles ascend inside a wave, lanes ascend inside a tile.
// A query with more distinct partners than the table holds is done in sub-passes over the targets with t % K == j.
// Bytes: 16 per non-empty query minimizer (window records) + 2 x 8 per anchor (index entry, twice) + 8 per anchor out;
// bound by the gathers of the index entries and by vector issue (about 240 instructions per 64 anchors).
// Replaces seed_kernel<true> + the rocPRIM radix sort + the group-head selection (ava_chain.hip) for batches of long
// queries with the pair-once rule (filter_overlap_slr2.py:51: -X of the ava-pb preset).
#include <algorithm>

#include "ava_internal.h"
#include "dev_prims.h"
#include "wave_ops.h"

namespace hlmi {

namespace {
constexpr int WG = 256;
constexpr int SG_WAVES = 8, SG_WG = 64 * SG_WAVES;
constexpr int SG_TAB_BITS = 11, SG_TAB = 1 << SG_TAB_BITS;
constexpr uint32_t SG_MAXD = SG_TAB / 2;           // distinct (target, strand) keys per sub-pass
constexpr uint32_t SG_EMPTY = 0xffffffffu;

struct NzRec { uint32_t off, lo, zq, cnt; };       // a query minimizer with partners: first anchor (batch numbering), first
                                                   // index entry, qpos << 9 | span << 1 | strand, partners
static_assert(sizeof(NzRec) == 16, "NzRec is read as one 16-byte word");

inline dim3 grid1(size_t n) { return dim3(cdiv(n ? n : 1, WG)); }

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
                                                      uint32_t *q_k1, uint32_t *n_nz) {
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
    if (i + 1 == n) *n_nz = k + (c ? 1u : 0u);
}

struct SgArgs {
    const NzRec *nz;
    const uint32_t *n_nz, *q_k0, *q_k1;
    const uint64_t *iy;                 // index entries: target << 32 | pos << 1 | strand
    const uint32_t *q_a0;               // first anchor of every query of the batch (batch numbering), n_q + 1 entries
    const uint32_t *q_order;            // the batch's queries, most anchors first (the order the workgroups take them in)
    const uint32_t *qlen;               // by global query
    uint32_t q_lo;
    int vb, tb;                         // anchor word: tpos << vb | qpos << 8 | span
    uint64_t *okey;
    uint32_t *gstart, *gsize, *gq, *gts;      // group records
    uint32_t gcap;
    uint32_t *counters;                 // [0] records asked for, [1] groups of any size, [2] give-up flag, [3] extra sub-passes
    uint32_t min_cnt;
};

__device__ __forceinline__ uint32_t sg_hash(uint32_t key) { return (key * 0x9E3779B1u) >> (32 - SG_TAB_BITS); }

__global__ __launch_bounds__(SG_WG) void seed_group_kernel(SgArgs a) {
    __shared__ uint32_t s_key[SG_TAB];                      // (target << 1 | strand) of the slot
    __shared__ uint32_t s_cnt[SG_WAVES / 2][SG_TAB];        // 16-bit counter of wave w: half w & 1 of word [w >> 1][slot]
    __shared__ uint32_t s_base[SG_TAB];                     // first anchor of the slot's group
    __shared__ uint32_t s_ws[2][SG_WAVES];
    __shared__ uint32_t s_ctl[4];                           // [0] distinct keys, [1] too many, [2] first record of the sub-pass
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint32_t qi = a.q_order[blockIdx.x];
    const uint32_t a0 = a.q_a0[qi], a1 = a.q_a0[qi + 1];
    if (a0 == a1) return;
    const uint32_t ql = a.qlen[a.q_lo + qi];
    const uint32_t n_nz = *a.n_nz;
    const uint32_t T = (a1 - a0 + 63u) >> 6;
    const uint32_t t_lo = (uint32_t)((uint64_t)T * (uint32_t)w / SG_WAVES), t_hi = (uint32_t)((uint64_t)T * (uint32_t)(w + 1) / SG_WAVES);
    // the record that holds the first anchor of the wave's share: last k with off <= A0
    uint32_t k_first = 0;
    if (t_lo < t_hi) {
        const uint32_t A0 = a0 + 64u * t_lo;
        uint32_t lo = a.q_k0[qi], hi = a.q_k1[qi];
        while (hi - lo > 1u) {
            const uint32_t mid = (lo + hi) >> 1;
            if (a.nz[mid].off <= A0) lo = mid; else hi = mid;
        }
        k_first = lo;
    }
    uint16_t *const c16 = (uint16_t *)&s_cnt[0][0];
    // the walk over the wave's tiles: body(active, y, zq, slot search key) per tile, the next tile's index entry in flight
    // window: the records kw .. kw + 63, one per lane (W_d = first index entry - first anchor: entry of anchor an = W_d + an)
    uint32_t W_off = 0, W_d = 0, W_zq = 0, W_end = 0, kw = 0;
    auto load_win = [&](uint32_t k) {
        kw = k;
        const uint32_t idx = k + (uint32_t)lane;
        W_off = 0xffffffffu; W_d = 0; W_zq = 0; W_end = 0xffffffffu;
        if (idx < n_nz) {
            const uint4 r = ((const uint4 *)a.nz)[idx];
            W_off = r.x; W_d = r.y - r.x; W_zq = r.z; W_end = r.x + r.w;
        }
    };
    // Window lane of every anchor of the tile that starts at A, without a search: the records that start inside the tile
    // (after A) set their bit in a 64-bit mask - an OR over the wave on DPP -, an anchor's record is the one that holds A
    // plus the set bits up to its own position.
    auto locate = [&](uint32_t A) {
        const int first = __popcll(__ballot(W_off <= A)) - 1;                 // (>= 0: the window starts at or before A)
        const uint32_t d = W_off - A;                                            // 1 .. 63: starts inside the tile
        const bool in_tile = d - 1u < 63u;
        int h_lo = in_tile && d < 32u ? (int)(1u << d) : 0, h_hi = in_tile && d >= 32u ? (int)(1u << (d - 32u)) : 0;
#define HLMI_OR_STEP(CTRL, RM) h_lo |= dpp_i32<CTRL, RM>(0, h_lo); h_hi |= dpp_i32<CTRL, RM>(0, h_hi);
        HLMI_OR_STEP(0x111, 0xf) HLMI_OR_STEP(0x112, 0xf) HLMI_OR_STEP(0x114, 0xf) HLMI_OR_STEP(0x118, 0xf)
        HLMI_OR_STEP(0x142, 0xa) HLMI_OR_STEP(0x143, 0xc)
#undef HLMI_OR_STEP
        const uint32_t H_lo = (uint32_t)__builtin_amdgcn_readlane(h_lo, 63), H_hi = (uint32_t)__builtin_amdgcn_readlane(h_hi, 63);
        const unsigned long long H = (unsigned long long)H_hi << 32 | H_lo;
        return first + __popcll(H & ((2ull << lane) - 1ull));
    };
    auto expand = [&](uint32_t tile, bool &act, uint32_t &e, uint32_t &zq) {
        const uint32_t A = a0 + 64u * tile, an = A + (uint32_t)lane;
        act = an < a1;
        const uint32_t last = (a1 - A < 64u ? a1 - A : 64u) - 1u + A;
        const uint32_t wend = (uint32_t)__builtin_amdgcn_readlane((int)W_end, 63);
        if (last >= wend) {
            // The window ends inside the tile: it moves up to the record that holds the tile's first anchor (the anchor before
            // it lay inside the window, so A <= wend, and A == wend is the first anchor of the record behind the window).  A
            // record has at least one anchor: 64 of them from that one on cover the tile.
            const int j0 = A >= wend ? 64 : __popcll(__ballot(W_off <= A)) - 1;
            load_win(kw + (uint32_t)j0);
        }
        const int j = locate(A);
        e = (uint32_t)__shfl((int)W_d, j, 64) + an;
        zq = (uint32_t)__shfl((int)W_zq, j, 64);
    };
    auto walk = [&](auto &&body) {
        if (t_lo >= t_hi) return;
        load_win(k_first);
        bool act_n; uint32_t e_n, zq_n;
        expand(t_lo, act_n, e_n, zq_n);
        uint64_t y_n = act_n ? a.iy[e_n] : 0ull;
        for (uint32_t tile = t_lo; tile < t_hi; ++tile) {
            const bool act = act_n;
            const uint32_t zq = zq_n;
            const uint64_t y = y_n;
            if (tile + 1 < t_hi) {
                expand(tile + 1, act_n, e_n, zq_n);
                y_n = act_n ? a.iy[e_n] : 0ull;
            }
            if (!body(act, y, zq)) break;
        }
    };

    uint32_t K = 1, j_sub = 0, sub_base = 0;
    for (;;) {
        // ---- clear ----
        for (int s = tid; s < SG_TAB; s += SG_WG) {
            s_key[s] = SG_EMPTY;
#pragma unroll
            for (int p = 0; p < SG_WAVES / 2; ++p) s_cnt[p][s] = 0;
        }
        if (tid < 4) s_ctl[tid] = 0;
        __syncthreads();
        // ---- pass A: count ----
        walk([&](bool act, uint64_t y, uint32_t zq) {
            (void)zq;
            if (*(volatile uint32_t *)&s_ctl[1]) return false;
            const uint32_t t = (uint32_t)(y >> 32);
            if (act && (t & (K - 1u)) == j_sub) {
                const uint32_t key = t << 1 | (((uint32_t)y ^ zq) & 1u);
                uint32_t h = sg_hash(key);
                for (;;) {                                  // (a plain read first: after a query's first tiles its keys are all there)
                    uint32_t old = s_key[h];
                    if (old == SG_EMPTY) {
                        old = atomicCAS(&s_key[h], SG_EMPTY, key);
                        if (old == SG_EMPTY) { if (atomicAdd(&s_ctl[0], 1u) >= SG_MAXD) s_ctl[1] = 1; break; }
                    }
                    if (old == key) break;
                    h = (h + 1u) & (uint32_t)(SG_TAB - 1);
                }
                const uint32_t was = atomicAdd(&s_cnt[w >> 1][h], 1u << (16 * (w & 1)));
                if (((was >> (16 * (w & 1))) & 0xffffu) == 0xffffu) a.counters[2] = 1;      // 2^16 anchors of one group in one wave's share
            }
            return true;
        });
        __syncthreads();
        if (s_ctl[1]) {                                     // too many partners for the table: halve the targets of the sub-pass
            __syncthreads();
            if (tid == 0) atomicAdd(&a.counters[3], 1u);
            K <<= 1;                                        // (K, j) -> (2 K, j), later (2 K, j + K)
            if (K > (1u << 22)) { if (tid == 0) a.counters[2] = 1; return; }
            continue;
        }
        // ---- scan: wave offsets inside every group, group places, records ----
        uint32_t tot[SG_TAB / SG_WG];
        uint32_t mine = 0, recs = 0, any = 0;
#pragma unroll
        for (int i = 0; i < SG_TAB / SG_WG; ++i) {
            const int s = tid * (SG_TAB / SG_WG) + i;
            uint32_t run = 0;
#pragma unroll
            for (int p = 0; p < SG_WAVES / 2; ++p) {
                const uint32_t word = s_cnt[p][s], c0 = word & 0xffffu, c1 = word >> 16;
                s_cnt[p][s] = (run & 0xffffu) | (run + c0) << 16;
                run += c0 + c1;
            }
            if (run >= 65536u) a.counters[2] = 1;           // a group of 2^16 anchors: the 16-bit offsets do not hold it
            tot[i] = run;
            mine += run;
            recs += run >= a.min_cnt && run ? 1u : 0u;
            any += run ? 1u : 0u;
        }
        const uint32_t incl = wave_prefix_sum_incl_dpp(mine), rincl = wave_prefix_sum_incl_dpp(recs);
        uint32_t any_w = any;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) any_w += (uint32_t)__shfl_xor((int)any_w, o, 64);
        if (lane == 63) { s_ws[0][w] = incl; s_ws[1][w] = rincl; }
        if (lane == 0 && any_w) atomicAdd(&s_ctl[3], any_w);
        __syncthreads();
        uint32_t before = 0, rbefore = 0, total = 0, rtotal = 0;
#pragma unroll
        for (int p = 0; p < SG_WAVES; ++p) {
            const uint32_t v = s_ws[0][p], r = s_ws[1][p];
            if (p < w) { before += v; rbefore += r; }
            total += v; rtotal += r;
        }
        if (tid == 0) {
            s_ctl[2] = rtotal ? atomicAdd(&a.counters[0], rtotal) : 0u;
            atomicAdd(&a.counters[1], s_ctl[3]);
        }
        __syncthreads();
        {
            uint32_t at = a0 + sub_base + before + incl - mine;
            uint32_t g = s_ctl[2] + rbefore + rincl - recs;
#pragma unroll
            for (int i = 0; i < SG_TAB / SG_WG; ++i) {
                const int s = tid * (SG_TAB / SG_WG) + i;
                s_base[s] = at;
                if (tot[i] >= a.min_cnt && tot[i]) {
                    if (g < a.gcap) { a.gstart[g] = at; a.gsize[g] = tot[i]; a.gq[g] = qi; a.gts[g] = s_key[s]; }
                    ++g;
                }
                at += tot[i];
            }
        }
        __syncthreads();
        // ---- pass B: place ----
        walk([&](bool act, uint64_t y, uint32_t zq) {
            const uint32_t t = (uint32_t)(y >> 32);
            const uint32_t strand = ((uint32_t)y ^ zq) & 1u;
            const bool in = act && (t & (K - 1u)) == j_sub;
            uint32_t h = 0;
            if (in) {
                const uint32_t key = t << 1 | strand;
                h = sg_hash(key);
                while (s_key[h] != key) h = (h + 1u) & (uint32_t)(SG_TAB - 1);
            }
            unsigned long long m = __ballot(in);
#pragma unroll
            for (int b = 0; b < SG_TAB_BITS; ++b) {
                const bool bit = (h >> b) & 1u;
                const unsigned long long bb = __ballot(in && bit);
                m &= bit ? bb : ~bb;
            }
            if (!in) m = 0;
            const uint32_t rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            uint32_t old = 0, base = 0;
            const int idx = (((w >> 1) * SG_TAB + (int)h) << 1) | (w & 1);
            if (in) { old = c16[idx]; base = s_base[h]; }   // (all lanes of a slot read the counter before its first lane moves it on)
            if (in && rank == 0) c16[idx] = (uint16_t)(old + (uint32_t)__popcll(m));
            if (in) {
                const uint32_t qpos = zq >> 9, span = (zq >> 1) & 0xffu, tpos = (uint32_t)y >> 1;
                const uint32_t qp = strand ? ql - (qpos + 1u - span) - 1u : qpos;
                a.okey[base + old + rank] = (uint64_t)tpos << a.vb | (uint64_t)qp << 8 | span;
            }
            return true;
        });
        __syncthreads();
        sub_base += total;
        // next sub-pass: climb while (K, j) is a right child, then step to the right sibling
        while (K > 1u && (j_sub & (K >> 1))) { j_sub -= K >> 1; K >>= 1; }
        if (K == 1u) break;
        j_sub += K >> 1;
    }
}

}  // namespace

bool seed_group_supported(size_t n_queries, uint64_t anchors, uint64_t max_per_query, int qpb) {
    if (qpb > 22) return false;                              // qpos << 9 | span << 1 | strand in 32 bits
    (void)max_per_query;                                     // (a wave's 16-bit counters are watched by the kernel, not bounded here)
    if (anchors >= (1ull << 31)) return false;
    (void)n_queries;
    return true;
}

bool seed_group(const AvaInput &in, const DevIndex &ix, const SeedPlan &plan, const uint32_t *d_qlen, size_t q_lo, size_t q_hi,
                int vb, int tb, int min_cnt, size_t A, GroupedAnchors &out) {
    const size_t m0 = in.qmz_off[q_lo], n_mz = in.qmz_off[q_hi] - m0, nq = q_hi - q_lo;
    const uint32_t *cnt = plan.cnt.p + m0, *lo = plan.lo.p + m0;
    // anchors before every minimizer / query of the batch
    DBuf<uint32_t> aoff(n_mz);
    exclusive_scan_u32(cnt, aoff.p, n_mz);
    std::vector<uint32_t> h_a0(nq + 1, 0);
    for (size_t q = 0; q < nq; ++q) h_a0[q + 1] = h_a0[q] + (uint32_t)plan.per_query[q_lo + q];
    DBuf<uint32_t> q_a0, q_order;
    q_a0.upload(h_a0);
    {   // the longest queries first: a workgroup's time grows with its query's anchors, the launch should end on short ones
        std::vector<uint32_t> ord(nq);
        for (size_t q = 0; q < nq; ++q) ord[q] = (uint32_t)q;
        std::stable_sort(ord.begin(), ord.end(), [&](uint32_t x, uint32_t y) { return plan.per_query[q_lo + x] > plan.per_query[q_lo + y]; });
        q_order.upload(ord);
    }
    // compacted records of the minimizers that have partners
    const size_t nw = (n_mz + 63) / 64;
    DBuf<uint32_t> wcnt(nw), woff(nw), q_k0(nq), q_k1(nq), n_nz(1);
    q_k0.zero(); q_k1.zero(); n_nz.zero();
    DBuf<NzRec> nz(n_mz);
    {
        KTimer kt("seed_group_prep");
        hipLaunchKernelGGL(nz_count_kernel, grid1(n_mz), dim3(WG), 0, stream(), cnt, n_mz, wcnt.p);
        exclusive_scan_u32(wcnt.p, woff.p, nw);
        hipLaunchKernelGGL(nz_fill_kernel, grid1(n_mz), dim3(WG), 0, stream(), in.d_qmz + m0, cnt, lo, aoff.p, n_mz, woff.p, (uint32_t)q_lo,
                           nz.p, q_k0.p, q_k1.p, n_nz.p);
    }
    HIP_CHECK(hipGetLastError());
    out.key.alloc(A);
    size_t gcap = std::max<size_t>((size_t)1 << 20, A / 64);
    DBuf<uint32_t> counters(4);
    for (int attempt = 0; attempt < 2; ++attempt) {
        out.gstart.alloc(gcap); out.gsize.alloc(gcap); out.gq.alloc(gcap); out.gts.alloc(gcap);
        counters.zero();
        SgArgs sa{};
        sa.nz = nz.p; sa.n_nz = n_nz.p; sa.q_k0 = q_k0.p; sa.q_k1 = q_k1.p;
        sa.iy = ix.y.p; sa.q_a0 = q_a0.p; sa.q_order = q_order.p; sa.qlen = d_qlen; sa.q_lo = (uint32_t)q_lo;
        sa.vb = vb; sa.tb = tb; sa.okey = out.key.p;
        sa.gstart = out.gstart.p; sa.gsize = out.gsize.p; sa.gq = out.gq.p; sa.gts = out.gts.p; sa.gcap = (uint32_t)gcap;
        sa.counters = counters.p; sa.min_cnt = (uint32_t)min_cnt;
        {
            KTimer kt("seed_group");
            hipLaunchKernelGGL(seed_group_kernel, dim3((unsigned)nq), dim3(SG_WG), 0, stream(), sa);
        }
        HIP_CHECK(hipGetLastError());
        const std::vector<uint32_t> hc = counters.download(4);
        if (hc[2]) return false;                             // a group the 16-bit offsets do not hold: the sort path takes the batch
        out.G = hc[0]; out.G_all = hc[1];
        stat_add("seed_group_split_passes", (double)hc[3]);
        if (out.G <= gcap) return true;
        gcap = out.G;                                        // more records than provided for: once more with room for all
    }
    return false;
}

}  // namespace hlmi
