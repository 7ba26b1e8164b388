// filter_stage.hip - HIP kernels for the PAF filter chain.
//
//   a4/a17  window_filter_kernel     one workgroup per 1000-row window, window state in LDS
//           (script/filter_trans_ovlp_inline_v4.py:31-85, _v3.py:39-80)
//   a2      intermediate order       only the order INSIDE an unordered pair matters downstream
//           (first row per pair, slr2:321-326 and :133-136); rows are grouped by (chunk, pair)
//           with a radix sort and each group is ordered by the reference's comparator
//           (slr2:57: numeric tlen, tstart, tend, then whole-line bytes = PafRec::tie)
//   a5      snp_count/snp_fill       CIGAR walk of the selected rows -> X-run events
//           (filter_overlap_slr2.py:289-367 long, :229-287 short)
//   a6      snp_support_kernel       v >= mc supporters and >= mc further spanning reads
//           (filter_overlap_slr2.py:370-405) -> per-pair disagreement counts
//   a7      pass2_kernel             predicates + first surviving row per pair + X digit sum
//           (filter_overlap_slr2.py:77-136,156-161)
//
// All of it is integer / byte work bounded by HBM traffic; no MFMA.
#include "filter_stage.h"

#include <memory>

#include <algorithm>

#include "dev_prims.h"
#include "wave_ops.h"

namespace hlmi {

namespace {

constexpr int WINDOW = 1000;     // filter_trans_ovlp_inline_v4.py:33
constexpr int QUERY_CAP = 60;    // filter_trans_ovlp_inline_v4.py:82
constexpr int WG = 256;

__device__ __forceinline__ uint64_t pair_key(uint32_t a, uint32_t b) {
    return a <= b ? ((uint64_t)a << 32 | b) : ((uint64_t)b << 32 | a);
}

// overhang test shared by v3/v4/pass 2 (minimap Alg. 5; v4:52-66, slr2:116-131)
__device__ __forceinline__ bool is_internal(const PafRec &r, int min_o) {
    int64_t ql = r.qlen, qs = r.qs, qe = r.qe, tl = r.tlen, ts = r.ts, te = r.te;
    if (r.flags & PF_REV) {
        int64_t s2 = tl - te, e2 = tl - ts;
        ts = s2; te = e2;
    }
    int64_t a = qs < ts ? qs : ts;
    int64_t b = (ql - qe) < (tl - te) ? (ql - qe) : (tl - te);
    int64_t overhang = a + b;
    int64_t maplen = (qe - qs) > (te - ts) ? (qe - qs) : (te - ts);
    double lim = (double)maplen * 0.8;
    double thr = (double)min_o < lim ? (double)min_o : lim;
    return (double)overhang > thr;
}

// ---------------------------------------------------------------------------------------
// a4 / a17
// ---------------------------------------------------------------------------------------
template <int VARIANT>
__global__ __launch_bounds__(WG) void window_filter_kernel(const PafRec *__restrict__ recs,
                                                           const uint64_t *__restrict__ win_start,
                                                           const uint32_t *__restrict__ win_len, int min_len,
                                                           double min_iden, int min_o, uint8_t *__restrict__ keep) {
    __shared__ uint64_t s_key[1024];
    __shared__ uint32_t s_q[1024];
    __shared__ uint8_t s_state[1024];   // bit0 candidate, bit1 internal, bit2 first of its pair
    const uint64_t base = win_start[blockIdx.x];
    const int m = (int)win_len[blockIdx.x];
    for (int i = threadIdx.x; i < 1024; i += WG) {
        uint64_t key = ~0ull;
        uint32_t q = 0;
        uint8_t st = 0;
        if (i < m) {
            PafRec r = recs[base + i];
            bool p1 = !(r.flags & PF_BAD) && !((int64_t)r.blen < (int64_t)min_len) && r.qid != r.tid;
            if (p1) p1 = !((double)r.nmatch / (double)r.blen < min_iden);
            bool internal = is_internal(r, min_o);
            bool cand = VARIANT == 4 ? (p1 && !internal) : p1;
            if (cand) key = pair_key(r.qid, r.tid);
            q = r.qid;
            st = (cand ? 1 : 0) | (internal ? 2 : 0);
        }
        s_key[i] = key;
        s_q[i] = q;
        s_state[i] = st;
    }
    __syncthreads();
    uint8_t first_bits[4];
    for (int k = 0, i = threadIdx.x; i < 1024; i += WG, ++k) {
        bool first = false;
        if (i < m && (s_state[i] & 1)) {
            first = true;
            const uint64_t key = s_key[i];
            for (int j = 0; j < i; ++j)
                if (s_key[j] == key) { first = false; break; }
        }
        first_bits[k] = first;
    }
    __syncthreads();
    for (int k = 0, i = threadIdx.x; i < 1024; i += WG, ++k)
        if (first_bits[k]) s_state[i] |= 4;
    __syncthreads();
    for (int i = threadIdx.x; i < m; i += WG) {
        bool out = false;
        if (s_state[i] & 4) {
            if (VARIANT == 4) {
                int cnt = 0;
                const uint32_t q = s_q[i];
                for (int j = 0; j < i; ++j) cnt += ((s_state[j] & 4) && s_q[j] == q) ? 1 : 0;
                out = cnt < QUERY_CAP;
            } else {
                out = !(s_state[i] & 2);
            }
        }
        keep[base + i] = out ? 1 : 0;
    }
}

// script/filter_ovlp_inline.py:12-106: per 1000-row window, rows that survive the length /
// identity / overhang tests and are not self hits compete per unordered pair; the longest (column 11, earlier
// row on ties) is kept and takes the output position of the pair's first surviving row (first_of).
__global__ __launch_bounds__(WG) void ovlp_inline_kernel(const PafRec *__restrict__ recs, const uint64_t *__restrict__ win_start,
                                                         const uint32_t *__restrict__ win_len, int min_len, double min_iden,
                                                         int o, double r, uint8_t *__restrict__ keep,
                                                         uint32_t *__restrict__ first_of) {
    __shared__ uint64_t s_key[1024];
    __shared__ uint32_t s_len[1024];
    const uint64_t base = win_start[blockIdx.x];
    const int m = (int)win_len[blockIdx.x];
    for (int i = threadIdx.x; i < 1024; i += WG) {
        uint64_t key = ~0ull;
        uint32_t len = 0;
        if (i < m) {
            const PafRec rc = recs[base + i];
            bool ok = !(rc.flags & PF_BAD) && !((int64_t)rc.blen < (int64_t)min_len);
            if (ok) ok = !((double)rc.nmatch / (double)rc.blen < min_iden);
            if (ok) {                                   // rm_intermatch: overhang > min(o, maplen * r)
                int64_t ql = rc.qlen, qs = rc.qs, qe = rc.qe, tl = rc.tlen, ts = rc.ts, te = rc.te;
                if (rc.flags & PF_REV) { const int64_t s2 = tl - te, e2 = tl - ts; ts = s2; te = e2; }
                const int64_t overhang = (qs < ts ? qs : ts) + ((ql - qe) < (tl - te) ? (ql - qe) : (tl - te));
                const int64_t maplen = (qe - qs) > (te - ts) ? (qe - qs) : (te - ts);
                const double lim = (double)maplen * r, thr = (double)o < lim ? (double)o : lim;
                ok = !((double)overhang > thr);
            }
            if (ok && rc.qid != rc.tid) { key = pair_key(rc.qid, rc.tid); len = rc.blen; }
        }
        s_key[i] = key;
        s_len[i] = len;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < m; i += WG) {
        const uint64_t key = s_key[i];
        bool best = key != ~0ull;
        int first = i;
        if (best) {
            const uint32_t len = s_len[i];
            for (int j = 0; j < m; ++j) {
                if (j == i || s_key[j] != key) continue;
                if (j < first) first = j;
                if (s_len[j] > len || (s_len[j] == len && j < i)) best = false;
            }
        }
        keep[base + i] = best ? 1 : 0;
        first_of[base + i] = (uint32_t)(base + first);
    }
}

// ---------------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------------
__global__ void iota_kernel(uint32_t *a, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) a[i] = (uint32_t)i;
}
__global__ void gather_u32_kernel(const uint32_t *src, const uint32_t *idx, uint32_t *dst, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
}
__global__ void gather_u64_kernel(const uint64_t *src, const uint32_t *idx, uint64_t *dst, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
}
__global__ void gather_rec_kernel(const PafRec *recs, const uint32_t *rows, PafRec *dst, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) dst[i] = recs[rows[i]];
}
__global__ void head_flags_kernel(const uint32_t *chunk, const uint64_t *key, uint8_t *head, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) head[i] = (i == 0 || chunk[i] != chunk[i - 1] || key[i] != key[i - 1]) ? 1 : 0;
}

inline dim3 grid1(size_t n) { return dim3(cdiv(n ? n : 1, WG)); }

// order (chunk[i], key[i]) lexicographically: returns the permutation and the sorted copies
struct SortedCK {
    DBuf<uint32_t> perm, chunk;
    DBuf<uint64_t> key;
};
// (chunk, key) -> one word chunk | key.hi | key.lo with the bit widths the data needs, so that ONE radix sort of
// cb + hb + lb bits (38 on C2 instead of 64 + 7 in two sorts) orders the rows; `bits` = OR of all keys.
__global__ void or_reduce_u64_kernel(const uint64_t *key, size_t n, unsigned long long *acc) {
    unsigned long long v = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) v |= key[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v |= __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0 && v) atomicOr(acc, v);
}
__global__ void compact_ck_kernel(const uint32_t *chunk, const uint64_t *key, size_t n, int hb, int lb, uint64_t *out) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) out[i] = ((uint64_t)chunk[i] << hb | key[i] >> 32) << lb | (key[i] & 0xffffffffull);
}
__global__ void expand_ck_kernel(const uint64_t *ck, size_t n, int hb, int lb, uint32_t *chunk, uint64_t *key) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t v = ck[i];
    chunk[i] = (uint32_t)(v >> (hb + lb));
    key[i] = ((v >> lb) & ((1ull << hb) - 1ull)) << 32 | (v & ((1ull << lb) - 1ull));
}

void sort_chunk_key(const uint32_t *d_chunk, const uint64_t *d_key, size_t n, uint32_t n_chunks, SortedCK &o) {
    o.perm.alloc(n);
    o.chunk.alloc(n);
    o.key.alloc(n);
    if (!n) return;
    hipLaunchKernelGGL(iota_kernel, grid1(n), dim3(WG), 0, stream(), o.perm.p, n);
    DBuf<unsigned long long> acc(1);
    acc.zero();
    hipLaunchKernelGGL(or_reduce_u64_kernel, dim3((unsigned)std::min<size_t>(cdiv(n, WG), 2048)), dim3(WG), 0, stream(), d_key,
                       n, acc.p);
    const uint64_t bits = acc.download(1)[0];
    const int hb = bits_for(bits >> 32), lb = bits_for(bits & 0xffffffffull), cb = bits_for(n_chunks > 1 ? n_chunks - 1 : 1);
    if (cb + hb + lb <= 64) {
        DBuf<uint64_t> ck(n);
        hipLaunchKernelGGL(compact_ck_kernel, grid1(n), dim3(WG), 0, stream(), d_chunk, d_key, n, hb, lb, ck.p);
        sort_pairs_u64_u32(ck, o.perm, n, 0, cb + hb + lb);
        hipLaunchKernelGGL(expand_ck_kernel, grid1(n), dim3(WG), 0, stream(), ck.p, n, hb, lb, o.chunk.p, o.key.p);
        return;
    }
    HIP_CHECK(hipMemcpyAsync(o.key.p, d_key, n * 8, hipMemcpyDeviceToDevice, stream()));
    sort_pairs_u64_u32(o.key.p, o.perm.p, n);
    if (n_chunks > 1) {
        hipLaunchKernelGGL(gather_u32_kernel, grid1(n), dim3(WG), 0, stream(), d_chunk, o.perm.p, o.chunk.p, n);
        sort_pairs_u32_u32(o.chunk.p, o.perm.p, n, 0, bits_for(n_chunks - 1));
        hipLaunchKernelGGL(gather_u64_kernel, grid1(n), dim3(WG), 0, stream(), d_key, o.perm.p, o.key.p, n);
    } else {
        hipLaunchKernelGGL(gather_u32_kernel, grid1(n), dim3(WG), 0, stream(), d_chunk, o.perm.p, o.chunk.p, n);
    }
}

// ---------------------------------------------------------------------------------------
// pair groups
// ---------------------------------------------------------------------------------------
__global__ void pair_keys_kernel(const PafRec *recs, const uint32_t *rows, size_t n, uint32_t *chunk, uint64_t *key) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const PafRec &r = recs[rows[i]];
    chunk[i] = r.chunk;
    key[i] = pair_key(r.qid, r.tid);
}

// ---- whole-line byte order of two rows the overlapper emitted (LC_ALL=C last resort of `sort -nk7 -k8 -k9 -k5`,
// slr2:57).  The text of such a row is a function of its fields (ava.hip:format_ava_row): name, numbers in decimal,
// strand, ..., NM:i:<blen - nmatch>, tp:A:S, cg:Z:<len><op>...  Two decimal numbers followed by the same terminator
// compare like their digit strings, and where one is a prefix of the other the terminator decides: TAB sorts below
// the digits, the CIGAR letters above them.
__device__ __forceinline__ int dec_digits(uint32_t v, uint8_t *d) {      // most significant first
    uint8_t t[10];
    int n = 0;
    do { t[n++] = (uint8_t)(v % 10u); v /= 10u; } while (v);
    for (int i = 0; i < n; ++i) d[i] = t[n - 1 - i];
    return n;
}
__device__ __forceinline__ int cmp_dec_text(uint32_t a, uint32_t b, bool term_high) {
    if (a == b) return 0;
    uint8_t da[10], db[10];
    const int na = dec_digits(a, da), nb = dec_digits(b, db);
    const int n = na < nb ? na : nb;
    for (int i = 0; i < n; ++i)
        if (da[i] != db[i]) return da[i] < db[i] ? -1 : 1;
    // one is a proper prefix of the other: its terminator meets a digit
    return ((na < nb) != term_high) ? -1 : 1;
}
__device__ __forceinline__ int op_char(uint32_t code) {      // '=' 'D' 'I' 'X'
    return code == OP_EQ ? 0x3d : code == OP_D ? 0x44 : code == OP_I ? 0x49 : code == OP_X ? 0x58 : 0x3f;
}
__device__ bool row_text_less(const PafRec &a, const PafRec &b, const uint32_t *ops) {
    int c;
    if (a.qid != b.qid) return a.qid < b.qid;                 // name ranks = strcmp order
    if ((c = cmp_dec_text(a.qlen, b.qlen, false))) return c < 0;
    if ((c = cmp_dec_text(a.qs, b.qs, false))) return c < 0;
    if ((c = cmp_dec_text(a.qe, b.qe, false))) return c < 0;
    if ((a.flags ^ b.flags) & PF_REV) return !(a.flags & PF_REV);      // '+' (0x2b) < '-' (0x2d)
    if (a.tid != b.tid) return a.tid < b.tid;
    if ((c = cmp_dec_text(a.tlen, b.tlen, false))) return c < 0;
    if ((c = cmp_dec_text(a.ts, b.ts, false))) return c < 0;
    if ((c = cmp_dec_text(a.te, b.te, false))) return c < 0;
    if ((c = cmp_dec_text(a.nmatch, b.nmatch, false))) return c < 0;
    if ((c = cmp_dec_text(a.blen, b.blen, false))) return c < 0;
    if ((c = cmp_dec_text(a.blen - a.nmatch, b.blen - b.nmatch, false))) return c < 0;
    const uint32_t n = a.cig_n < b.cig_n ? a.cig_n : b.cig_n;
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t x = ops[a.cig_off + i], y = ops[b.cig_off + i];
        if (x == y) continue;
        if ((c = cmp_dec_text(x >> 4, y >> 4, true))) return c < 0;
        return op_char(x & 15u) < op_char(y & 15u);
    }
    if (a.cig_n != b.cig_n) return a.cig_n < b.cig_n;          // the shorter line is a prefix of the longer one
    return a.tie < b.tie;                                      // identical lines
}

// the reference's intermediate order restricted to one chunk (slr2:57)
__device__ __forceinline__ bool row_less(const PafRec &a, const PafRec &b, const uint32_t *ops) {
    if (a.tlen != b.tlen) return a.tlen < b.tlen;
    if (a.ts != b.ts) return a.ts < b.ts;
    if (a.te != b.te) return a.te < b.te;
    if (a.flags & b.flags & PF_GEN) return row_text_less(a, b, ops);
    return a.tie < b.tie;
}

// one thread per pair group: order the group's rows, select the rows that feed the pile-up
__global__ void pair_order_select_kernel(const PafRec *recs, const uint32_t *ops, uint32_t *grows /* rows, grouped */,
                                         const uint32_t *seg_start, size_t n_seg, size_t n_rows, int long_mode,
                                         uint8_t *sel /* per grouped position */) {
    size_t s = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (s >= n_seg) return;
    size_t b = seg_start[s], e = (s + 1 < n_seg) ? seg_start[s + 1] : n_rows;
    for (size_t i = b + 1; i < e; ++i) {   // insertion sort, groups are tiny
        uint32_t x = grows[i];
        size_t j = i;
        while (j > b && row_less(recs[x], recs[grows[j - 1]], ops)) { grows[j] = grows[j - 1]; --j; }
        grows[j] = x;
    }
    bool taken = false;
    for (size_t i = b; i < e; ++i) {
        const PafRec &r = recs[grows[i]];
        bool s1 = false;
        if (r.qid != r.tid) {
            if (long_mode) {
                if (!(r.flags & PF_STAR) && !taken) { s1 = true; taken = true; }
            } else {
                s1 = true;
            }
        }
        sel[i] = s1;
    }
}

// ---------------------------------------------------------------------------------------
// a5: events
// ---------------------------------------------------------------------------------------
// One wavefront per row: the CIGAR is read 64 ops at a time (coalesced), the positions before each op come from
// wave prefix sums, every 'X' lane writes its own events.
constexpr uint32_t NO_PAIR = 0xffffffffu;
__device__ __forceinline__ uint32_t wave_incl_sum_u32(uint32_t v, int lane) {
    (void)lane;
    return wave_prefix_sum_incl_dpp(v);
}

// maxv[0] / maxv[1]: largest read id / read length among the selected rows (they size the event sort key)
__global__ __launch_bounds__(WG) void snp_count_kernel(const PafRec *recs, const uint32_t *ops, const uint32_t *grows,
                                                        const uint8_t *sel, size_t n, int long_mode, uint32_t *n_ev,
                                                        uint32_t *n_iv, uint32_t *maxv) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
    const size_t n_waves = ((size_t)gridDim.x * blockDim.x) >> 6;
    uint32_t max_id = 0, max_len = 0;
    for (size_t i = wave; i < n; i += n_waves) {
        uint32_t ne = 0, ni = 0;
        if (sel[i]) {
            const PafRec &r = recs[grows[i]];
            max_id = max(max_id, max(r.qid, r.tid));
            max_len = max(max_len, max(r.qlen, r.tlen));
            const uint32_t *o = ops + r.cig_off;
            for (uint32_t k0 = 0; k0 < r.cig_n; k0 += 64)
                ne += (uint32_t)__popcll(__ballot(k0 + lane < r.cig_n && (o[k0 + lane] & 15u) == OP_X));
            if (long_mode) ne *= 2;
            ni = (r.ts < r.te ? 1u : 0u) + ((long_mode && r.qs < r.qe) ? 1u : 0u);
        }
        if (lane == 0) { n_ev[i] = ne; n_iv[i] = ni; }
    }
    // (a quarter of a million waves on one counter word would serialise: only a wave that raises the maximum it
    // can see sends an atomic)
    if (lane == 0) {
        if (max_id > __hip_atomic_load(&maxv[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&maxv[0], max_id);
        if (max_len > __hip_atomic_load(&maxv[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&maxv[1], max_len);
    }
}

__global__ __launch_bounds__(WG) void snp_fill_kernel(const PafRec *recs, const uint32_t *ops, const uint32_t *grows,
                                                       const uint8_t *sel, size_t n, int long_mode, const uint32_t *ev_off,
                                                       const uint32_t *iv_off, const uint32_t *row_pair, int hb, int lb,
                                                       uint64_t *ev_ck, uint32_t *ev_pair,
                                                       uint64_t *iv_sck, uint64_t *iv_eck) {
    // sort word of (chunk, read, position): chunk | read (hb bits) | position (lb bits) - numeric order = the
    // lexicographic order of the triple
    auto ck = [&](uint32_t chunk, uint32_t id, uint32_t pos) { return ((uint64_t)chunk << hb | id) << lb | pos; };
    const int lane = threadIdx.x & 63;
    const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
    const size_t n_waves = ((size_t)gridDim.x * blockDim.x) >> 6;
    for (size_t i = wave; i < n; i += n_waves) {
        if (!sel[i]) continue;
        const PafRec r = recs[grows[i]];
        if (lane == 0) {
            uint32_t w = iv_off[i];
            if (r.ts < r.te) { iv_sck[w] = ck(r.chunk, r.tid, r.ts); iv_eck[w] = ck(r.chunk, r.tid, r.te); ++w; }
            if (long_mode && r.qs < r.qe) { iv_sck[w] = ck(r.chunk, r.qid, r.qs); iv_eck[w] = ck(r.chunk, r.qid, r.qe); }
        }
        // the pair group of this row = the pair every event of the row counts for (read and partner are the row's two
        // names); NO_PAIR when nobody will read that pair's counter
        const uint32_t pg = row_pair[i];
        const bool rev = r.flags & PF_REV;
        uint32_t p1 = rev ? r.qlen - r.qe : r.qs;   // slr2:334   (positions after the ops handled so far)
        uint32_t p2 = r.ts;                         // slr2:336
        uint32_t e = ev_off[i];
        const uint32_t *o = ops + r.cig_off;
        for (uint32_t k0 = 0; k0 < r.cig_n; k0 += 64) {
            uint32_t len = 0, code = 0;
            if (k0 + lane < r.cig_n) { const uint32_t op = o[k0 + lane]; len = op >> 4; code = op & 15u; }
            const bool isx = code == OP_X;
            const uint32_t d1 = (code == OP_EQ || code == OP_I || isx) ? len : 0u;
            const uint32_t d2 = (code == OP_EQ || code == OP_D || isx) ? len : 0u;
            const uint32_t s1 = wave_incl_sum_u32(d1, lane), s2 = wave_incl_sum_u32(d2, lane);
            const unsigned long long xm = __ballot(isx);
            if (isx) {
                const uint32_t q1 = p1 + s1, q2 = p2 + s2;              // positions after this op
                uint32_t at = e + (uint32_t)__popcll(xm & ((1ull << lane) - 1ull)) * (long_mode ? 2u : 1u);
                if (long_mode) {
                    const uint32_t qp = rev ? r.qlen - q1 + 1 : q1;    // slr2:357
                    ev_ck[at] = ck(r.chunk, r.qid, qp); ev_pair[at] = pg; ++at;
                }
                ev_ck[at] = ck(r.chunk, r.tid, q2); ev_pair[at] = pg;
            }
            p1 += (uint32_t)__shfl((int)s1, 63, 64);
            p2 += (uint32_t)__shfl((int)s2, 63, 64);
            e += (uint32_t)__popcll(xm) * (long_mode ? 2u : 1u);
        }
    }
}

// ---------------------------------------------------------------------------------------
// a6: support test per SNP key, then per-pair counts
// ---------------------------------------------------------------------------------------
// Two balanced kernels: (1) one thread per distinct SNP key decides "supported" (v >= mc supporters and >= mc
// further spanning reads) and writes the verdict over the key's event range; (2) one thread per EVENT of a
// supported key bumps the counter of its row's pair group (recorded with the event).  (A key on a deeply covered read has
// hundreds of events; a single thread walking them all was the long pole of the filter stage.)
__device__ __forceinline__ size_t lower_bound_u64(const uint64_t *v, size_t n, uint64_t k) {
    size_t lo = 0, hi = n;
    while (lo < hi) { const size_t mid = (lo + hi) >> 1; if (v[mid] < k) lo = mid + 1; else hi = mid; }
    return lo;
}
__global__ void snp_support_kernel(const uint64_t *ev_ck, const uint32_t *kseg_start, size_t n_kseg, size_t n_ev,
                                   const uint64_t *ivs_ck, const uint64_t *ive_ck, size_t n_iv, int lb, int mc,
                                   uint8_t *ev_supported) {
    size_t s = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (s >= n_kseg) return;
    size_t b = kseg_start[s], e = (s + 1 < n_kseg) ? kseg_start[s + 1] : n_ev;
    int64_t v = (int64_t)(e - b);
    if (v < mc) return;
    const uint64_t key = ev_ck[b];
    const uint64_t read0 = key >> lb << lb;                       // (chunk, read, position 0)
    // #intervals of this read with start < pos  minus  #with end <= pos   (start < end holds for all).
    // Both arrays hold the same intervals grouped by (chunk, read), so the read's block [a, b) is the same index
    // range in both: one full search for a, a short one for b (a read rarely has more than a few thousand
    // intervals in a chunk), then two searches inside the block.
    const size_t a = lower_bound_u64(ivs_ck, n_iv, read0);
    const uint64_t next_read = read0 + (1ull << lb);
    size_t span = n_iv - a < 4096 ? n_iv - a : 4096;
    if (span == 4096 && ivs_ck[a + span - 1] < next_read) span = n_iv - a;      // a very deep read: search the rest
    const size_t b_ = a + lower_bound_u64(ivs_ck + a, span, next_read);
    const int64_t n_start_lt = (int64_t)lower_bound_u64(ivs_ck + a, b_ - a, key);
    const int64_t n_end_le = (int64_t)lower_bound_u64(ive_ck + a, b_ - a, key + 1);
    int64_t con = n_start_lt - n_end_le;
    if (con - v < mc) return;
    for (size_t i = b; i < e; ++i) ev_supported[i] = 1;      // contiguous bytes; v is at most the read depth
}

// ev_pair travelled through the sort as the value of its event: a streaming read, one atomic per supported event
__global__ void snp_pair_count_kernel(const uint32_t *ev_pair_sorted, const uint8_t *ev_supported, size_t n_ev,
                                      uint32_t *pair_mut) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n_ev && ev_supported[i]) {
        const uint32_t pg = ev_pair_sorted[i];
        if (pg != NO_PAIR) atomicAdd(&pair_mut[pg], 1u);
    }
}

// ---------------------------------------------------------------------------------------
// a5 + a6 without materialised events: one workgroup per (chunk, read)
// ---------------------------------------------------------------------------------------
// A SNP key is (chunk, read, position) and everything the reference asks about it - how many selected rows put an X
// there (v), how many of the read's intervals span it (con) - is local to one read of one chunk.  So the selected rows
// are listed once under their target read and (long mode) once under their query read, the list is sorted by
// (chunk, read), and a workgroup takes one read: two 16-bit arrays over the read's positions in LDS hold first the
// interval starts / ends (prefix-summed into the coverage), then the X counts; a second walk over the same CIGARs
// bumps the pair counter of every event whose key is supported (slr2:370-405).  Nothing per event ever goes to memory:
// the sort of 2.7e9 16-byte events per C3 slice is replaced by a sort of 2.4e7 row references.
constexpr int PILE_TILE = 15360;             // positions per pass (2 x 16-bit x 15360 = 60 KiB of LDS); longer reads: several passes
constexpr int PILE_WG_HEAVY = 1024;          // reads with many rows in the chunk: one wave walks one row at a time
constexpr int PILE_LIGHT_ROWS = 16;          // reads with at most this many rows take snp_pileup_light_kernel
constexpr int PILE_EV_SLOTS = 4096;           // the event statistic is summed over this many counters (one word would
                                             // serialise millions of atomics)
constexpr uint32_t PILE_MAX_ROWS = 60000;    // rows per (chunk, read): the counters are 16 bit

// entry of row i (position in the grouped row list) under one of its reads: key = chunk << 32 | read, val = i << 1 | side
// (side 0: the read is the row's target, 1: its query).  Unselected rows get the sentinel key.
__global__ void pile_entries_kernel(const PafRec *recs, const uint32_t *grows, const uint8_t *sel, size_t m, int long_mode,
                                    uint64_t *key, uint32_t *val) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= m) return;
    uint64_t kt = ~0ull, kq = ~0ull;
    if (sel[i]) {
        const PafRec &r = recs[grows[i]];
        kt = (uint64_t)r.chunk << 32 | r.tid;
        if (long_mode) kq = (uint64_t)r.chunk << 32 | r.qid;
    }
    key[2 * i] = kt; val[2 * i] = (uint32_t)(i << 1);
    key[2 * i + 1] = kq; val[2 * i + 1] = (uint32_t)(i << 1 | 1u);
}
__global__ void pile_entry_rows_kernel(const uint32_t *ent_val, const uint32_t *grows, size_t n, uint32_t *ent_rec) {
    size_t k = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (k < n) ent_rec[k] = grows[ent_val[k] >> 1];
}
__global__ void add_one_u8_kernel(const uint8_t *in, uint8_t *out, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i] + 1;
}
__global__ void pile_seg_size_kernel(const uint32_t *seg_start, size_t n_seg, size_t n_ent, uint32_t *max_rows) {
    size_t s = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (s >= n_seg) return;
    const uint32_t n = (uint32_t)((s + 1 < n_seg ? seg_start[s + 1] : n_ent) - seg_start[s]);
    if (n > __hip_atomic_load(max_rows, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(max_rows, n);
}

// counters of 8 or 16 bits packed into 32-bit LDS words (LDS atomics are 32 bit wide; a counter never overflows into its
// neighbour: it counts rows of the segment, and the segment's class bounds those)
template <typename CT> __device__ __forceinline__ void lds_inc(uint32_t *a, uint32_t idx) {
    constexpr uint32_t per = 4 / sizeof(CT);
    atomicAdd(&a[idx / per], 1u << (8 * sizeof(CT) * (idx % per)));
}
template <typename CT> __device__ __forceinline__ uint32_t lds_get(const uint32_t *a, uint32_t idx) { return ((const CT *)a)[idx]; }

// class of every segment: 0 = fewer than 2 mc rows (no key can be supported), 1 = light, 2 = heavy
__global__ void pile_seg_class_kernel(const uint32_t *seg_start, size_t n_seg, size_t n_ent, int mc, uint8_t *cls) {
    size_t s = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (s >= n_seg) return;
    const uint32_t n = (uint32_t)((s + 1 < n_seg ? seg_start[s + 1] : n_ent) - seg_start[s]);
    cls[s] = (long long)n < 2ll * mc ? 0 : (n <= (uint32_t)PILE_LIGHT_ROWS ? 1 : 2);
}
// X events of the rows under segments nobody piles (statistics only): one wave per entry
__global__ __launch_bounds__(WG) void pile_tiny_events_kernel(const PafRec *recs, const uint32_t *ops, const uint32_t *grows,
                                                               const uint32_t *ent_val, const uint32_t *seg_start, const uint32_t *list,
                                                               size_t n_list, size_t n_seg, size_t n_ent, unsigned long long *n_events) {
    const int lane = threadIdx.x & 63;
    const size_t w = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
    if (w >= n_list) return;
    const size_t seg = list[w];
    const size_t b = seg_start[seg], e = seg + 1 < n_seg ? seg_start[seg + 1] : n_ent;
    unsigned long long ev = 0;
    for (size_t k = b; k < e; ++k) {
        const PafRec &r = recs[grows[ent_val[k] >> 1]];
        const uint32_t *o = ops + r.cig_off;
        for (uint32_t k0 = 0; k0 < r.cig_n; k0 += 64)
            ev += (unsigned long long)__popcll(__ballot(k0 + lane < r.cig_n && (o[k0 + lane] & 15u) == OP_X));
    }
    if (lane == 0 && ev) atomicAdd(&n_events[w & (PILE_EV_SLOTS - 1)], ev);
}

template <int PILE_WG, typename CT>
__global__ __launch_bounds__(PILE_WG) __attribute__((amdgpu_waves_per_eu(8, 8))) void snp_pileup_kernel(const PafRec *recs, const uint32_t *ops, const uint32_t *ent_rec,
                                                              const uint32_t *row_pair, const uint64_t *ent_key, const uint32_t *ent_val,
                                                              const uint32_t *seg_start, const uint32_t *list, size_t n_seg, size_t n_ent,
                                                              int long_mode, int mc, uint32_t *pair_mut, unsigned long long *n_events) {
    constexpr uint32_t PER = 4 / sizeof(CT);                          // counters per LDS word
    __shared__ uint32_t sA[PILE_TILE / PER], sB[PILE_TILE / PER];
    __shared__ uint32_t s_part[2][PILE_WG / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t seg = list[blockIdx.x];
    const size_t b = seg_start[seg], e = seg + 1 < n_seg ? seg_start[seg + 1] : n_ent;
    // the read's length as its rows state it (a PAF file may disagree with itself: take the largest claim)
    __shared__ uint32_t s_len;
    if (tid == 0) s_len = 0;
    __syncthreads();
    {
        uint32_t l = 0;
        for (size_t k = b + (size_t)tid; k < e; k += PILE_WG) {
            const uint32_t v = ent_val[k];
            const PafRec &r = recs[ent_rec[k]];
            l = max(l, (v & 1u) ? r.qlen : r.tlen);
        }
        if (l) atomicMax(&s_len, l);
    }
    __syncthreads();
    const uint32_t read_len = s_len;
    // the X events of one row, one wave: f(position on this read, lane has an event).  Only the walked side's
    // positions are tracked (one wave scan per 64 ops): query side = ops that consume query bases (= X I), target side
    // = ops that consume target bases (= X D)   (slr2:334-365)
    auto walk = [&](const PafRec &r, bool qside, auto &&f) {
        const bool rev = r.flags & PF_REV;
        uint32_t p = qside ? (rev ? r.qlen - r.qe : r.qs) : r.ts;
        const uint32_t skip = qside ? OP_D : OP_I;
        const uint32_t *o = ops + r.cig_off;
        // four loads of 64 ops in flight (a row of C3 has ~280 ops: one round trip to memory instead of five)
        for (uint32_t k0 = 0; k0 < r.cig_n; k0 += 256) {
            uint32_t opv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const uint32_t idx = k0 + 64u * u + lane; opv[u] = idx < r.cig_n ? o[idx] : skip; }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (k0 + 64u * u >= r.cig_n) break;
                const uint32_t len = opv[u] >> 4, code = opv[u] & 15u;
                const bool isx = code == OP_X;
                const uint32_t sc = wave_prefix_sum_incl_dpp(code == skip ? 0u : len);
                const uint32_t after = p + sc;                           // position after this op
                f(qside && rev ? r.qlen - after + 1 : after, isx);        // slr2:357 / :361
                p += (uint32_t)__builtin_amdgcn_readlane((int)sc, 63);
            }
        }
    };
    unsigned long long ev = 0;
    for (uint32_t t0 = 0; t0 < read_len + 2; t0 += PILE_TILE) {
        const uint32_t n_pos = min((uint32_t)PILE_TILE, read_len + 2 - t0), n_words = (n_pos + PER - 1) / PER;
        for (uint32_t w = tid; w < n_words; w += PILE_WG) { sA[w] = 0; sB[w] = 0; }
        __syncthreads();
        // intervals: A counts starts at s + 1, B counts ends at e (con = #(s < pos) - #(e <= pos), strict on both sides)
        for (size_t k = b + (size_t)tid; k < e; k += PILE_WG) {
            const uint32_t v = ent_val[k];
            const PafRec &r = recs[ent_rec[k]];
            const uint32_t s0 = (v & 1u) ? r.qs : r.ts, e0 = (v & 1u) ? r.qe : r.te;
            if (s0 < e0) {
                const uint32_t ia = max(s0 + 1, t0) - t0, ib = max(e0, t0) - t0;
                if (ia < n_pos) lds_inc<CT>(sA, ia);
                if (ib < n_pos) lds_inc<CT>(sB, ib);
            }
        }
        __syncthreads();
        {   // prefix sums of both arrays, coverage = A - B into A, B cleared: every thread owns a contiguous stretch
            const uint32_t per = (n_pos + PILE_WG - 1) / PILE_WG, lo = min(n_pos, (uint32_t)tid * per), hi = min(n_pos, lo + per);
            uint32_t ta = 0, tb = 0;
            for (uint32_t i = lo; i < hi; ++i) { ta += lds_get<CT>(sA, i); tb += lds_get<CT>(sB, i); }
            // exclusive offsets of the stretches: wave scan, then the totals of the waves before
            const uint32_t ia = wave_prefix_sum_incl_dpp(ta), ib = wave_prefix_sum_incl_dpp(tb);
            if (lane == 63) { s_part[0][wave] = ia; s_part[1][wave] = ib; }
            __syncthreads();                               // (also: every stretch total has been read before the arrays change)
            uint32_t oa = ia - ta, ob = ib - tb;
            for (int w = 0; w < wave; ++w) { oa += s_part[0][w]; ob += s_part[1][w]; }
            // stretches share words at odd boundaries: write 16-bit halves with atomics-free read-modify-write of OWN halves only
            CT *ca = (CT *)sA, *cb = (CT *)sB;
            for (uint32_t i = lo; i < hi; ++i) {
                oa += ca[i]; ob += cb[i];
                ca[i] = (CT)(oa - ob);
                cb[i] = 0;
            }
        }
        __syncthreads();
        // X counts.  The record of a wave's next row is requested before the current row is walked: the chain
        // entry -> record -> CIGAR is three trips to memory that would otherwise line up behind every walk.
        {
            size_t k = b + (size_t)wave;
            uint32_t v = 0;
            PafRec r{};
            if (k < e) { v = ent_val[k]; r = recs[ent_rec[k]]; }
            while (k < e) {
                const size_t kn = k + PILE_WG / 64;
                uint32_t vn = 0;
                PafRec rn{};
                if (kn < e) { vn = ent_val[kn]; rn = recs[ent_rec[kn]]; }
                walk(r, (v & 1u) != 0, [&](uint32_t pos, bool isx) {
                    if (isx && pos >= t0 && pos - t0 < n_pos) lds_inc<CT>(sB, pos - t0);
                    if (t0 == 0) ev += (unsigned long long)__popcll(__ballot(isx));
                });
                k = kn; v = vn; r = rn;
            }
        }
        __syncthreads();
        // supported keys -> their rows' pair counters (rows of pairs nobody reads are skipped before their record is fetched)
        {
            auto next_live = [&](size_t k) {               // first entry of this wave from k on whose pair counter is read
                while (k < e && row_pair[ent_val[k] >> 1] == NO_PAIR) k += PILE_WG / 64;
                return k;
            };
            size_t k = next_live(b + (size_t)wave);
            uint32_t v = 0;
            PafRec r{};
            if (k < e) { v = ent_val[k]; r = recs[ent_rec[k]]; }
            while (k < e) {
                const size_t kn = next_live(k + PILE_WG / 64);
                uint32_t vn = 0;
                PafRec rn{};
                if (kn < e) { vn = ent_val[kn]; rn = recs[ent_rec[kn]]; }
                uint32_t hits = 0;
                walk(r, (v & 1u) != 0, [&](uint32_t pos, bool isx) {
                    if (isx && pos >= t0 && pos - t0 < n_pos) {
                        const int cnt = (int)lds_get<CT>(sB, pos - t0), cov = (int)lds_get<CT>(sA, pos - t0);
                        if (cnt >= mc && cov - cnt >= mc) ++hits;
                    }
                });
                hits = wave_incl_sum_u32(hits, lane);
                if (lane == 63 && hits) atomicAdd(&pair_mut[row_pair[v >> 1]], hits);
                k = kn; v = vn; r = rn;
            }
        }
        __syncthreads();
    }
    if (lane == 0 && ev) atomicAdd(&n_events[(blockIdx.x * 16u + (uint32_t)wave) & (PILE_EV_SLOTS - 1)], ev);
}

// Reads with at most PILE_LIGHT_ROWS rows in the chunk (nearly every (chunk, query) pair: a chunk holds ~500 of the
// 100 000 reads a query could overlap): eight waves, every wave walks at most two rows ONCE and keeps the event
// positions in registers; one 8-bit array of X counts over the read's positions in LDS; the coverage of the few
// positions that reach mc is counted directly from the segment's interval list.  One trip to memory per row, 15 KiB
// of LDS per workgroup.
constexpr int PILE_LIGHT_WG = 512, PILE_LIGHT_IT = 6;      // 6 x 64 CIGAR ops per row are cached
__global__ __launch_bounds__(PILE_LIGHT_WG) void snp_pileup_light_kernel(const PafRec *recs, const uint32_t *ops, const uint32_t *ent_rec,
                                                                          const uint32_t *row_pair, const uint32_t *ent_val,
                                                                          const uint32_t *seg_start, const uint32_t *list, size_t n_seg,
                                                                          size_t n_ent, int mc, uint32_t *pair_mut,
                                                                          unsigned long long *n_events) {
    __shared__ uint32_t sV[PILE_TILE / 4];                            // four 8-bit counters per word
    __shared__ uint32_t s_is[PILE_LIGHT_ROWS], s_ie[PILE_LIGHT_ROWS], s_len;
    // the X events of the cached ops of the rows somebody reads: (position - t0) | row << 16, tested by ALL threads
    __shared__ uint32_t s_ev[PILE_LIGHT_ROWS * 64 * PILE_LIGHT_IT], s_nev, s_hits[PILE_LIGHT_ROWS], s_pair[PILE_LIGHT_ROWS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t seg = list[blockIdx.x];
    const size_t b = seg_start[seg], e = seg + 1 < n_seg ? seg_start[seg + 1] : n_ent;
    const int n_rows = (int)(e - b);
    if (tid == 0) s_len = 0;
    if (tid < PILE_LIGHT_ROWS) s_pair[tid] = NO_PAIR;
    __syncthreads();
    // ---- every wave: its (at most two) rows, walked once ---------------------------------------------------------------
    uint32_t pos[2][PILE_LIGHT_IT];
    uint32_t xm[2] = {0, 0}, pg[2] = {NO_PAIR, NO_PAIR};
    bool spill[2] = {false, false};                                    // row longer than the cache: walked again per phase
    PafRec rr[2];
    bool side[2] = {false, false};
    unsigned long long ev = 0;
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) {
        const int k = wave + 8 * sl;
        if (k >= n_rows) continue;
        const uint32_t v = ent_val[b + k];
        const PafRec r = recs[ent_rec[b + k]];
        rr[sl] = r; side[sl] = (v & 1u) != 0;
        pg[sl] = row_pair[v >> 1];
        const bool qside = side[sl], rev = r.flags & PF_REV;
        if (lane == 0) {
            const uint32_t s0 = qside ? r.qs : r.ts, e0 = qside ? r.qe : r.te;
            s_is[k] = s0 < e0 ? s0 : 0; s_ie[k] = s0 < e0 ? e0 : 0;
            s_pair[k] = pg[sl];
            atomicMax(&s_len, qside ? r.qlen : r.tlen);
        }
        spill[sl] = r.cig_n > 64u * PILE_LIGHT_IT;
        uint32_t p = qside ? (rev ? r.qlen - r.qe : r.qs) : r.ts;
        const uint32_t skip = qside ? OP_D : OP_I;
        const uint32_t *o = ops + r.cig_off;
        uint32_t opv[PILE_LIGHT_IT];
#pragma unroll
        for (int u = 0; u < PILE_LIGHT_IT; ++u) { const uint32_t idx = 64u * u + lane; opv[u] = idx < r.cig_n ? o[idx] : skip; }
#pragma unroll
        for (int u = 0; u < PILE_LIGHT_IT; ++u) {
            const uint32_t len = opv[u] >> 4, code = opv[u] & 15u;
            const uint32_t sc = wave_prefix_sum_incl_dpp(code == skip ? 0u : len);
            const uint32_t after = p + sc;
            pos[sl][u] = qside && rev ? r.qlen - after + 1 : after;
            if (code == OP_X) xm[sl] |= 1u << u;
            p += (uint32_t)__builtin_amdgcn_readlane((int)sc, 63);
        }
        // events of the row (statistics): all its ops, cached or not
        for (uint32_t k0 = 0; k0 < r.cig_n; k0 += 64)
            ev += (unsigned long long)__popcll(__ballot(k0 + lane < r.cig_n && (k0 < 64u * PILE_LIGHT_IT ? ((xm[sl] >> (k0 >> 6)) & 1u) != 0
                                                                                 : (o[k0 + lane] & 15u) == OP_X)));
    }
    __syncthreads();
    const uint32_t read_len = s_len;
    // the uncached tail of a long row: f(position, is X) for the ops from 64 * PILE_LIGHT_IT on
    auto tail = [&](int sl, auto &&f) {
        const PafRec &r = rr[sl];
        const bool qside = side[sl], rev = r.flags & PF_REV;
        const uint32_t skip = qside ? OP_D : OP_I;
        const uint32_t *o = ops + r.cig_off;
        uint32_t p = qside ? (rev ? r.qlen - r.qe : r.qs) : r.ts;
        for (uint32_t k0 = 0; k0 < r.cig_n; k0 += 64) {
            uint32_t len = 0, code = skip;
            if (k0 + lane < r.cig_n) { const uint32_t op = o[k0 + lane]; len = op >> 4; code = op & 15u; }
            const uint32_t sc = wave_prefix_sum_incl_dpp(code == skip ? 0u : len);
            const uint32_t after = p + sc;
            if (k0 >= 64u * PILE_LIGHT_IT) f(qside && rev ? r.qlen - after + 1 : after, code == OP_X);
            p += (uint32_t)__builtin_amdgcn_readlane((int)sc, 63);
        }
    };
    for (uint32_t t0 = 0; t0 < read_len + 2; t0 += PILE_TILE) {
        const uint32_t n_pos = min((uint32_t)PILE_TILE, read_len + 2 - t0);
        for (uint32_t w = tid; w < (n_pos + 3) / 4; w += PILE_LIGHT_WG) sV[w] = 0;
        if (tid < PILE_LIGHT_ROWS) s_hits[tid] = 0;
        if (tid == 0) s_nev = 0;
        __syncthreads();
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            if (wave + 8 * sl >= n_rows) continue;
#pragma unroll
            for (int u = 0; u < PILE_LIGHT_IT; ++u) {
                const bool in = ((xm[sl] >> u) & 1u) && pos[sl][u] >= t0 && pos[sl][u] - t0 < n_pos;
                if (in) lds_inc<uint8_t>(sV, pos[sl][u] - t0);
                if (pg[sl] != NO_PAIR) {                               // (wave-uniform: the row's pair)
                    const unsigned long long m = __ballot(in);
                    if (m) {
                        uint32_t at = 0;
                        if (lane == 0) at = atomicAdd(&s_nev, (uint32_t)__popcll(m));
                        at = (uint32_t)__builtin_amdgcn_readfirstlane((int)at);
                        if (in) s_ev[at + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = (pos[sl][u] - t0) | (uint32_t)(wave + 8 * sl) << 16;
                    }
                }
            }
            if (spill[sl]) tail(sl, [&](uint32_t q, bool isx) { if (isx && q >= t0 && q - t0 < n_pos) lds_inc<uint8_t>(sV, q - t0); });
        }
        __syncthreads();
        auto supported = [&](uint32_t q) {
            const int cnt = (int)lds_get<uint8_t>(sV, q - t0);
            if (cnt < mc) return false;
            int cov = 0;                                              // slr2:383-392: intervals with start < q < end
            for (int i = 0; i < n_rows; ++i) cov += (s_is[i] < q && q < s_ie[i]) ? 1 : 0;
            return cov - cnt >= mc;
        };
        // The coverage test of the X positions, one event per thread: with a wave per row the waves of a segment of five
        // rows left three quarters of the workgroup idle here, and this test is half of the workgroup's time (phase
        // timers: 19 % + 34 % waiting at the barrier behind it for the slowest row).
        {
            const uint32_t n_ev = s_nev;
            for (uint32_t i = tid; i < n_ev; i += PILE_LIGHT_WG) {
                const uint32_t w = s_ev[i];
                if (supported(t0 + (w & 0xffffu))) atomicAdd(&s_hits[w >> 16], 1u);
            }
        }
        // (the uncached tail of a long row stays with the row's wave)
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            if (wave + 8 * sl >= n_rows || pg[sl] == NO_PAIR || !spill[sl]) continue;
            uint32_t hits = 0;
            tail(sl, [&](uint32_t q, bool isx) { if (isx && q >= t0 && q - t0 < n_pos && supported(q)) ++hits; });
            hits = wave_prefix_sum_incl_dpp(hits);
            if (lane == 63 && hits) atomicAdd(&s_hits[wave + 8 * sl], hits);
        }
        __syncthreads();
        if (tid < PILE_LIGHT_ROWS && s_hits[tid]) atomicAdd(&pair_mut[s_pair[tid]], s_hits[tid]);
        __syncthreads();
    }
    if (lane == 0 && ev) atomicAdd(&n_events[(blockIdx.x * 16u + (uint32_t)wave) & (PILE_EV_SLOTS - 1)], ev);
}

// ---------------------------------------------------------------------------------------
// a7: pass 2
// ---------------------------------------------------------------------------------------
// Pair groups whose supported-mutation count pass 2 can ever look at: some row passes the tests that do not depend on
// the count (slr2:102-131).  With len_over near the read length that is a few per cent of the pairs; the events of the
// others still take part in the support counting, but nobody reads their pair's counter.
// row_pair[i] = the pair group of row i, or NO_PAIR for the rows of a group nobody will look at (the event kernel
// reads it instead of searching the group starts)
__global__ void pair_live_kernel(const PafRec *recs, const uint32_t *grows, const uint32_t *seg_start, size_t n_seg,
                                 size_t n_rows, int len_over, int min_o, uint32_t *row_pair) {
    size_t s = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (s >= n_seg) return;
    size_t b = seg_start[s], e = (s + 1 < n_seg) ? seg_start[s + 1] : n_rows;
    bool any = false;
    for (size_t i = b; i < e && !any; ++i) {
        const PafRec &r = recs[grows[i]];
        any = r.qid != r.tid && (int64_t)r.nmatch >= (int64_t)len_over && !is_internal(r, min_o);
    }
    for (size_t i = b; i < e; ++i) row_pair[i] = any ? (uint32_t)s : 0xffffffffu;
}

__global__ void pass2_kernel(const PafRec *recs, const uint32_t *ops, const uint32_t *grows,
                             const uint32_t *seg_start, size_t n_seg, size_t n_rows, const uint32_t *pair_mut,
                             int long_mode, int len_over, double thre, int min_o, uint8_t *keep, uint32_t *xdig) {
    size_t s = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (s >= n_seg) return;
    size_t b = seg_start[s], e = (s + 1 < n_seg) ? seg_start[s + 1] : n_rows;
    const uint32_t mut = pair_mut[s];
    bool done = false;
    for (size_t i = b; i < e; ++i) {
        keep[i] = 0;
        if (done) continue;
        const PafRec &r = recs[grows[i]];
        if (mut > 0) {                                             // slr2:90-100
            if (!long_mode) continue;
            if ((double)mut / (double)r.nmatch > thre) continue;
        }
        if (r.qid == r.tid) continue;                              // slr2:102
        if ((int64_t)r.nmatch < (int64_t)len_over) continue;       // slr2:105
        if (is_internal(r, min_o)) continue;                       // slr2:116-131
        done = true;                                               // slr2:133-136 first surviving row of the pair
        keep[i] = 1;
        uint32_t sum = 0;                                          // slr2:156-161 digit before every 'X'
        const uint32_t *o = ops + r.cig_off;
        for (uint32_t k = 0; k < r.cig_n; ++k)
            if ((o[k] & 15u) == OP_X) sum += (o[k] >> 4) % 10u;
        xdig[i] = sum;
    }
}

std::vector<uint64_t> make_windows(const std::vector<uint64_t> &chunk_row_start, std::vector<uint32_t> &len) {
    std::vector<uint64_t> start;
    for (size_t c = 0; c + 1 < chunk_row_start.size(); ++c)
        for (uint64_t p = chunk_row_start[c]; p < chunk_row_start[c + 1]; p += WINDOW) {
            start.push_back(p);
            len.push_back((uint32_t)std::min<uint64_t>(WINDOW, chunk_row_start[c + 1] - p));
        }
    return start;
}

}  // namespace

namespace {
// rows come chunk by chunk: a wave whose rows share one chunk adds its sum with one atomic
__global__ void chunk_ops_kernel(const PafRec *recs, size_t n, unsigned long long *ops_of_chunk) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const bool live = i < n;
    const uint32_t c = live ? recs[i].chunk : 0xffffffffu;
    unsigned long long v = live ? recs[i].cig_n : 0;
    const uint32_t c0 = (uint32_t)__shfl((int)c, 0, 64);
    if (__all(!live || c == c0)) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if ((threadIdx.x & 63) == 0 && v) atomicAdd(&ops_of_chunk[c0], v);
    } else if (live && v) {
        atomicAdd(&ops_of_chunk[c], v);
    }
}
}  // namespace

std::vector<uint64_t> ops_per_chunk(const PafRec *d_recs, size_t n, uint32_t n_chunk_ids) {
    DBuf<unsigned long long> acc(n_chunk_ids ? n_chunk_ids : 1);
    acc.zero();
    if (n) hipLaunchKernelGGL(chunk_ops_kernel, grid1(n), dim3(WG), 0, stream(), d_recs, n, acc.p);
    HIP_CHECK(hipGetLastError());
    std::vector<unsigned long long> h = acc.download(n_chunk_ids);
    return std::vector<uint64_t>(h.begin(), h.end());
}

std::vector<PafRec> download_rows(const PafRec *d_recs, const std::vector<uint32_t> &idx) {
    std::vector<PafRec> out;
    if (idx.empty()) return out;
    DBuf<uint32_t> d_idx;
    d_idx.upload(idx);
    DBuf<PafRec> tmp(idx.size());
    hipLaunchKernelGGL(gather_rec_kernel, grid1(idx.size()), dim3(WG), 0, stream(), d_recs, d_idx.p, tmp.p, idx.size());
    HIP_CHECK(hipGetLastError());
    return tmp.download(idx.size());
}

void window_filter_device(const PafRec *d_recs, size_t n, const std::vector<uint64_t> &chunk_row_start, int variant,
                          int min_len, double min_iden, int min_o, uint8_t *d_keep) {
    if (!n) return;
    std::vector<uint32_t> wlen;
    std::vector<uint64_t> wstart = make_windows(chunk_row_start, wlen);
    DBuf<uint64_t> d_ws;
    DBuf<uint32_t> d_wl;
    d_ws.upload(wstart);
    d_wl.upload(wlen);
    if (variant == 4)
        hipLaunchKernelGGL(window_filter_kernel<4>, dim3((unsigned)wstart.size()), dim3(WG), 0, stream(), d_recs, d_ws.p,
                           d_wl.p, min_len, min_iden, min_o, d_keep);
    else if (variant == 3)
        hipLaunchKernelGGL(window_filter_kernel<3>, dim3((unsigned)wstart.size()), dim3(WG), 0, stream(), d_recs, d_ws.p,
                           d_wl.p, min_len, min_iden, min_o, d_keep);
    else
        fail(HLMI_EINVAL, "window filter variant must be 3 or 4");
    HIP_CHECK(hipGetLastError());
    sync();
}

void ovlp_inline_device(const PafRec *d_recs, size_t n, int min_len, double min_iden, int o, double r, uint8_t *d_keep,
                        uint32_t *d_first_of) {
    if (!n) return;
    std::vector<uint32_t> wlen;
    std::vector<uint64_t> wstart = make_windows({0, (uint64_t)n}, wlen);
    DBuf<uint64_t> d_ws;
    DBuf<uint32_t> d_wl;
    d_ws.upload(wstart);
    d_wl.upload(wlen);
    hipLaunchKernelGGL(ovlp_inline_kernel, dim3((unsigned)wstart.size()), dim3(WG), 0, stream(), d_recs, d_ws.p, d_wl.p, min_len,
                       min_iden, o, r, d_keep, d_first_of);
    HIP_CHECK(hipGetLastError());
    sync();
}

void filter_stage_device(const PafRec *d_recs, size_t n, const uint32_t *d_ops,
                         const std::vector<uint64_t> &chunk_row_start, const FilterCfg &cfg, FilterOut &out) {
    out = FilterOut();
    if (!n) return;
    // chunk ids inside the rows may be global (a caller filtering a sub-range of chunks): size the chunk sort by the
    // largest id, not by the number of chunks in this call
    const uint32_t n_chunks = std::max<uint32_t>((uint32_t)(chunk_row_start.size() - 1), cfg.chunk_id_bound);
    const int lm = cfg.long_mode ? 1 : 0;

    std::unique_ptr<HostTimer> ht(new HostTimer("filter_v4_select"));
    // ---- a4 ------------------------------------------------------------------------------
    DBuf<uint8_t> keep1(n);
    { KTimer kt("filter_v4");
    window_filter_device(d_recs, n, chunk_row_start, 4, cfg.v4_min_len, cfg.v4_min_iden, cfg.v4_min_o, keep1.p);
    }
    DBuf<uint32_t> rows(n);
    const size_t m = select_flagged_indices(keep1.p, rows.p, n);
    out.n_after_v4 = m;
    if (!m) return;

    ht.reset(); ht.reset(new HostTimer("filter_pair_groups"));
    // ---- group by (chunk, unordered pair) ------------------------------------------------------
    DBuf<uint32_t> rchunk(m);
    DBuf<uint64_t> rkey(m);
    hipLaunchKernelGGL(pair_keys_kernel, grid1(m), dim3(WG), 0, stream(), d_recs, rows.p, m, rchunk.p, rkey.p);
    SortedCK g;
    sort_chunk_key(rchunk.p, rkey.p, m, n_chunks, g);
    DBuf<uint32_t> grows(m);
    hipLaunchKernelGGL(gather_u32_kernel, grid1(m), dim3(WG), 0, stream(), rows.p, g.perm.p, grows.p, m);
    DBuf<uint8_t> head(m);
    hipLaunchKernelGGL(head_flags_kernel, grid1(m), dim3(WG), 0, stream(), g.chunk.p, g.key.p, head.p, m);
    DBuf<uint32_t> pseg_start(m);
    const size_t n_pseg = select_flagged_indices(head.p, pseg_start.p, m);
    out.n_pairs = n_pseg;
    DBuf<uint8_t> sel(m);
    hipLaunchKernelGGL(pair_order_select_kernel, grid1(n_pseg), dim3(WG), 0, stream(), d_recs, d_ops, grows.p, pseg_start.p,
                       n_pseg, m, lm, sel.p);

    ht.reset(); ht.reset(new HostTimer("filter_events"));
    DBuf<uint32_t> pair_mut(n_pseg);
    pair_mut.zero();
    // ---- a5 + a6: per-read pile-up in LDS (the sort-based form below stays as the form very deep reads take) ---------
    bool piled = false;
    if (!hook("HLMI_SNP_SORT")) {
        DBuf<uint64_t> ekey(2 * m);
        DBuf<uint32_t> eval(2 * m);
        hipLaunchKernelGGL(pile_entries_kernel, grid1(m), dim3(WG), 0, stream(), d_recs, grows.p, sel.p, m, lm, ekey.p, eval.p);
        sort_pairs_u64_u32(ekey, eval, 2 * m, 0, 32 + bits_for(n_chunks > 1 ? n_chunks - 1 : 1));
        DBuf<uint32_t> eseg(2 * m);
        // (the sentinel keys of unselected rows form the last run: it is not a segment)
        size_t n_eseg = select_run_heads_u64(ekey.p, 2 * m, 0, eseg.p);
        size_t n_ent = 2 * m;
        if (n_eseg && download_one(ekey.p + (2 * m - 1)) == ~0ull) { --n_eseg; n_ent = download_one(eseg.p + n_eseg); }
        DBuf<uint32_t> max_rows(1);
        max_rows.zero();
        if (n_eseg) hipLaunchKernelGGL(pile_seg_size_kernel, grid1(n_eseg), dim3(WG), 0, stream(), eseg.p, n_eseg, n_ent, max_rows.p);
        if (!n_eseg || download_one(max_rows.p) <= PILE_MAX_ROWS) {
            DBuf<uint32_t> row_pair(m);
            DBuf<unsigned long long> n_ev_d(PILE_EV_SLOTS);
            n_ev_d.zero();
            hipLaunchKernelGGL(pair_live_kernel, grid1(n_pseg), dim3(WG), 0, stream(), d_recs, grows.p, pseg_start.p, n_pseg, m,
                               cfg.len_over, cfg.min_o, row_pair.p);
            if (n_eseg) {
                // a key needs v >= mc rows with an X and mc further spanning rows: reads with fewer than 2 mc rows in the
                // chunk cannot carry one (their events only count for the statistics); the others are piled, reads
                // with many rows by large workgroups (one wave walks one row at a time)
                DBuf<uint32_t> erec(n_ent);
                hipLaunchKernelGGL(pile_entry_rows_kernel, grid1(n_ent), dim3(WG), 0, stream(), eval.p, grows.p, n_ent, erec.p);
                DBuf<uint8_t> cls(n_eseg);
                hipLaunchKernelGGL(pile_seg_class_kernel, grid1(n_eseg), dim3(WG), 0, stream(), eseg.p, n_eseg, n_ent, cfg.mc, cls.p);
                DBuf<uint32_t> l0(n_eseg), l1(n_eseg), l2(n_eseg), l3(1), cnt4(4);
                DBuf<uint8_t> cls1(n_eseg);
                hipLaunchKernelGGL(add_one_u8_kernel, grid1(n_eseg), dim3(WG), 0, stream(), cls.p, cls1.p, n_eseg);
                select_classes4_async(cls1.p, n_eseg, l0.p, l1.p, l2.p, l3.p, cnt4.p);
                const std::vector<uint32_t> nc = cnt4.download(4);
                if (nc[0]) hipLaunchKernelGGL(pile_tiny_events_kernel, dim3(cdiv(nc[0], (size_t)(WG / 64))), dim3(WG), 0, stream(), d_recs,
                                              d_ops, grows.p, eval.p, eseg.p, l0.p, (size_t)nc[0], n_eseg, n_ent, n_ev_d.p);
                if (nc[1]) {
                    KTimer kt("filter_pileup_light");
                    hipLaunchKernelGGL(snp_pileup_light_kernel, dim3(nc[1]), dim3(PILE_LIGHT_WG), 0, stream(), d_recs, d_ops, erec.p,
                                       row_pair.p, eval.p, eseg.p, l1.p, n_eseg, n_ent, cfg.mc, pair_mut.p, n_ev_d.p);
                }
                if (nc[2]) {
                    KTimer kt("filter_pileup_heavy");
                    hipLaunchKernelGGL((snp_pileup_kernel<PILE_WG_HEAVY, uint16_t>), dim3(nc[2]), dim3(PILE_WG_HEAVY), 0, stream(), d_recs, d_ops, erec.p,
                                       row_pair.p, ekey.p, eval.p, eseg.p, l2.p, n_eseg, n_ent, lm, cfg.mc, pair_mut.p, n_ev_d.p);
                }
            }
            HIP_CHECK(hipGetLastError());
            {
                const std::vector<unsigned long long> evs = n_ev_d.download(PILE_EV_SLOTS);
                unsigned long long tot = 0;
                for (unsigned long long v : evs) tot += v;
                out.n_events = (size_t)tot;
            }
            piled = true;
        }
    }
    if (!piled) {
    // ---- a5: events + intervals -------------------------------------------------------------------
    DBuf<uint32_t> n_ev(m), n_iv(m), ev_off(m), iv_off(m);
    const dim3 rows_grid((unsigned)std::min<size_t>(cdiv(m ? m : 1, WG / 64), 65536));    // one wave per row, grid-stride
    DBuf<uint32_t> maxv(2);
    maxv.zero();
    hipLaunchKernelGGL(snp_count_kernel, rows_grid, dim3(WG), 0, stream(), d_recs, d_ops, grows.p, sel.p, m, lm, n_ev.p,
                       n_iv.p, maxv.p);
    exclusive_scan_u32(n_ev.p, ev_off.p, m);
    exclusive_scan_u32(n_iv.p, iv_off.p, m);
    const size_t E = (size_t)download_one(ev_off.p + (m - 1)) + download_one(n_ev.p + (m - 1));
    const size_t I = (size_t)download_one(iv_off.p + (m - 1)) + download_one(n_iv.p + (m - 1));
    out.n_events = E;
    if (E) {
        // one sort word per event / interval end: chunk | read | position, as wide as this call's ids and lengths need
        const std::vector<uint32_t> mx = maxv.download(2);
        const int cb = bits_for(n_chunks > 1 ? n_chunks - 1 : 1), hb = bits_for(mx[0] ? mx[0] : 1), lb = bits_for((uint64_t)mx[1] + 1);
        if (cb + hb + lb > 64) fail(HLMI_EINVAL, "SNP pile-up key needs %d bits (chunks %d, reads %d, positions %d)", cb + hb + lb, cb, hb, lb);
        DBuf<uint32_t> ev_pair(E);
        DBuf<uint64_t> ev_ck(E), iv_sck(I ? I : 1), iv_eck(I ? I : 1);
    { KTimer kt("filter_event_fill");
        DBuf<uint32_t> row_pair(m);
        hipLaunchKernelGGL(pair_live_kernel, grid1(n_pseg), dim3(WG), 0, stream(), d_recs, grows.p, pseg_start.p, n_pseg, m,
                           cfg.len_over, cfg.min_o, row_pair.p);
        hipLaunchKernelGGL(snp_fill_kernel, rows_grid, dim3(WG), 0, stream(), d_recs, d_ops, grows.p, sel.p, m, lm,
                           ev_off.p, iv_off.p, row_pair.p, hb, lb, ev_ck.p, ev_pair.p, iv_sck.p, iv_eck.p);
    }
    { KTimer kt("filter_event_sort");
        sort_pairs_u64_u32(ev_ck, ev_pair, E, 0, cb + hb + lb);      // the pair group rides along: no gather afterwards
        sort_keys_u64(iv_sck, I, 0, cb + hb + lb);
        sort_keys_u64(iv_eck, I, 0, cb + hb + lb);
    }
        DBuf<uint32_t> kseg_start(E);
        const size_t n_kseg = select_run_heads_u64(ev_ck.p, E, 0, kseg_start.p);

        // ---- a6 ---------------------------------------------------------------------------------
        DBuf<uint8_t> ev_sup(E);
        ev_sup.zero();
    { KTimer kt("filter_support");
        hipLaunchKernelGGL(snp_support_kernel, grid1(n_kseg), dim3(WG), 0, stream(), ev_ck.p, kseg_start.p, n_kseg, E,
                           iv_sck.p, iv_eck.p, I, lb, cfg.mc, ev_sup.p);
    }
    { KTimer kt("filter_pair_count");
        hipLaunchKernelGGL(snp_pair_count_kernel, grid1(E), dim3(WG), 0, stream(), ev_pair.p, ev_sup.p, E, pair_mut.p);
    }
        HIP_CHECK(hipGetLastError());
        sync();
    }

    }

    ht.reset(); ht.reset(new HostTimer("filter_pass2_order"));
    // ---- a7 ---------------------------------------------------------------------------------------
    DBuf<uint8_t> keep2(m);
    DBuf<uint32_t> xdig(m);
    xdig.zero();
    hipLaunchKernelGGL(pass2_kernel, grid1(n_pseg), dim3(WG), 0, stream(), d_recs, d_ops, grows.p, pseg_start.p, n_pseg,
                       m, pair_mut.p, lm, cfg.len_over, cfg.thre, cfg.min_o, keep2.p, xdig.p);
    HIP_CHECK(hipGetLastError());
    DBuf<uint32_t> kidx(m);
    const size_t nk = select_flagged_indices(keep2.p, kidx.p, m);
    if (!nk) return;
    DBuf<uint32_t> krows(nk), kx(nk);
    hipLaunchKernelGGL(gather_u32_kernel, grid1(nk), dim3(WG), 0, stream(), grows.p, kidx.p, krows.p, nk);
    hipLaunchKernelGGL(gather_u32_kernel, grid1(nk), dim3(WG), 0, stream(), xdig.p, kidx.p, kx.p, nk);
    std::vector<uint32_t> hrows = krows.download(nk), hx = kx.download(nk);
    if (!cfg.reference_order) {
        out.rows.swap(hrows);
        out.x_digit_sum.swap(hx);
        return;
    }
    // reference write order: per chunk, the slr2:57 order.  Only kept rows are ordered on the host.
    std::vector<PafRec> hrecs(nk);
    {
        DBuf<PafRec> tmp(nk);   // gather the kept records (64 B each) for the ordering keys
        hipLaunchKernelGGL(gather_rec_kernel, grid1(nk), dim3(WG), 0, stream(), d_recs, krows.p, tmp.p, nk);
        hrecs = tmp.download(nk);
    }
    std::vector<uint32_t> ord(nk);
    for (size_t i = 0; i < nk; ++i) ord[i] = (uint32_t)i;
    std::sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t b) {
        const PafRec &x = hrecs[a], &y = hrecs[b];
        if (x.chunk != y.chunk) return x.chunk < y.chunk;
        if (x.tlen != y.tlen) return x.tlen < y.tlen;
        if (x.ts != y.ts) return x.ts < y.ts;
        if (x.te != y.te) return x.te < y.te;
        return x.tie < y.tie;
    });
    out.rows.resize(nk);
    out.x_digit_sum.resize(nk);
    for (size_t i = 0; i < nk; ++i) {
        out.rows[i] = hrows[ord[i]];
        out.x_digit_sum[i] = hx[ord[i]];
    }
}

}  // namespace hlmi
