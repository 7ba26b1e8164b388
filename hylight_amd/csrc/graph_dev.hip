// graph_dev.hip - the data-parallel half of the overlap-graph build, resident in HBM from the PAF text to the
// transitively reduced string graph (what miniasm does in hit.c, sdict.c, asm.c:9-39 and
// asg.c:104-193 with the flags of script/HyLight.py:137,140,171).
//
//   text      the file is uploaded as bytes; one thread per line splits its tab fields and converts the eleven
//             numeric columns with strtol's rules (paf.c:34-61)
//   read ids  "order of first appearance" (sdict.c:27-45) without a hash table: every name occurrence is hashed,
//             occurrences are grouped by a stable sort on the hash, a group's first occurrence ranks the name
//   windows   per read the longest stretch covered by >= min_dp overlaps (hit.c:109-160): all interval ends of all
//             reads are sorted as (read, position, is_end) words in ONE device sort, then a wave sweeps each read's
//             run with a prefix sum of +-1 and a running "last start" maximum
//   per-hit   clipping to the windows (hit.c:162-193), overlap -> arc classification (miniasm.h:86-104) as
//             filter / containment test / arc generator, each followed by an order-preserving compaction
//   graph     arcs sorted by (vertex, length), vertex index, transitive reduction (asg.c:148-193) with one wave
//             per vertex: the vertex's out-neighbours sit in an LDS hash table that carries their mark, the
//             neighbour lists are streamed through it; duplicate arcs (asg.c:104-121) and arcs without their
//             reverse (asg.c:124-138) per vertex / per arc
// The two sorts whose tie order shows in the output (overlaps by (read, start), arcs by (vertex, length)) take their
// permutation from reference_sort_order() (graph_host.cpp); the records themselves never leave the device.
// Integer and byte work: HBM-bound, no MFMA.
#include <algorithm>

#include "dev_prims.h"
#include "graph.h"
#include "paf_io.h"
#include "wave_ops.h"

namespace hlmi {

namespace {
constexpr int WG = 256;
inline dim3 grid1(size_t n) { return dim3(cdiv(n ? n : 1, WG)); }

// ---------------------------------------------------------------------------------------------
// a9: text -> rows
// ---------------------------------------------------------------------------------------------
struct PafRow {                 // one parsed line (paf.h:21-25)
    uint32_t ql, qs, qe, tl, ts, te, ml, bl;
    uint64_t qn_off, tn_off;     // (byte offsets into the file: a merged PAF may exceed 4 GiB)
    uint32_t qn_len, tn_len;
    uint32_t rev;
};

// flags of the window txt[base .. base + n): 1 where a line starts
__global__ void line_start_kernel(const uint8_t *txt, size_t base, size_t n, uint8_t *flag) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) flag[i] = (base + i == 0 || txt[base + i - 1] == '\n') ? 1 : 0;
}
__global__ void add_base_kernel(const uint32_t *rel, size_t n, uint64_t base, uint64_t *out) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) out[i] = base + rel[i];
}

// strtol(token, 0, 10) truncated to 32 bits: white space, sign, digits, saturation at LONG_MIN / LONG_MAX
__device__ uint32_t strtol_u32(const uint8_t *s, uint64_t a, uint64_t b) {
    while (a < b && (s[a] == ' ' || (s[a] >= 9 && s[a] <= 13))) ++a;
    bool neg = false;
    if (a < b && (s[a] == '+' || s[a] == '-')) { neg = s[a] == '-'; ++a; }
    const unsigned long long lim = neg ? 0x8000000000000000ull : 0x7fffffffffffffffull;
    unsigned long long v = 0;
    bool sat = false;
    for (; a < b && s[a] >= '0' && s[a] <= '9'; ++a) {
        if (sat) continue;
        const unsigned d = s[a] - '0';
        if (v > (lim - d) / 10) { v = lim; sat = true; }
        else v = v * 10 + d;
    }
    return (uint32_t)(neg ? 0ull - v : v);
}

__global__ void parse_rows_kernel(const uint8_t *txt, size_t n_bytes, const uint64_t *line_start, size_t n_lines, int min_span,
                                  int min_match, PafRow *rows, uint8_t *ok) {
    size_t L = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (L >= n_lines) return;
    uint64_t a = line_start[L];
    uint64_t e = L + 1 < n_lines ? line_start[L + 1] - 1 : (uint64_t)(txt[n_bytes - 1] == '\n' ? n_bytes - 1 : n_bytes);
    if (e - a > 1 && txt[e - 1] == '\r') --e;           // the line reader drops one trailing CR (kseq.h)
    PafRow r{};
    int t = 0;
    uint64_t p = a;
    for (uint64_t i = a; i <= e; ++i) {
        if (i < e && txt[i] != '\t') continue;
        switch (t) {                                     // paf.c:42-55
            case 0: r.qn_off = p; r.qn_len = (uint32_t)(i - p); break;
            case 1: r.ql = strtol_u32(txt, p, i); break;
            case 2: r.qs = strtol_u32(txt, p, i); break;
            case 3: r.qe = strtol_u32(txt, p, i); break;
            case 4: r.rev = p < i && txt[p] == '-'; break;
            case 5: r.tn_off = p; r.tn_len = (uint32_t)(i - p); break;
            case 6: r.tl = strtol_u32(txt, p, i); break;
            case 7: r.ts = strtol_u32(txt, p, i); break;
            case 8: r.te = strtol_u32(txt, p, i); break;
            case 9: r.ml = strtol_u32(txt, p, i) & 0x7fffffffu; break;
            case 10: r.bl = strtol_u32(txt, p, i); break;
            default: break;
        }
        ++t;
        p = i + 1;
    }
    // paf.c:57 (fewer than ten fields: not a record) and hit.c:85 (unsigned differences against int thresholds)
    const bool keep = t >= 10 && !(r.qe - r.qs < (uint32_t)min_span || r.te - r.ts < (uint32_t)min_span || (int)r.ml < min_match);
    rows[L] = r;
    ok[L] = keep ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------
// a9: names -> ids in order of first appearance
// ---------------------------------------------------------------------------------------------
// occurrence 2 i = query name of kept row i, 2 i + 1 = its target name (the order sd_put sees them, hit.c:88-90)
__device__ __forceinline__ void occ_name(const PafRow *rows, const uint32_t *kept, uint32_t occ, uint64_t &off, uint32_t &len) {
    const PafRow &r = rows[kept[occ >> 1]];
    off = (occ & 1) ? r.tn_off : r.qn_off;
    len = (occ & 1) ? r.tn_len : r.qn_len;
}
__global__ void name_hash_kernel(const uint8_t *txt, const PafRow *rows, const uint32_t *kept, size_t n_occ, uint64_t seed,
                                 uint64_t *hash, uint32_t *occ_id) {
    size_t o = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (o >= n_occ) return;
    uint64_t off;
    uint32_t len;
    occ_name(rows, kept, (uint32_t)o, off, len);
    uint64_t h = seed ^ (0x9e3779b97f4a7c15ull * (len + 1));
    for (uint32_t i = 0; i < len; ++i) {
        h = (h ^ txt[off + i]) * 0x100000001b3ull;
        h ^= h >> 29;
    }
    h ^= h >> 32;
    hash[o] = h * 0xd6e8feb86659fd93ull;
    occ_id[o] = (uint32_t)o;
}
// sorted by hash (stable: occurrences ascending inside a group).  head[i] = first of its group; every other member is
// compared byte by byte with the member before it, so two different names with one hash are noticed (and the caller
// starts over with another seed).
__global__ void name_group_kernel(const uint8_t *txt, const PafRow *rows, const uint32_t *kept, const uint64_t *hash,
                                  const uint32_t *occ_sorted, size_t n_occ, uint8_t *head, uint32_t *collision) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n_occ) return;
    if (i == 0 || hash[i] != hash[i - 1]) { head[i] = 1; return; }
    head[i] = 0;
    uint64_t o1, o2;
    uint32_t l1, l2;
    occ_name(rows, kept, occ_sorted[i - 1], o1, l1);
    occ_name(rows, kept, occ_sorted[i], o2, l2);
    bool same = l1 == l2;
    for (uint32_t k = 0; same && k < l1; ++k) same = txt[o1 + k] == txt[o2 + k];
    if (!same) *collision = 1;
}
// first[o] = 1 where occurrence o is the first one of its name
__global__ void name_first_kernel(const uint32_t *occ_sorted, const uint8_t *head, size_t n_occ, uint8_t *first) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n_occ && head[i]) first[occ_sorted[i]] = 1;
}
// group g (0-based, from the scan of head) starts at sorted position group_start[g]; its id is the number of first
// occurrences before its own first occurrence
__global__ void name_assign_kernel(const uint32_t *occ_sorted, const uint32_t *group_of, const uint32_t *group_start,
                                   const uint32_t *first_rank, size_t n_occ, uint32_t *id_of_occ) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n_occ) return;
    const uint32_t g = group_of[i] - 1;                   // inclusive scan of head
    id_of_occ[occ_sorted[i]] = first_rank[occ_sorted[group_start[g]]];
}
__global__ void widen_u8_kernel(const uint8_t *in, uint32_t *out, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i];
}
__global__ void inclusive_from_exclusive_kernel(const uint32_t *excl, const uint32_t *in, uint32_t *out, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) out[i] = excl[i] + in[i];
}
__global__ void name_ref_kernel(const PafRow *rows, const uint32_t *kept, const uint8_t *first, const uint32_t *first_rank,
                                size_t n_occ, uint64_t *ref_off, uint32_t *ref_len) {
    size_t o = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (o >= n_occ || !first[o]) return;
    uint64_t off;
    uint32_t len;
    occ_name(rows, kept, (uint32_t)o, off, len);
    ref_off[first_rank[o]] = off;
    ref_len[first_rank[o]] = len;
}

// ---------------------------------------------------------------------------------------------
// a9: rows -> overlaps (each row also from its target's side, hit.c:92-98)
// ---------------------------------------------------------------------------------------------
__global__ void ovl_count_kernel(const uint32_t *id_of_occ, size_t n_rows, uint32_t *cnt) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n_rows) cnt[i] = id_of_occ[2 * i] != id_of_occ[2 * i + 1] ? 2u : 1u;
}
__global__ void ovl_fill_kernel(const PafRow *rows, const uint32_t *kept, const uint32_t *id_of_occ, const uint32_t *at,
                                size_t n_rows, Ovl *out, uint64_t *key) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n_rows) return;
    const PafRow &r = rows[kept[i]];
    const uint32_t q = id_of_occ[2 * i], t = id_of_occ[2 * i + 1];
    const uint32_t mlr = r.ml | (r.rev ? 0x80000000u : 0u), bl = r.bl & 0x7fffffffu;
    uint32_t w = at[i];
    out[w] = Ovl{q, r.qs, r.qe, t, r.ts, r.te, mlr, bl};
    key[w] = (uint64_t)q << 32 | r.qs;
    if (q != t) {
        ++w;
        out[w] = Ovl{t, r.ts, r.te, q, r.qs, r.qe, mlr, bl};
        key[w] = (uint64_t)t << 32 | r.ts;
    }
}
template <typename T>
__global__ void gather_kernel(const T *src, const uint32_t *idx, T *dst, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
}

// ---------------------------------------------------------------------------------------------
// a10: coverage windows
// ---------------------------------------------------------------------------------------------
constexpr uint64_t NO_END = ~0ull;
// two words per overlap: read << 32 | position << 1 | is_end   (hit.c:122-130; self matches and overlaps below the
// identity floor contribute nothing)
__global__ void win_ends_kernel(const Ovl *ovl, size_t n, float min_iden, uint64_t *ends) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Ovl &h = ovl[i];
    const int ml = (int)(h.ml_rev & 0x7fffffffu), bl = (int)h.bl;
    const bool skip = h.t == h.q || (float)ml < (float)bl * min_iden || !(h.qe > h.qs);
    ends[2 * i] = skip ? NO_END : (uint64_t)h.q << 32 | (uint32_t)(h.qs << 1);
    ends[2 * i + 1] = skip ? NO_END : (uint64_t)h.q << 32 | (uint32_t)(h.qe << 1 | 1u);
}
__device__ __forceinline__ size_t lower_bound_u64(const uint64_t *a, size_t n, uint64_t v) {
    size_t lo = 0, hi = n;
    while (lo < hi) {
        const size_t m = (lo + hi) >> 1;
        if (a[m] < v) lo = m + 1; else hi = m;
    }
    return lo;
}
__device__ __forceinline__ int wave_incl_sum_i32(int v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int u = __shfl_up(v, o, 64);
        if (lane >= o) v += u;
    }
    return v;
}
__device__ __forceinline__ uint32_t wave_incl_max_u32(uint32_t v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t u = __shfl_up(v, o, 64);
        if (lane >= o) v = u > v ? u : v;
    }
    return v;
}
// One wave per read that still has overlaps.  Depth after the j-th end = prefix sum of +1 / -1; a window opens where
// the depth reaches min_dp going up and closes where it leaves it going down; the longest window wins, the first one
// among equals (hit.c:133-147: strict `>`).  Positions ascend along the run, so "start of the window that closes
// here" is a running maximum over the opening positions.
__global__ __launch_bounds__(WG) void win_sweep_kernel(const Ovl *ovl, const uint32_t *run_head, size_t n_runs, const uint64_t *ends,
                                                        size_t n_ends, int min_dp, ReadWin *win) {
    const int lane = threadIdx.x & 63;
    const size_t r = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
    if (r >= n_runs) return;
    const uint32_t q = ovl[run_head[r]].q;
    const size_t lo = lower_bound_u64(ends, n_ends, (uint64_t)q << 32), hi = lower_bound_u64(ends, n_ends, ((uint64_t)q + 1) << 32);
    int depth0 = 0;                         // depth before this block of 64
    uint32_t open0 = 0;                     // 1 + position where the current window opened (0: none yet)
    unsigned long long best = 0;            // length << 32 | ~ordinal of the closing end
    uint32_t best_s = 0, best_e = 0;
    uint32_t n_closed = 0;
    for (size_t b = lo; b < hi; b += 64) {
        const size_t j = b + (size_t)lane;
        const bool live = j < hi;
        const uint32_t w = live ? (uint32_t)ends[j] : 0u;
        const int step = live ? ((w & 1u) ? -1 : 1) : 0;
        const int dp = depth0 + wave_incl_sum_i32(step, lane), old = dp - step;
        const bool opens = live && old < min_dp && dp >= min_dp, closes = live && old >= min_dp && dp < min_dp;
        uint32_t o = opens ? (w >> 1) + 1u : 0u;
        o = wave_incl_max_u32(o, lane);
        if (o == 0) o = open0;
        const unsigned long long cm = __ballot(closes);
        if (closes) {
            const uint32_t start = o - 1u, end = w >> 1;
            const uint32_t ord = n_closed + (uint32_t)__popcll(cm & ((1ull << lane) - 1ull));
            const unsigned long long cand = (unsigned long long)(end - start) << 32 | (0xffffffffu - ord);
            if (cand > best) { best = cand; best_s = start; best_e = end; }
        }
        n_closed += (uint32_t)__popcll(cm);
        depth0 = __shfl(dp, 63, 64);
        open0 = __shfl(o, 63, 64);
    }
    // the wave's best: largest length, smallest ordinal
    unsigned long long m = best;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long u = __shfl_xor(m, o, 64);
        m = u > m ? u : m;
    }
    const unsigned long long who = __ballot(best == m && (m >> 32) != 0);
    if (who) {
        const int src = __ffsll((long long)who) - 1;
        const uint32_t s = __shfl(best_s, src, 64), e = __shfl(best_e, src, 64);
        if (lane == 0) win[q] = ReadWin{s & 0x7fffffffu, e, 0, 0};
    } else if (lane == 0) {
        win[q] = ReadWin{0, 0, 1, 0};
    }
}

// ---------------------------------------------------------------------------------------------
// a11-a13: per-overlap kernels
// ---------------------------------------------------------------------------------------------
// hit.c:162-193.  The reference mixes int and unsigned here (window starts are 31-bit fields that promote to int,
// window ends are 32-bit unsigned); the casts below reproduce its comparisons.
__global__ void clip_kernel(Ovl *ovl, size_t n, const ReadWin *win, int min_span, uint8_t *keep) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    Ovl h = ovl[i];
    const ReadWin wq = win[h.q], wt = win[h.t];
    if (wq.del || wt.del) { keep[i] = 0; return; }
    const bool rev = h.ml_rev >> 31;
    int qs, qe, ts, te;
    if (rev) {
        qs = (int)(h.te < wt.e ? h.qs : h.qs + (h.te - wt.e));
        qe = (int)(h.ts > wt.s ? h.qe : h.qe - (wt.s - h.ts));
        ts = (int)(h.qe < wq.e ? h.ts : h.ts + (h.qe - wq.e));
        te = (int)(h.qs > wq.s ? h.te : h.te - (wq.s - h.qs));
    } else {
        qs = (int)(h.ts > wt.s ? h.qs : h.qs + (wt.s - h.ts));
        qe = (int)(h.te < wt.e ? h.qe : h.qe - (h.te - wt.e));
        ts = (int)(h.qs > wq.s ? h.ts : h.ts + (wq.s - h.qs));
        te = (int)(h.qe < wq.e ? h.te : h.te - (h.qe - wq.e));
    }
    const int sq = (int)wq.s, st = (int)wt.s;
    qs = (qs > sq ? qs : sq) - sq;
    qe = (int)(((uint32_t)qe < wq.e ? (uint32_t)qe : wq.e) - (uint32_t)sq);
    ts = (ts > st ? ts : st) - st;
    te = (int)(((uint32_t)te < wt.e ? (uint32_t)te : wt.e) - (uint32_t)st);
    const bool ok = qe - qs >= min_span && te - ts >= min_span;
    keep[i] = ok ? 1 : 0;
    if (ok) { h.qs = (uint32_t)qs; h.qe = (uint32_t)qe; h.ts = (uint32_t)ts; h.te = (uint32_t)te; ovl[i] = h; }
}

// overlap -> arc (miniasm.h:86-104): < 0 = internal match / query contained / target contained / too short, else the
// arc from an end of the query read to an end of the target read.  ql, tl: lengths of the reads' windows.
constexpr int OV_INTERNAL = -1, OV_QCONT = -2, OV_TCONT = -3, OV_SHORT = -4;
__device__ int classify_ovl(const Ovl &h, int ql, int tl, int max_hang, float int_frac, int min_ovlp, Arc &arc) {
    const bool rev = h.ml_rev >> 31;
    const int32_t qs = (int32_t)h.qs;
    // overhang of the target beyond the alignment, on the query's 5' and 3' side
    const int32_t t5 = rev ? (int32_t)((uint32_t)tl - h.te) : (int32_t)h.ts;
    const int32_t t3 = rev ? (int32_t)h.ts : (int32_t)((uint32_t)tl - h.te);
    const uint32_t q3 = (uint32_t)ql - h.qe;                           // unsigned in the reference
    const int32_t ext5 = qs < t5 ? qs : t5;
    const int32_t ext3 = q3 < (uint32_t)t3 ? (int32_t)q3 : t3;
    const uint32_t span = h.qe - (uint32_t)qs;
    if (ext5 > max_hang || ext3 > max_hang || (float)span < (float)(span + (uint32_t)ext5 + (uint32_t)ext3) * int_frac) return OV_INTERNAL;
    if (qs <= t5 && q3 <= (uint32_t)t3) return OV_QCONT;
    if (qs >= t5 && q3 >= (uint32_t)t3) return OV_TCONT;
    uint32_t u, v, l;
    if (qs > t5) { u = 0; v = rev ? 1u : 0u; l = (uint32_t)(qs - t5); }
    else { u = 1; v = rev ? 0u : 1u; l = q3 - (uint32_t)t3; }
    if (span + (uint32_t)ext5 + (uint32_t)ext3 < (uint32_t)min_ovlp || h.te - h.ts + (uint32_t)ext5 + (uint32_t)ext3 < (uint32_t)min_ovlp) return OV_SHORT;
    arc.ul = (uint64_t)(h.q << 1 | u) << 32 | l;
    arc.v = h.t << 1 | v;
    arc.ol_del = ((uint32_t)ql - l) & 0x7fffffffu;
    return (int)l;
}
__device__ __forceinline__ int win_len(const ReadWin &w) { return (int)(w.e - w.s); }

// hit.c:195-216: keep what would be an arc or a containment under relaxed thresholds
__global__ void crude_filter_kernel(const Ovl *ovl, size_t n, const ReadWin *win, int max_hang, int min_ovlp, uint8_t *keep) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Ovl h = ovl[i];
    const ReadWin wq = win[h.q], wt = win[h.t];
    Arc a;
    int r = OV_INTERNAL;
    if (!wq.del && !wt.del) r = classify_ovl(h, win_len(wq), win_len(wt), max_hang, .5f, min_ovlp, a);
    keep[i] = (!wq.del && !wt.del && (r >= 0 || r == OV_QCONT || r == OV_TCONT)) ? 1 : 0;
}
// hit.c:218-223
__global__ void win_merge_kernel(ReadWin *a, const ReadWin *b, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    ReadWin w = a[i];
    w.e = w.s + b[i].e;
    w.s = (w.s + b[i].s) & 0x7fffffffu;
    a[i] = w;
}
// hit.c:231-237 + hit.c:24-36: contained reads are dropped, so are reads no overlap mentions
__global__ void contained_kernel(const Ovl *ovl, size_t n, ReadWin *win, const GraphOpt o, uint32_t *used) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Ovl h = ovl[i];
    Arc a;
    const int r = classify_ovl(h, win_len(win[h.q]), win_len(win[h.t]), o.max_hang, o.int_frac, o.min_ovlp, a);
    if (r == OV_QCONT) win[h.q].del = 1;
    else if (r == OV_TCONT) win[h.t].del = 1;
    used[h.q] = 1;
    used[h.t] = 1;
}
__global__ void read_alive_kernel(const ReadWin *win, const uint32_t *used, size_t n, uint32_t *alive) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) alive[i] = (!win[i].del && used[i]) ? 1u : 0u;
}
__global__ void squeeze_reads_kernel(const ReadWin *win, const uint64_t *ref_off, const uint32_t *ref_len, const uint32_t *alive,
                                     const uint32_t *new_id, size_t n, ReadWin *win2, uint64_t *ref_off2, uint32_t *ref_len2) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n || !alive[i]) return;
    const uint32_t k = new_id[i];
    win2[k] = win[i];
    ref_off2[k] = ref_off[i];
    ref_len2[k] = ref_len[i];
}
__global__ void renumber_kernel(Ovl *ovl, size_t n, const uint32_t *alive, const uint32_t *new_id, uint8_t *keep) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    Ovl h = ovl[i];
    const bool ok = alive[h.q] && alive[h.t];
    keep[i] = ok ? 1 : 0;
    if (ok) { h.q = new_id[h.q]; h.t = new_id[h.t]; ovl[i] = h; }
}

// asm.c:9-39
__global__ void seq_init_kernel(const ReadWin *win, size_t n, uint32_t *seq) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) seq[i] = ((win[i].e - win[i].s) & 0x7fffffffu) | (win[i].del ? 0x80000000u : 0u);
}
__global__ void make_arcs_kernel(const Ovl *ovl, size_t n, const ReadWin *win, const GraphOpt o, Arc *arc, uint8_t *is_arc,
                                 uint32_t *seq) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Ovl h = ovl[i];
    Arc a{};
    const int r = classify_ovl(h, win_len(win[h.q]), win_len(win[h.t]), o.max_hang, o.int_frac, o.min_ovlp, a);
    bool emit = false;
    if (r >= 0) {
        if (h.q == h.t) {       // a read against its own reverse complement over the same interval (asm.c:27-31)
            if (h.qs == h.ts && h.qe == h.te && (h.ml_rev >> 31)) atomicOr(&seq[h.q], 0x80000000u);
        } else emit = true;
    } else if (r == OV_QCONT) atomicOr(&seq[h.q], 0x80000000u);
    is_arc[i] = emit ? 1 : 0;
    if (emit) arc[i] = a;
}

// ---------------------------------------------------------------------------------------------
// graph maintenance (asg.c:27-80)
// ---------------------------------------------------------------------------------------------
__global__ void arc_live_kernel(const Arc *arc, size_t n, const uint32_t *seq, uint8_t *live) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Arc a = arc[i];
    live[i] = (!(a.ol_del >> 31) && !(seq[a.ul >> 33] >> 31) && !(seq[a.v >> 1] >> 31)) ? 1 : 0;
}
__global__ void arc_key_kernel(const Arc *arc, size_t n, uint64_t *key) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) key[i] = arc[i].ul;
}
// idx[v] = first arc << 32 | number of arcs (arcs sorted by source vertex)
__global__ void vertex_index_kernel(const Arc *arc, size_t n, uint64_t *idx) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t v = (uint32_t)(arc[i].ul >> 32);
    if (i == 0 || (uint32_t)(arc[i - 1].ul >> 32) != v) {
        size_t e = i + 1;
        while (e < n && (uint32_t)(arc[e].ul >> 32) == v) ++e;      // degrees are small; one thread per run head
        idx[v] = (uint64_t)i << 32 | (uint64_t)(e - i);
    }
}

// ---------------------------------------------------------------------------------------------
// a14: transitive reduction, one wave per vertex (asg.c:148-193)
// ---------------------------------------------------------------------------------------------
// The out-neighbours of v sit in an open-addressing table in LDS: key = neighbour vertex, value = its mark (1 =
// neighbour, 2 = reachable through another neighbour within L = longest arc + fuzz) and the index of the FIRST arc
// v -> w (the reference keeps one mark per vertex, so of several arcs to one w only the first is ever deleted).  The
// outer walk over v's arcs stays in arc order - a neighbour already marked 2 is not expanded, exactly as in the
// reference - while the lanes stream the neighbour's own arc list through the table.
constexpr int TR_SLOTS = 2048;              // per wave; vertices with more than TR_SLOTS / 2 arcs take the global-memory table
constexpr int TR_BITS = 11;
constexpr uint32_t TR_EMPTY = 0xffffffffu;
constexpr int TR_WAVES = 4;
static_assert(TR_SLOTS == 1 << TR_BITS, "table size and hash bits");
__device__ __forceinline__ uint32_t tr_hash(uint32_t w, int bits) { return (w * 0x9e3779b1u) >> (32 - bits); }

// Table access.  The LDS table is plain memory of the wave.  The global-memory table of a big vertex is touched by one wave
// only as well, but through the vector L1, which does not see the wave's own atomics: its reads are agent-scope atomic
// loads (served by L2, where the atomics happen).
template <bool BIG> __device__ __forceinline__ uint32_t tr_load(const uint32_t *p) {
    if constexpr (BIG) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return *p;
}

// one wave, one vertex: key / val = its table of (1 << bits) slots, already holding TR_EMPTY / (1 << 30 | 0x3fffffff)
template <bool BIG>
__device__ void reduce_vertex(const Arc *arc, const uint64_t *idx, uint32_t v, uint32_t fuzz, uint8_t *del, uint32_t *n_reduced,
                              uint32_t *key, uint32_t *val, int bits, int lane) {
    const uint32_t nv = (uint32_t)idx[v], mask = (1u << bits) - 1u;
    const uint64_t b = idx[v] >> 32;
    const Arc *av = arc + b;
    for (uint32_t i = lane; i < nv; i += 64) {         // duplicates of a neighbour share one slot: smallest arc index wins
        const uint32_t w = av[i].v;
        uint32_t h = tr_hash(w, bits);
        for (;;) {
            const uint32_t old = atomicCAS(&key[h], TR_EMPTY, w);
            if (old == TR_EMPTY || old == w) break;
            h = (h + 1) & mask;
        }
        atomicMin(&val[h], 1u << 30 | i);
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    auto slot_of = [&](uint32_t w) -> int {            // -1: not a neighbour of v
        uint32_t h = tr_hash(w, bits);
        for (;;) {
            const uint32_t k = tr_load<BIG>(&key[h]);
            if (k == w) return (int)h;
            if (k == TR_EMPTY) return -1;
            h = (h + 1) & mask;
        }
    };
    const uint32_t L = (uint32_t)av[nv - 1].ul + fuzz;
    for (uint32_t i = 0; i < nv; ++i) {
        const uint32_t w = av[i].v, li = (uint32_t)av[i].ul;
        const int sw = slot_of(w);                      // uniform across the wave
        if ((tr_load<BIG>(&val[sw]) >> 30) != 1u) continue;
        const uint32_t nw = (uint32_t)idx[w];
        const Arc *aw = arc + (idx[w] >> 32);
        for (uint32_t j0 = 0; j0 < nw; j0 += 64) {
            const uint32_t j = j0 + lane;
            bool in = false;
            uint32_t x = 0;
            if (j < nw) { const Arc a = aw[j]; in = (uint32_t)a.ul + li <= L; x = a.v; }
            if (in) {
                const int sx = slot_of(x);
                if (sx >= 0) atomicOr(&val[sx], 2u << 30);          // mark 1 -> 3, 2 stays: "not 1" either way
            }
            // lengths ascend along aw: once a lane falls outside, so do all later ones
            if (__ballot(j < nw && !in)) break;
        }
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
    }
    uint32_t red = 0;
    for (uint32_t i = lane; i < nv; i += 64) {
        const int s = slot_of(av[i].v);
        const uint32_t m = tr_load<BIG>(&val[s]);
        if ((m >> 30) != 1u && (m & 0x3fffffffu) == i) { del[b + i] = 1; ++red; }
    }
    red = (uint32_t)wave_incl_sum_i32((int)red, lane);
    if (lane == 63 && red) atomicAdd(n_reduced, red);
}

__global__ __launch_bounds__(64 * TR_WAVES) void reduce_kernel(const Arc *arc, const uint32_t *seq, const uint64_t *idx, uint32_t n_vtx,
                                                                uint32_t fuzz, uint8_t *del, uint32_t *n_reduced, uint32_t *big_list,
                                                                uint32_t *n_big) {
    __shared__ uint32_t s_key[TR_WAVES][TR_SLOTS];
    __shared__ uint32_t s_val[TR_WAVES][TR_SLOTS];      // mark << 30 | first arc index
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t v = blockIdx.x * TR_WAVES + wv;
    if (v >= n_vtx) return;
    const uint32_t nv = (uint32_t)idx[v];
    if (!nv) return;
    const uint64_t b = idx[v] >> 32;
    if (seq[v >> 1] >> 31) {                            // arcs of a deleted read (asg.c:157-160)
        for (uint32_t i = lane; i < nv; i += 64) del[b + i] = 1;
        if (lane == 0) atomicAdd(n_reduced, nv);
        return;
    }
    if (nv > TR_SLOTS / 2) {
        if (lane == 0) big_list[atomicAdd(n_big, 1u)] = v;
        return;
    }
    uint32_t *key = s_key[wv], *val = s_val[wv];
    for (int k = lane; k < TR_SLOTS; k += 64) { key[k] = TR_EMPTY; val[k] = 1u << 30 | 0x3fffffffu; }     // mark 1, no arc yet
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    reduce_vertex<false>(arc, idx, v, fuzz, del, n_reduced, key, val, TR_BITS, lane);
}

// Vertices with more arcs than the LDS table holds (a read end overlapped by more than a thousand reads none of which is
// contained in another: deep equal-length layouts, or the union of many --nsplit chunks' 60 rows per query): the same
// procedure, one wave per vertex, with a table of >= 4 slots per arc in global memory.
__device__ __forceinline__ int big_bits(uint32_t nv) { int b = TR_BITS + 1; while ((1u << b) < 4u * nv && b < 31) ++b; return b; }
__global__ void big_table_size_kernel(const uint64_t *idx, const uint32_t *big_list, uint32_t n_big, uint32_t *slots) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n_big) slots[k] = 1u << big_bits((uint32_t)idx[big_list[k]]);
}
__global__ __launch_bounds__(64 * TR_WAVES) void reduce_big_kernel(const Arc *arc, const uint64_t *idx, const uint32_t *big_list,
                                                                    uint32_t n_big, const uint64_t *tab_off, uint32_t fuzz, uint32_t *keys,
                                                                    uint32_t *vals, uint8_t *del, uint32_t *n_reduced) {
    const int lane = threadIdx.x & 63;
    const uint32_t k = blockIdx.x * TR_WAVES + (threadIdx.x >> 6);
    if (k >= n_big) return;
    const uint32_t v = big_list[k];
    const int bits = big_bits((uint32_t)idx[v]);
    uint32_t *key = keys + tab_off[k], *val = vals + tab_off[k];      // keys: filled with TR_EMPTY by the host
    for (uint32_t s = lane; s < (1u << bits); s += 64) val[s] = 1u << 30 | 0x3fffffffu;
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    reduce_vertex<true>(arc, idx, v, fuzz, del, n_reduced, key, val, bits, lane);
}

__global__ void apply_del_kernel(Arc *arc, const uint8_t *del, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n && del[i]) arc[i].ol_del |= 0x80000000u;
}

// asg.c:104-121: of several arcs v -> w only the first stays.  One thread per arc looks back along its vertex's list.
__global__ void dup_arcs_kernel(const Arc *arc, const uint64_t *idx, size_t n, uint8_t *del, uint32_t *count) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Arc a = arc[i];
    const uint64_t b = idx[a.ul >> 32] >> 32;
    bool dup = false;
    for (uint64_t k = b; k < i && !dup; ++k) dup = arc[k].v == a.v;
    del[i] = dup ? 1 : 0;
    if (dup) atomicAdd(count, 1u);
}
// asg.c:124-138: an arc u -> v needs its partner v^1 -> u^1
__global__ void unpaired_arcs_kernel(const Arc *arc, const uint64_t *idx, size_t n, uint8_t *del, uint32_t *count) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Arc a = arc[i];
    const uint32_t v = a.v ^ 1u, u = (uint32_t)(a.ul >> 32) ^ 1u;
    const uint64_t b = idx[v] >> 32;
    const uint32_t nv = (uint32_t)idx[v];
    bool found = false;
    for (uint32_t k = 0; k < nv && !found; ++k) found = arc[b + k].v == u;
    del[i] = found ? 0 : 1;
    if (!found) atomicAdd(count, 1u);
}

// ---------------------------------------------------------------------------------------------
// host-side helpers
// ---------------------------------------------------------------------------------------------
template <typename T>
void compact(DBuf<T> &a, size_t &n, const uint8_t *d_keep) {       // order-preserving
    if (!n) return;
    DBuf<uint32_t> idx(n);
    const size_t m = select_flagged_indices(d_keep, idx.p, n);
    DBuf<T> out(m ? m : 1);
    if (m) hipLaunchKernelGGL(gather_kernel<T>, grid1(m), dim3(WG), 0, stream(), a.p, idx.p, out.p, m);
    HIP_CHECK(hipGetLastError());
    a = std::move(out);
    n = m;
}
// the reference's order among equal keys: keys to the host, permutation back
template <typename T>
void reference_order(DBuf<T> &a, size_t n, const DBuf<uint64_t> &d_key) {
    if (n < 2) return;
    if (n >= (1ull << 32)) fail(HLMI_EINVAL, "overlap graph: more than 2^32 records in one sort");
    const std::vector<uint64_t> keys = d_key.download(n);
    std::vector<uint32_t> perm;
    reference_sort_order(keys, perm);
    DBuf<uint32_t> d_perm;
    d_perm.upload(perm);
    DBuf<T> out(n);
    hipLaunchKernelGGL(gather_kernel<T>, grid1(n), dim3(WG), 0, stream(), a.p, d_perm.p, out.p, n);
    HIP_CHECK(hipGetLastError());
    sync();
    a = std::move(out);
}

struct DevGraph {
    DBuf<Arc> arc;
    size_t n_arc = 0;
    DBuf<uint32_t> seq;
    size_t n_seq = 0;
    DBuf<uint64_t> idx;
    // asg_arc_rm + asg_arc_index: drop deleted arcs and arcs of deleted reads (order kept), rebuild the vertex index
    void cleanup() {
        if (n_arc) {
            DBuf<uint8_t> live(n_arc);
            hipLaunchKernelGGL(arc_live_kernel, grid1(n_arc), dim3(WG), 0, stream(), arc.p, n_arc, seq.p, live.p);
            compact(arc, n_arc, live.p);
        }
        index();
    }
    void index() {
        idx.alloc(2 * n_seq ? 2 * n_seq : 1);
        idx.zero();
        if (n_arc) hipLaunchKernelGGL(vertex_index_kernel, grid1(n_arc), dim3(WG), 0, stream(), arc.p, n_arc, idx.p);
        HIP_CHECK(hipGetLastError());
    }
};

__global__ void q_run_head_kernel(const Ovl *ovl, size_t n, uint8_t *head) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) head[i] = (i == 0 || ovl[i].q != ovl[i - 1].q) ? 1 : 0;
}

// windows of all reads from the current overlaps (a read without overlaps keeps {0, 0, alive}: calloc in hit.c:115)
void coverage_windows(const DBuf<Ovl> &ovl, size_t n, size_t n_reads, const GraphOpt &o, DBuf<ReadWin> &win) {
    win.alloc(n_reads ? n_reads : 1);
    win.zero();
    if (!n) return;
    DBuf<uint64_t> ends(2 * n);
    hipLaunchKernelGGL(win_ends_kernel, grid1(n), dim3(WG), 0, stream(), ovl.p, n, o.min_iden, ends.p);
    sort_keys_u64(ends, 2 * n, 0, 64);
    DBuf<uint8_t> head(n);
    hipLaunchKernelGGL(q_run_head_kernel, grid1(n), dim3(WG), 0, stream(), ovl.p, n, head.p);
    DBuf<uint32_t> run_head(n);
    const size_t n_runs = select_flagged_indices(head.p, run_head.p, n);
    hipLaunchKernelGGL(win_sweep_kernel, dim3(cdiv(n_runs, (size_t)(WG / 64))), dim3(WG), 0, stream(), ovl.p, run_head.p, n_runs,
                       ends.p, 2 * n, o.min_dp, win.p);
    HIP_CHECK(hipGetLastError());
}

uint32_t scan_total(const DBuf<uint32_t> &in, const DBuf<uint32_t> &excl, size_t n) {
    return n ? download_one(excl.p + (n - 1)) + download_one(in.p + (n - 1)) : 0u;
}
}  // namespace

void graph_device(const char *paf_path, const GraphOpt &o, const std::string &until, GraphState &out) {
    require_device();
    out = GraphState();
    out.paf = read_file(paf_path);
    const size_t nb = out.paf.size();
    if (!nb) return;
    KTimer kt_all("graph_device");
    DBuf<uint8_t> txt;
    txt.upload((const uint8_t *)out.paf.data(), nb);
    // ---- a9: lines -> rows ----------------------------------------------------------------------------------------
    // Line starts as 64-bit byte offsets (a merged PAF may exceed 4 GiB), found window by window (a launch and a selection
    // hold fewer than 2^32 items; HLMI_GRAPH_WINDOW_MB: small windows with small files, test hook): the windows' counts
    // first, then the offsets into one array.
    size_t pwin = (size_t)1 << 30;
    if (const char *e = hook("HLMI_GRAPH_WINDOW_MB")) pwin = (size_t)std::max(1, atoi(e)) << 20;
    DBuf<uint64_t> line_start;
    size_t n_lines = 0;
    {
        DBuf<uint8_t> flag(std::min(pwin, nb));
        DBuf<uint32_t> rel(std::min(pwin, nb));
        std::vector<size_t> per_window;
        for (size_t b = 0; b < nb; b += pwin) {
            const size_t n = std::min(pwin, nb - b);
            hipLaunchKernelGGL(line_start_kernel, grid1(n), dim3(WG), 0, stream(), txt.p, b, n, flag.p);
            per_window.push_back(select_flagged_indices(flag.p, rel.p, n));
            n_lines += per_window.back();
        }
        line_start.alloc(n_lines ? n_lines : 1);
        size_t at = 0, w = 0;
        for (size_t b = 0; b < nb; b += pwin, ++w) {
            const size_t n = std::min(pwin, nb - b);
            if (per_window.size() > 1) {                             // (one window: its offsets are still in `rel`)
                hipLaunchKernelGGL(line_start_kernel, grid1(n), dim3(WG), 0, stream(), txt.p, b, n, flag.p);
                select_flagged_indices(flag.p, rel.p, n);
            }
            if (per_window[w]) hipLaunchKernelGGL(add_base_kernel, grid1(per_window[w]), dim3(WG), 0, stream(), rel.p, per_window[w], (uint64_t)b, line_start.p + at);
            at += per_window[w];
        }
        HIP_CHECK(hipGetLastError());
        stat_set("graph_parse_windows", (double)per_window.size());
    }
    if (n_lines >= (1ull << 32)) fail(HLMI_EINVAL, "overlap graph: more than 2^32 lines in %s", paf_path);
    DBuf<PafRow> rows(n_lines);
    DBuf<uint32_t> kept(n_lines);
    size_t n_rows;
    {
        DBuf<uint8_t> ok(n_lines);
        hipLaunchKernelGGL(parse_rows_kernel, grid1(n_lines), dim3(WG), 0, stream(), txt.p, nb, line_start.p, n_lines, o.min_span,
                           o.min_match, rows.p, ok.p);
        HIP_CHECK(hipGetLastError());
        n_rows = select_flagged_indices(ok.p, kept.p, n_lines);
    }
    line_start.release();
    stat_set("graph_rows", (double)n_lines);
    stat_set("graph_rows_kept", (double)n_rows);
    if (!n_rows) return;
    if (n_rows >= (1ull << 30)) fail(HLMI_EINVAL, "overlap graph: more than 2^30 PAF rows");
    // ---- a9: read ids ----------------------------------------------------------------------------------------------
    const size_t n_occ = 2 * n_rows;
    DBuf<uint32_t> id_of_occ(n_occ);
    DBuf<uint64_t> ref_off;
    DBuf<uint32_t> ref_len;
    size_t n_reads = 0;
    for (int attempt = 0;; ++attempt) {
        DBuf<uint64_t> hash(n_occ);
        DBuf<uint32_t> occ(n_occ);
        hipLaunchKernelGGL(name_hash_kernel, grid1(n_occ), dim3(WG), 0, stream(), txt.p, rows.p, kept.p, n_occ,
                           0x243f6a8885a308d3ull + 0x9e3779b97f4a7c15ull * (uint64_t)attempt, hash.p, occ.p);
        sort_pairs_u64_u32(hash, occ, n_occ, 0, 64);
        DBuf<uint8_t> head(n_occ), first(n_occ);
        DBuf<uint32_t> collision(1);
        collision.zero();
        first.zero();
        hipLaunchKernelGGL(name_group_kernel, grid1(n_occ), dim3(WG), 0, stream(), txt.p, rows.p, kept.p, hash.p, occ.p, n_occ, head.p,
                           collision.p);
        HIP_CHECK(hipGetLastError());
        if (download_one(collision.p)) {
            if (attempt == 3) fail(HLMI_EINVAL, "overlap graph: read names collide under four hash seeds");
            continue;
        }
        DBuf<uint32_t> group_start(n_occ);
        n_reads = select_flagged_indices(head.p, group_start.p, n_occ);
        DBuf<uint32_t> head32(n_occ), group_of(n_occ), first32(n_occ), first_rank(n_occ);
        hipLaunchKernelGGL(widen_u8_kernel, grid1(n_occ), dim3(WG), 0, stream(), head.p, head32.p, n_occ);
        exclusive_scan_u32(head32.p, group_of.p, n_occ);
        hipLaunchKernelGGL(inclusive_from_exclusive_kernel, grid1(n_occ), dim3(WG), 0, stream(), group_of.p, head32.p, group_of.p, n_occ);
        hipLaunchKernelGGL(name_first_kernel, grid1(n_occ), dim3(WG), 0, stream(), occ.p, head.p, n_occ, first.p);
        hipLaunchKernelGGL(widen_u8_kernel, grid1(n_occ), dim3(WG), 0, stream(), first.p, first32.p, n_occ);
        exclusive_scan_u32(first32.p, first_rank.p, n_occ);
        hipLaunchKernelGGL(name_assign_kernel, grid1(n_occ), dim3(WG), 0, stream(), occ.p, group_of.p, group_start.p, first_rank.p, n_occ,
                           id_of_occ.p);
        ref_off.alloc(n_reads);
        ref_len.alloc(n_reads);
        hipLaunchKernelGGL(name_ref_kernel, grid1(n_occ), dim3(WG), 0, stream(), rows.p, kept.p, first.p, first_rank.p, n_occ, ref_off.p,
                           ref_len.p);
        HIP_CHECK(hipGetLastError());
        sync();
        break;
    }
    // ---- a9: overlaps, each row from both sides, in the reference's order ---------------------------------------------
    DBuf<Ovl> ovl;
    size_t n_ovl;
    {
        DBuf<uint32_t> cnt(n_rows), at(n_rows);
        hipLaunchKernelGGL(ovl_count_kernel, grid1(n_rows), dim3(WG), 0, stream(), id_of_occ.p, n_rows, cnt.p);
        exclusive_scan_u32(cnt.p, at.p, n_rows);
        n_ovl = scan_total(cnt, at, n_rows);
        ovl.alloc(n_ovl);
        DBuf<uint64_t> key(n_ovl);
        hipLaunchKernelGGL(ovl_fill_kernel, grid1(n_rows), dim3(WG), 0, stream(), rows.p, kept.p, id_of_occ.p, at.p, n_rows, ovl.p, key.p);
        HIP_CHECK(hipGetLastError());
        reference_order(ovl, n_ovl, key);                 // hit.c:104: by (read, start)
    }
    rows.release(); kept.release(); id_of_occ.release(); txt.release();
    stat_set("graph_reads", (double)n_reads);
    stat_set("graph_overlaps", (double)n_ovl);
    // ---- a10-a11: two rounds of read selection (main.c:119-142) ---------------------------------------------------------
    DBuf<ReadWin> win;
    DBuf<uint8_t> keep(n_ovl ? n_ovl : 1);
    coverage_windows(ovl, n_ovl, n_reads, o, win);
    hipLaunchKernelGGL(clip_kernel, grid1(n_ovl), dim3(WG), 0, stream(), ovl.p, n_ovl, win.p, o.min_span, keep.p);
    compact(ovl, n_ovl, keep.p);
    hipLaunchKernelGGL(crude_filter_kernel, grid1(n_ovl), dim3(WG), 0, stream(), ovl.p, n_ovl, win.p, (int)(o.max_hang * 1.5),
                       (int)(o.min_ovlp * .5), keep.p);
    compact(ovl, n_ovl, keep.p);
    {
        DBuf<ReadWin> win2;
        coverage_windows(ovl, n_ovl, n_reads, o, win2);
        hipLaunchKernelGGL(clip_kernel, grid1(n_ovl), dim3(WG), 0, stream(), ovl.p, n_ovl, win2.p, o.min_span, keep.p);
        compact(ovl, n_ovl, keep.p);
        hipLaunchKernelGGL(win_merge_kernel, grid1(n_reads), dim3(WG), 0, stream(), win.p, win2.p, n_reads);
    }
    // ---- a12: containment, unused reads, renumbering (hit.c:225-256) -------------------------------------------------------
    {
        DBuf<uint32_t> used(n_reads), alive(n_reads), new_id(n_reads);
        used.zero();
        if (n_ovl) hipLaunchKernelGGL(contained_kernel, grid1(n_ovl), dim3(WG), 0, stream(), ovl.p, n_ovl, win.p, o, used.p);
        hipLaunchKernelGGL(read_alive_kernel, grid1(n_reads), dim3(WG), 0, stream(), win.p, used.p, n_reads, alive.p);
        exclusive_scan_u32(alive.p, new_id.p, n_reads);
        const size_t n_left = scan_total(alive, new_id, n_reads);
        DBuf<ReadWin> win2(n_left ? n_left : 1);
        DBuf<uint64_t> off2(n_left ? n_left : 1);
        DBuf<uint32_t> len2(n_left ? n_left : 1);
        hipLaunchKernelGGL(squeeze_reads_kernel, grid1(n_reads), dim3(WG), 0, stream(), win.p, ref_off.p, ref_len.p, alive.p, new_id.p,
                           n_reads, win2.p, off2.p, len2.p);
        if (n_ovl) hipLaunchKernelGGL(renumber_kernel, grid1(n_ovl), dim3(WG), 0, stream(), ovl.p, n_ovl, alive.p, new_id.p, keep.p);
        HIP_CHECK(hipGetLastError());
        compact(ovl, n_ovl, keep.p);
        win = std::move(win2); ref_off = std::move(off2); ref_len = std::move(len2);
        n_reads = n_left;
    }
    stat_set("graph_reads_selected", (double)n_reads);
    stat_set("graph_overlaps_selected", (double)n_ovl);
    out.win = win.download(n_reads);
    {
        const std::vector<uint64_t> off = ref_off.download(n_reads);
        const std::vector<uint32_t> len = ref_len.download(n_reads);
        out.name.resize(n_reads);
        for (size_t i = 0; i < n_reads; ++i) out.name[i] = NameRef{off[i], len[i]};
    }
    if (until == "paf") out.ovl = ovl.download(n_ovl);
    if (until == "bed" || until == "paf") return;
    // ---- a13: string graph (asm.c:9-39, asg.c:57-80) ----------------------------------------------------------------------
    DevGraph g;
    g.n_seq = n_reads;
    g.seq.alloc(n_reads ? n_reads : 1);
    hipLaunchKernelGGL(seq_init_kernel, grid1(n_reads), dim3(WG), 0, stream(), win.p, n_reads, g.seq.p);
    g.arc.alloc(n_ovl ? n_ovl : 1);
    g.n_arc = n_ovl;
    if (n_ovl) {
        hipLaunchKernelGGL(make_arcs_kernel, grid1(n_ovl), dim3(WG), 0, stream(), ovl.p, n_ovl, win.p, o, g.arc.p, keep.p, g.seq.p);
        HIP_CHECK(hipGetLastError());
        compact(g.arc, g.n_arc, keep.p);
    }
    ovl.release();
    if (g.n_arc) {          // arcs of reads deleted while the arcs were made go first, then the order-defining sort
        DBuf<uint8_t> live(g.n_arc);
        hipLaunchKernelGGL(arc_live_kernel, grid1(g.n_arc), dim3(WG), 0, stream(), g.arc.p, g.n_arc, g.seq.p, live.p);
        compact(g.arc, g.n_arc, live.p);
    }
    if (g.n_arc) {
        DBuf<uint64_t> key(g.n_arc);
        hipLaunchKernelGGL(arc_key_kernel, grid1(g.n_arc), dim3(WG), 0, stream(), g.arc.p, g.n_arc, key.p);
        reference_order(g.arc, g.n_arc, key);             // asg.c:27-42: by (vertex, arc length)
    }
    g.index();
    stat_set("graph_arcs", (double)g.n_arc);
    // ---- a14: transitive reduction, then duplicate and unpaired arcs (asg.c:104-193) ----------------------------------------
    const uint32_t n_vtx = (uint32_t)(2 * n_reads);
    uint32_t n_reduced = 0;
    if (g.n_arc) {
        DBuf<uint8_t> del(g.n_arc);
        DBuf<uint32_t> counters(2), big(n_vtx ? n_vtx : 1);
        del.zero();
        counters.zero();
        {
            KTimer kt("graph_reduce");
            hipLaunchKernelGGL(reduce_kernel, dim3(cdiv(n_vtx, (size_t)TR_WAVES)), dim3(64 * TR_WAVES), 0, stream(), g.arc.p, g.seq.p,
                               g.idx.p, n_vtx, (uint32_t)o.gap_fuzz, del.p, counters.p, big.p, counters.p + 1);
        }
        HIP_CHECK(hipGetLastError());
        const uint32_t n_big = download_one(counters.p + 1);
        if (n_big) {
            KTimer kt("graph_reduce_big");
            DBuf<uint32_t> slots(n_big);
            DBuf<uint64_t> tab_off(n_big);
            hipLaunchKernelGGL(big_table_size_kernel, grid1(n_big), dim3(WG), 0, stream(), g.idx.p, big.p, n_big, slots.p);
            exclusive_scan_u32_to_u64(slots.p, tab_off.p, n_big);
            const uint64_t total = download_one(tab_off.p + (n_big - 1)) + download_one(slots.p + (n_big - 1));
            DBuf<uint32_t> keys(total), vals(total);
            keys.fill_ff();
            hipLaunchKernelGGL(reduce_big_kernel, dim3(cdiv(n_big, (size_t)TR_WAVES)), dim3(64 * TR_WAVES), 0, stream(), g.arc.p, g.idx.p,
                               big.p, n_big, tab_off.p, (uint32_t)o.gap_fuzz, keys.p, vals.p, del.p, counters.p);
            HIP_CHECK(hipGetLastError());
            stat_set("graph_big_table_slots", (double)total);
        }
        n_reduced = download_one(counters.p);
        stat_set("graph_arcs_reduced", (double)n_reduced);
        stat_set("graph_big_vertices", (double)n_big);
        if (n_reduced) {
            hipLaunchKernelGGL(apply_del_kernel, grid1(g.n_arc), dim3(WG), 0, stream(), g.arc.p, del.p, g.n_arc);
            g.cleanup();
            for (int pass = 0; pass < 2 && g.n_arc; ++pass) {     // asg_symm: duplicates, then arcs without their reverse
                DBuf<uint8_t> d2(g.n_arc);
                counters.zero();
                if (pass == 0) hipLaunchKernelGGL(dup_arcs_kernel, grid1(g.n_arc), dim3(WG), 0, stream(), g.arc.p, g.idx.p, g.n_arc, d2.p, counters.p);
                else hipLaunchKernelGGL(unpaired_arcs_kernel, grid1(g.n_arc), dim3(WG), 0, stream(), g.arc.p, g.idx.p, g.n_arc, d2.p, counters.p);
                HIP_CHECK(hipGetLastError());
                if (download_one(counters.p)) {
                    hipLaunchKernelGGL(apply_del_kernel, grid1(g.n_arc), dim3(WG), 0, stream(), g.arc.p, d2.p, g.n_arc);
                    g.cleanup();
                }
            }
        }
    }
    out.arc = g.arc.download(g.n_arc);
    out.seq_len = g.seq.download(n_reads);
    out.have_graph = true;
    out.symmetric = n_reduced != 0;
}

}  // namespace hlmi
