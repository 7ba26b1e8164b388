// row_text.hip - the stage's final rows as text, written by the GPU (filter_overlap_slr2.py:138-151).
//
// A short-read call keeps millions of rows per pass (C4s: 3.2 M per call); turning them into the 14 tab-separated columns
// on the host threads was a third of such a step.  Here a thread per kept row does what paf_io.cpp:format_scored_row does:
//   score  = 0.4 mc / ((ql + tl) / 2) + 0.6 mc / len,   score2 = 1 - sum_of_X_digits / mc,   score3 = mc / len
// as IEEE doubles in the reference's evaluation order (the library is built with -ffp-contract=off; double division on
// gfx950 is correctly rounded), "%.4f" by the exact integer conversion of paf_io.cpp:format_fixed4 (v x 10^4 rounded to
// nearest, ties to even), the `float(score2) < iden` test on the four printed decimals, the sort key of column 12 as an
// integer.  A value outside that conversion's range (negative, non-finite, >= 2^40) is left to the host formatter: the row
// is flagged, never guessed.  Two passes: lengths (+ key), exclusive scan, bytes.
#include "row_text.h"

#include "dev_prims.h"

namespace hlmi {

namespace {
constexpr int WG = 256;
inline dim3 grid1(size_t n) { return dim3(cdiv(n ? n : 1, WG)); }

constexpr uint32_t ROW_DROPPED = 0u, ROW_TO_HOST = 0xffffffffu;

// "%.4f" of a finite v in [0, 2^40): q = v x 10^4 rounded to nearest even.  false: not in that range.
__device__ __forceinline__ bool fixed4_q(double v, uint64_t &q) {
    const uint64_t bits = (uint64_t)__double_as_longlong(v);
    const int be = (int)((bits >> 52) & 0x7ff);
    if ((bits >> 63) || be == 0x7ff || !(v < 1099511627776.0)) return false;
    uint64_t m = bits & ((1ull << 52) - 1);
    int e = be - 1075;                                       // v = m x 2^e
    if (be) m |= 1ull << 52; else e = -1074;
    if (e >= 0) { q = (m << e) * 10000ull; return true; }
    const unsigned __int128 N = (unsigned __int128)m * 10000u;        // < 2^67
    const int sh = -e;
    if (sh > 68) { q = 0; return true; }
    q = (uint64_t)(N >> sh);
    const unsigned __int128 rem = N & (((unsigned __int128)1 << sh) - 1), half = (unsigned __int128)1 << (sh - 1);
    if (rem > half || (rem == half && (q & 1))) ++q;
    return true;
}
__device__ __forceinline__ int dec_len(uint64_t v) {
    int n = 1;
    while (v >= 10) { v /= 10; ++n; }
    return n;
}
__device__ __forceinline__ char *put_dec(char *p, uint64_t v) {
    const int n = dec_len(v);
    for (int i = n - 1; i >= 0; --i) { p[i] = (char)('0' + (int)(v % 10)); v /= 10; }
    return p + n;
}
__device__ __forceinline__ int fixed4_len(uint64_t q) { return dec_len(q / 10000u) + 5; }
__device__ __forceinline__ char *put_fixed4(char *p, uint64_t q) {
    p = put_dec(p, q / 10000u);
    uint32_t fp = (uint32_t)(q % 10000u);
    *p++ = '.';
    p[3] = (char)('0' + fp % 10); fp /= 10;
    p[2] = (char)('0' + fp % 10); fp /= 10;
    p[1] = (char)('0' + fp % 10); fp /= 10;
    p[0] = (char)('0' + fp);
    return p + 4;
}

struct Scores { uint64_t q1, q2, q3; };
// the three scores of a row as integers x 10^4; 0: dropped by the identity test, 1: kept, 2: the host formats this row
__device__ __forceinline__ int row_scores(const PafRec &r, uint32_t xsum, double iden, Scores &s) {
    const double mc = (double)r.nmatch, ln = (double)r.blen;
    const double mlen = (double)((uint64_t)r.qlen + r.tlen) / 2.0;
    const double t1 = mc / mlen, t2 = mc / ln;
    const double a = 0.4 * t1, b = 0.6 * t2;
    const double score = a + b;
    const double mis = (double)xsum / mc;
    const double score2 = 1.0 - mis;
    if (!fixed4_q(score, s.q1) || !fixed4_q(score2, s.q2) || !fixed4_q(t2, s.q3)) return 2;
    if (s.q2 >= (1ull << 53)) return 2;
    const double shown = (double)s.q2 / 10000.0;             // float("0.9876"): the correctly rounded 9876 / 10^4
    return shown < iden ? 0 : 1;
}

struct RowTextArgs {
    const PafRec *recs;
    const uint32_t *idx, *xsum;
    size_t n;
    const char *names;
    const uint64_t *name_off;
    double iden;
};

__global__ __launch_bounds__(WG) void row_len_kernel(RowTextArgs a, uint32_t *len, uint32_t *key) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const PafRec r = a.recs[a.idx[i]];
    Scores s;
    const int st = row_scores(r, a.xsum[i], a.iden, s);
    if (st != 1) { len[i] = st ? ROW_TO_HOST : ROW_DROPPED; key[i] = 0xffffffffu; return; }
    const uint32_t nq = (uint32_t)(a.name_off[r.qid + 1] - a.name_off[r.qid]), nt = (uint32_t)(a.name_off[r.tid + 1] - a.name_off[r.tid]);
    // 14 columns, a TAB behind every one of them (the 14th too: slr2:151), no newline (the writer adds it)
    len[i] = nq + nt + (uint32_t)(dec_len(r.qlen) + dec_len(r.qs) + dec_len(r.qe) + 1 + dec_len(r.tlen) + dec_len(r.ts) + dec_len(r.te) +
                                   dec_len(r.nmatch) + dec_len(r.blen) + fixed4_len(s.q1) + fixed4_len(s.q2) + fixed4_len(s.q3)) + 14u;
    key[i] = s.q1 < (uint64_t)SCORE_KEY_LIMIT ? (uint32_t)s.q1 : 0xffffffffu;
}

__global__ __launch_bounds__(WG) void row_write_kernel(RowTextArgs a, const uint32_t *len, const uint64_t *off, char *text) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= a.n || len[i] == ROW_DROPPED || len[i] == ROW_TO_HOST) return;
    const PafRec r = a.recs[a.idx[i]];
    Scores s;
    row_scores(r, a.xsum[i], a.iden, s);
    char *p = text + off[i];
    auto name = [&](uint32_t id) {
        const char *src = a.names + a.name_off[id];
        const uint32_t n = (uint32_t)(a.name_off[id + 1] - a.name_off[id]);
        for (uint32_t k = 0; k < n; ++k) p[k] = src[k];
        p += n;
    };
    name(r.qid); *p++ = '\t';
    p = put_dec(p, r.qlen); *p++ = '\t';
    p = put_dec(p, r.qs); *p++ = '\t';
    p = put_dec(p, r.qe); *p++ = '\t';
    *p++ = (r.flags & PF_REV) ? '-' : '+'; *p++ = '\t';
    name(r.tid); *p++ = '\t';
    p = put_dec(p, r.tlen); *p++ = '\t';
    p = put_dec(p, r.ts); *p++ = '\t';
    p = put_dec(p, r.te); *p++ = '\t';
    p = put_dec(p, r.nmatch); *p++ = '\t';
    p = put_dec(p, r.blen); *p++ = '\t';
    p = put_fixed4(p, s.q1); *p++ = '\t';
    p = put_fixed4(p, s.q2); *p++ = '\t';
    p = put_fixed4(p, s.q3); *p++ = '\t';
}

// len with the two flag values read as 0
__global__ __launch_bounds__(WG) void row_len_clean_kernel(const uint32_t *len, uint32_t *clean, size_t n) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) clean[i] = len[i] == ROW_TO_HOST ? 0u : len[i];
}
}  // namespace

void DevNames::upload(const std::vector<std::string> &names_by_id) {
    std::vector<uint64_t> off(names_by_id.size() + 1, 0);
    for (size_t i = 0; i < names_by_id.size(); ++i) off[i + 1] = off[i] + names_by_id[i].size();
    std::string all;
    all.reserve(off.back());
    for (const auto &s : names_by_id) all.append(s);
    text.alloc(all.size() ? all.size() : 1);
    if (all.size()) HIP_CHECK(hipMemcpyAsync(text.p, all.data(), all.size(), hipMemcpyHostToDevice, stream()));
    this->off.upload(off);
    sync();
    n = names_by_id.size();
}

void format_rows_device(const PafRec *d_recs, const std::vector<uint32_t> &idx, const std::vector<uint32_t> &xsum, const DevNames &names,
                        double iden, RowText &out) {
    const size_t n = idx.size();
    out.at.assign(n, 0); out.len.assign(n, 0); out.key.assign(n, 0xffffffffu);
    out.text.clear();
    if (!n) return;
    DBuf<uint32_t> d_idx, d_x, d_len(n), d_key(n), d_clean(n);
    d_idx.upload(idx);
    d_x.upload(xsum);
    RowTextArgs a{d_recs, d_idx.p, d_x.p, n, names.text.p, names.off.p, iden};
    DBuf<uint64_t> d_off(n);
    {
        KTimer kt("row_text");
        hipLaunchKernelGGL(row_len_kernel, grid1(n), dim3(WG), 0, stream(), a, d_len.p, d_key.p);
        hipLaunchKernelGGL(row_len_clean_kernel, grid1(n), dim3(WG), 0, stream(), d_len.p, d_clean.p, n);
        exclusive_scan_u32_to_u64(d_clean.p, d_off.p, n);
    }
    HIP_CHECK(hipGetLastError());
    out.len = d_len.download(n);
    out.key = d_key.download(n);
    const std::vector<uint64_t> off = d_off.download(n);
    const uint64_t total = off[n - 1] + (out.len[n - 1] == ROW_TO_HOST ? 0u : out.len[n - 1]);
    DBuf<char> d_text(total ? total : 1);
    {
        KTimer kt("row_text");
        hipLaunchKernelGGL(row_write_kernel, grid1(n), dim3(WG), 0, stream(), a, d_len.p, d_off.p, d_text.p);
    }
    HIP_CHECK(hipGetLastError());
    out.text.resize(total);
    if (total) HIP_CHECK(hipMemcpyAsync(&out.text[0], d_text.p, total, hipMemcpyDeviceToHost, stream()));
    sync();
    for (size_t i = 0; i < n; ++i) out.at[i] = off[i];
}

}  // namespace hlmi
