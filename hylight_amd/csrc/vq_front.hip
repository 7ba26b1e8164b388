// vq_front.hip - SURVEY 8f rank 3, STARTED: the front of the SAVAGE / ViralQuasispecies overlap-graph assembler that
// consumes the path's 13-column overlaps file (tools/HaploConduct/src).  Two pieces:
//   * the text of record: which lines of the overlaps file become edge candidates (EdgeCalculator.cpp:561-666 in front
//     of process_overlaps, Overlap.h:37-72,196-203) - host, integer and string rules only;
//   * transitive edges by intersection of sorted adjacency lists (GraphAlgos.cpp:746-795,938-993) - device.
// Between the two the reference scores every candidate from the reads' bases and qualities (EdgeCalculator.cpp:26-139,
// log / pow / exp thresholds) and orients it (Edge.h): not built.  The reference needs Boost and cannot be compiled in
// this image, so both pieces are checked against oracle/vq.py only: PARITY UNPINNED.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <unordered_map>
#include <string_view>
#include <cstring>
#include <string>
#include <vector>

#include "common.h"
#include "dev_prims.h"
#include "graph.h"
#include "paf_io.h"

namespace hlmi {

// ---------------------------------------------------------------------------------------------
// 13-column overlaps -> edge candidates
// ---------------------------------------------------------------------------------------------
static bool one_of(const std::string &s, const char *set) { return s.size() == 1 && strchr(set, s[0]) != nullptr; }
static std::string without(std::string s, const char *drop) {
    std::string r;
    for (char c : s) if (!strchr(drop, c)) r.push_back(c);
    return r;
}

void vq_parse_overlaps(const char *path, uint32_t min_len, uint32_t min_perc, int relax_pe, uint64_t max_overlaps,
                       hlmi_vq_overlap *out, uint64_t cap, uint64_t *n_out, uint64_t *n_nonedge, uint64_t *n_skipped) {
    const std::string data = read_file(path);
    uint64_t kept = 0, nonedge = 0, skipped = 0, i = 0;
    size_t pos = 0;
    std::vector<std::string> f;
    while (pos < data.size() && i < max_overlaps) {          // (getline: a last line without '\n' counts)
        size_t e = data.find('\n', pos);
        if (e == std::string::npos) e = data.size();
        size_t b = pos;
        pos = e + 1;
        ++i;
        while (b < e && (data[b] == '\t' || data[b] == ' ')) ++b;                  // boost::trim_if(line, "\t ")
        while (e > b && (data[e - 1] == '\t' || data[e - 1] == ' ')) --e;
        f.clear();
        if (e > b) {                                                               // split at tabs (an empty line: no field)
            size_t p = b;
            for (;;) {
                size_t t = data.find('\t', p);
                if (t == std::string::npos || t >= e) { f.emplace_back(data, p, e - p); break; }
                f.emplace_back(data, p, t - p);
                p = t + 1;
            }
        }
        if (f.size() != 13) { ++skipped; continue; }                               // "incorrect overlap; skipping"
        hlmi_vq_overlap o{};
        o.id1 = strtoul(f[0].c_str(), nullptr, 0);
        o.id2 = strtoul(f[1].c_str(), nullptr, 0);
        const int pos1 = atoi(f[2].c_str()), pos2 = f[3] == "-" ? 0 : atoi(f[3].c_str());
        const int perc1 = atoi(f[7].c_str()), perc2 = f[3] == "-" ? 0 : atoi(f[8].c_str());
        const int len1 = atoi(f[9].c_str()), len2 = f[3] == "-" ? 0 : atoi(f[10].c_str());
        // the checks of Overlap.h:59-72 (the reference exits / asserts)
        const std::string ord = f[4].size() == 1 ? f[4] : without(f[4], " ");
        const std::string ori1 = f[5].size() == 1 ? f[5] : without(f[5], " "), ori2 = f[6].size() == 1 ? f[6] : without(f[6], " ");
        const std::string ty1 = f[11].size() == 1 ? f[11] : without(f[11], "\n\t "), ty2 = f[12].size() == 1 ? f[12] : without(f[12], "\n\t ");
        if (pos1 < 0 || pos2 < 0 || perc1 < 0 || perc1 > 100 || perc2 < 0 || perc2 > 100 || len1 < 0 || len2 < 0 ||
            !one_of(ori1, "+-") || !one_of(ori2, "+-") || !one_of(ty1, "sp") || !one_of(ty2, "sp") || !one_of(ord, "12-") ||
            ((ty1 == "s" || ty2 == "s") != (ord == "-")))
            fail(HLMI_EINVAL, "%s: line %llu is not a valid overlap (Overlap.h:59-170)", path, (unsigned long long)i);
        o.pos1 = (uint32_t)pos1; o.pos2 = (uint32_t)pos2; o.perc1 = (uint32_t)perc1; o.perc2 = (uint32_t)perc2;
        o.len1 = (uint32_t)len1; o.len2 = (uint32_t)len2;
        o.ord = ord[0]; o.ori1 = ori1[0]; o.ori2 = ori2[0]; o.type1 = ty1[0]; o.type2 = ty2[0];
        if (o.id1 == o.id2) { ++skipped; continue; }
        const unsigned perc = o.perc2 > 0 ? (unsigned)(0.5 * (double)(o.perc1 + o.perc2)) : o.perc1;     // Overlap.h:196-203
        const bool ss = o.type1 == 's' && o.type2 == 's', anyp = o.type1 == 'p' || o.type2 == 'p';
        bool edge = false, decided = false;
        if (o.len1 >= min_len && ss) { decided = true; edge = perc >= min_perc; }
        else if ((double)o.len1 >= 0.5 * (double)min_len && (double)o.len2 >= 0.5 * (double)min_len && anyp) { decided = true; edge = perc >= min_perc; }
        else if (relax_pe && o.len1 + o.len2 >= min_len && anyp) { decided = true; edge = perc >= min_perc; }
        if (!decided) { ++nonedge; continue; }                // written back to nonedge_overlaps.txt
        if (!edge) { ++skipped; continue; }                   // long enough, identity too low: dropped without a trace
        if (out && kept < cap) out[kept] = o;
        ++kept;
    }
    *n_out = kept; *n_nonedge = nonedge; *n_skipped = skipped;
}

// ---------------------------------------------------------------------------------------------
// transitive edges
// ---------------------------------------------------------------------------------------------
namespace {
constexpr int WG = 256;
constexpr int WAVES = WG / 64;
constexpr uint32_t SET_CAP = 2048;                 // LDS hash slots per wave: vertices with up to SET_CAP / 2 out-edges
constexpr uint32_t EMPTY = 0xffffffffu;
inline dim3 grid1(size_t n) { return dim3((unsigned)cdiv(n ? n : 1, (size_t)WG)); }

__global__ void edge_keys_kernel(const uint32_t *a, const uint32_t *b, const uint32_t *ids, size_t n, uint64_t *key, uint32_t *val) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t k = ids ? ids[i] : (uint32_t)i;
    key[i] = (uint64_t)a[k] << 32 | b[k];
    val[i] = k;
}
// first position whose key's high word is >= v, for v = 0 .. n_vertices
__global__ void offsets_kernel(const uint64_t *key, size_t n, uint32_t n_vertices, uint32_t *off) {
    size_t v = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (v > n_vertices) return;
    size_t lo = 0, hi = n;
    while (lo < hi) {
        const size_t mid = (lo + hi) >> 1;
        if ((uint32_t)(key[mid] >> 32) < (uint32_t)v) lo = mid + 1; else hi = mid;
    }
    off[v] = (uint32_t)lo;
}

__device__ __forceinline__ uint32_t slot_of(uint32_t w) { return (w * 2654435761u) >> 21; }        // 11 bits

// One wave per vertex u.  out(u) goes into an open-addressing table in LDS; for every out-edge u -> v the lanes stream
// in(v) and ask the table.  Vertices with more out-edges than half the table are left to big_kernel.
__global__ __launch_bounds__(WG) void trans_kernel(const uint64_t *okey, const uint32_t *oval, const uint32_t *ooff,
                                                   const uint64_t *ikey, const uint32_t *ioff, uint32_t n_vertices,
                                                   uint8_t *flag, uint32_t *big_list, uint32_t *n_big) {
    __shared__ uint32_t s_set[WAVES][SET_CAP];
    const int lane = threadIdx.x & 63;
    uint32_t *set = s_set[threadIdx.x >> 6];
    const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
    const size_t n_waves = ((size_t)gridDim.x * blockDim.x) >> 6;
    for (size_t u = wave; u < n_vertices; u += n_waves) {
        const uint32_t b = ooff[u], e = ooff[u + 1];
        if (b == e) continue;
        if (e - b > SET_CAP / 2) { if (lane == 0) big_list[atomicAdd(n_big, 1u)] = (uint32_t)u; continue; }
        for (uint32_t k = (uint32_t)lane; k < SET_CAP; k += 64) set[k] = EMPTY;
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
        for (uint32_t k = b + (uint32_t)lane; k < e; k += 64) {
            const uint32_t w = (uint32_t)okey[k];
            uint32_t s = slot_of(w);
            for (;;) {
                const uint32_t old = atomicCAS(&set[s], EMPTY, w);
                if (old == EMPTY || old == w) break;
                s = (s + 1) & (SET_CAP - 1);
            }
        }
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
        for (uint32_t k = b; k < e; ++k) {
            const uint32_t v = (uint32_t)okey[k];
            const uint32_t ib = ioff[v], ie = ioff[v + 1];
            bool hit = false;
            for (uint32_t j0 = ib; j0 < ie && !hit; j0 += 64) {
                const uint32_t j = j0 + (uint32_t)lane;
                bool mine = false;
                if (j < ie) {
                    const uint32_t w = (uint32_t)ikey[j];           // w -> v
                    uint32_t s = slot_of(w);
                    for (;;) {
                        const uint32_t x = set[s];
                        if (x == w) { mine = true; break; }
                        if (x == EMPTY) break;
                        s = (s + 1) & (SET_CAP - 1);
                    }
                }
                hit = __any(mine);
            }
            if (lane == 0) flag[oval[k]] = hit ? 1 : 0;
        }
        __builtin_amdgcn_wave_barrier();
    }
}
// the out-edges of the vertices in big_list: one thread per in-neighbour w of v, binary search of w in out(u)
__global__ __launch_bounds__(WG) void trans_big_kernel(const uint64_t *okey, const uint32_t *oval, const uint32_t *ooff,
                                                       const uint64_t *ikey, const uint32_t *ioff, const uint32_t *big_list,
                                                       uint32_t n_big, uint8_t *flag) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
    const size_t n_waves = ((size_t)gridDim.x * blockDim.x) >> 6;
    for (size_t bi = 0; bi < n_big; ++bi) {
        const uint32_t u = big_list[bi];
        const uint32_t b = ooff[u], e = ooff[u + 1];
        for (size_t k = b + wave; k < e; k += n_waves) {
            const uint32_t v = (uint32_t)okey[k];
            const uint32_t ib = ioff[v], ie = ioff[v + 1];
            bool hit = false;
            for (uint32_t j0 = ib; j0 < ie && !hit; j0 += 64) {
                const uint32_t j = j0 + (uint32_t)lane;
                bool mine = false;
                if (j < ie) {
                    const uint32_t w = (uint32_t)ikey[j];
                    uint32_t lo = b, hi = e;
                    while (lo < hi) {
                        const uint32_t mid = (lo + hi) >> 1;
                        if ((uint32_t)okey[mid] < w) lo = mid + 1; else hi = mid;
                    }
                    mine = lo < e && (uint32_t)okey[lo] == w;
                }
                hit = __any(mine);
            }
            if (lane == 0) flag[oval[k]] = hit ? 1 : 0;
        }
    }
}
__global__ void spread_flags_kernel(const uint8_t *flag_sub, const uint32_t *ids, size_t n, uint8_t *out) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n && flag_sub[i]) out[ids[i]] |= 1;
}
// branch reduction (GraphAlgos.cpp:970-993): the longest transitive out-edge per source / in-edge per target ...
__global__ void trans_max_kernel(const uint32_t *src, const uint32_t *dst, const uint32_t *len, const uint8_t *flag, size_t n,
                                 uint32_t *max_out, uint32_t *max_in) {
    size_t k = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (k >= n || !(flag[k] & 1)) return;
    atomicMax(&max_out[src[k]], len[k] + 1);          // (+1: 0 means "no transitive edge here")
    atomicMax(&max_in[dst[k]], len[k] + 1);
}
// ... and every edge no longer than it at the same end is scheduled for deletion
__global__ void branch_mark_kernel(const uint32_t *src, const uint32_t *dst, const uint32_t *len, size_t n, const uint32_t *max_out,
                                   const uint32_t *max_in, uint8_t *flag) {
    size_t k = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (k >= n) return;
    if (len[k] + 1 <= max_out[src[k]] || len[k] + 1 <= max_in[dst[k]]) flag[k] |= 2;
}
}  // namespace

void vq_transitive_edges(uint32_t n_vertices, uint64_t n_edges, const uint32_t *src, const uint32_t *dst, const uint32_t *ovlen,
                         int remove_trans, uint8_t *flags, uint64_t *n_transitive) {
    if (remove_trans < 1 || remove_trans > 3) fail(HLMI_EINVAL, "remove_trans must be 1, 2 or 3");
    if (n_edges >= (1ull << 32)) fail(HLMI_EINVAL, "more than 2^32 edges");
    *n_transitive = 0;
    if (!n_edges) return;
    for (uint64_t k = 0; k < n_edges; ++k)
        if (src[k] >= n_vertices || dst[k] >= n_vertices) fail(HLMI_EINVAL, "edge %llu names a vertex >= n_vertices", (unsigned long long)k);
    const size_t E = (size_t)n_edges;
    DBuf<uint32_t> d_src, d_dst;
    d_src.upload(src, E);
    d_dst.upload(dst, E);
    DBuf<uint8_t> result(E);
    result.zero();
    DBuf<uint32_t> ids;                    // edges of the current graph (round 1: all)
    size_t n_cur = E;
    size_t found = 0;
    for (int round = 1; round <= remove_trans && n_cur; ++round) {
        DBuf<uint64_t> okey(n_cur), ikey(n_cur);
        DBuf<uint32_t> oval(n_cur), ival(n_cur), ooff((size_t)n_vertices + 1), ioff((size_t)n_vertices + 1);
        const uint32_t *cur = round == 1 ? nullptr : ids.p;
        hipLaunchKernelGGL(edge_keys_kernel, grid1(n_cur), dim3(WG), 0, stream(), d_src.p, d_dst.p, cur, n_cur, okey.p, oval.p);
        hipLaunchKernelGGL(edge_keys_kernel, grid1(n_cur), dim3(WG), 0, stream(), d_dst.p, d_src.p, cur, n_cur, ikey.p, ival.p);
        sort_pairs_u64_u32(okey, oval, n_cur, 0, 64);
        sort_pairs_u64_u32(ikey, ival, n_cur, 0, 64);
        hipLaunchKernelGGL(offsets_kernel, grid1((size_t)n_vertices + 1), dim3(WG), 0, stream(), okey.p, n_cur, n_vertices, ooff.p);
        hipLaunchKernelGGL(offsets_kernel, grid1((size_t)n_vertices + 1), dim3(WG), 0, stream(), ikey.p, n_cur, n_vertices, ioff.p);
        // flags of this round, indexed by the ORIGINAL edge number (oval holds those)
        DBuf<uint8_t> fl(E);
        fl.zero();
        DBuf<uint32_t> big(n_vertices ? n_vertices : 1), n_big(1);
        n_big.zero();
        const unsigned nb = (unsigned)std::min<size_t>(cdiv((size_t)n_vertices, (size_t)WAVES), 256 * 16);
        hipLaunchKernelGGL(trans_kernel, dim3(nb ? nb : 1), dim3(WG), 0, stream(), okey.p, oval.p, ooff.p, ikey.p, ioff.p, n_vertices,
                           fl.p, big.p, n_big.p);
        HIP_CHECK(hipGetLastError());
        const uint32_t hb = download_one(n_big.p);
        if (hb) hipLaunchKernelGGL(trans_big_kernel, dim3(256 * 8), dim3(WG), 0, stream(), okey.p, oval.p, ooff.p, ikey.p, ioff.p,
                                   big.p, hb, fl.p);
        HIP_CHECK(hipGetLastError());
        // the edges found are the next round's graph, and (after the last round) the answer
        DBuf<uint32_t> next(E);
        found = select_flagged_indices(fl.p, next.p, E);
        ids = std::move(next);
        n_cur = found;
    }
    if (found) {
        DBuf<uint8_t> ones(found);
        ones.fill_ff();
        hipLaunchKernelGGL(spread_flags_kernel, grid1(found), dim3(WG), 0, stream(), ones.p, ids.p, found, result.p);
    }
    if (remove_trans == 1 && ovlen && found) {
        DBuf<uint32_t> d_len, max_out(n_vertices), max_in(n_vertices);
        d_len.upload(ovlen, E);
        max_out.zero();
        max_in.zero();
        hipLaunchKernelGGL(trans_max_kernel, grid1(E), dim3(WG), 0, stream(), d_src.p, d_dst.p, d_len.p, result.p, E, max_out.p, max_in.p);
        hipLaunchKernelGGL(branch_mark_kernel, grid1(E), dim3(WG), 0, stream(), d_src.p, d_dst.p, d_len.p, E, max_out.p, max_in.p, result.p);
    }
    HIP_CHECK(hipGetLastError());
    const std::vector<uint8_t> h = result.download(E);
    memcpy(flags, h.data(), E);
    *n_transitive = found;
}


// ---------------------------------------------------------------------------------------------
// quality-aware overlap score of single-single overlaps (EdgeCalculator.cpp:26-139, :186-222)
// ---------------------------------------------------------------------------------------------
// The score of one overlap is exp(mean over the overlapping positions of log p), p = probability that the two bases are
// copies of one base given their phred qualities; the sum runs over the positions in sequence order in double precision.
// log and pow come from the HOST's libm, as in the reference: p depends on the two qualities and on match / mismatch only,
// so the host tabulates log p per (quality, quality) pair and the kernel - one thread per overlap, positions in order -
// adds table entries: the same doubles in the same order as the reference's loop.  exp and the final division are done on
// the host again.
namespace {
constexpr int NQ = 94;                         // phred characters '!' .. '~'
struct VqReadRef { uint64_t off; uint32_t len; uint32_t pad; };
struct VqScoreOut { double total; uint32_t len; uint32_t mismatches; uint32_t status; uint32_t pad; };   // status 1: zero score
__device__ __forceinline__ int vq_code(uint8_t c, bool comp) {        // A C G T -> 0..3 (complemented), N -> 4, else 5
    int v;
    switch (c) { case 'A': v = 0; break; case 'C': v = 1; break; case 'G': v = 2; break; case 'T': v = 3; break; case 'N': return 4; default: return 5; }
    return comp ? 3 - v : v;
}
__global__ void vq_score_kernel(const uint8_t *seq, const uint8_t *qual, const VqReadRef *reads, const uint32_t *idx1, const uint32_t *idx2,
                                const uint32_t *pos1, const uint8_t *ori, size_t n, const double *log_match, const double *log_mis,
                                const uint8_t *bad_match, const uint8_t *bad_mis, uint32_t min_read_len, VqScoreOut *out) {
    const size_t k = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (k >= n) return;
    const VqReadRef a = reads[idx1[k]], b = reads[idx2[k]];
    const bool f1 = ori[k] & 1, f2 = (ori[k] >> 1) & 1;
    const uint32_t pos = pos1[k];
    VqScoreOut o{0.0, 0, 0, 0, 0};
    if (pos >= a.len || a.len < min_read_len || b.len < min_read_len) { o.status = 1; out[k] = o; return; }
    const uint32_t L = min(a.len - pos, b.len);
    for (uint32_t i = 0; i < L; ++i) {
        // position j of the oriented read: the read itself, or its reverse complement with the qualities reversed (Read.h:158-200)
        const uint64_t ia = a.off + (f1 ? (uint64_t)(pos + i) : (uint64_t)(a.len - 1 - (pos + i)));
        const uint64_t ib = b.off + (f2 ? (uint64_t)i : (uint64_t)(b.len - 1 - i));
        const int c1 = vq_code(seq[ia], !f1), c2 = vq_code(seq[ib], !f2);
        if (c1 == 5 || c2 == 5) { o.status = 2; break; }                      // the reference asserts A/C/G/T/N
        if (c1 == 4 || c2 == 4) continue;                                     // N: the position is skipped
        const int q = (int)qual[ia] * NQ + (int)qual[ib];
        if (c1 == c2) {
            if (bad_match[q]) { o.status = 1; break; }
            o.total += log_match[q];
        } else {
            ++o.mismatches;
            if (bad_mis[q]) { o.status = 1; break; }
            o.total += log_mis[q];
        }
        ++o.len;
    }
    out[k] = o;
}
}  // namespace

void vq_overlap_scores(const char *fastq, const hlmi_vq_overlap *ov, uint64_t n, double mismatch, uint32_t min_read_len,
                       double *score, double *mismatch_rate, int64_t *pos3) {
    // singles.fastq: 4-line records, id = strtoul of the first word behind '@', bases upper-cased (FastqStorage.cpp:92-150)
    const std::string data = read_file(fastq);
    std::vector<VqReadRef> reads;
    std::unordered_map<uint64_t, uint32_t> index_of;
    std::string seq, qual;
    {
        size_t pos = 0, line = 0;
        uint64_t id = 0;
        std::string cur_seq;
        while (pos < data.size()) {
            size_t e = data.find('\n', pos);
            if (e == std::string::npos) e = data.size();
            const std::string_view l(data.data() + pos, e - pos);
            pos = e + 1;
            switch (line++ % 4) {
                case 0: {
                    if (l.empty() || l[0] != '@') fail(HLMI_EINVAL, "%s: read id does not start with @ (line %zu)", fastq, line);
                    size_t b = 1;
                    while (b < l.size() && isspace((unsigned char)l[b])) ++b;
                    size_t w = b;
                    while (w < l.size() && !isspace((unsigned char)l[w])) ++w;
                    id = strtoul(std::string(l.substr(b, w - b)).c_str(), nullptr, 0);
                    break;
                }
                case 1:
                    cur_seq.assign(l);
                    for (char &c : cur_seq) c = (char)toupper((unsigned char)c);
                    break;
                case 2: break;
                case 3: {
                    if (cur_seq.empty()) fail(HLMI_EINVAL, "%s: single read %llu has an empty sequence", fastq, (unsigned long long)id);
                    if (l.size() != cur_seq.size()) fail(HLMI_EINVAL, "%s: read %llu: %zu bases, %zu qualities", fastq, (unsigned long long)id, cur_seq.size(), l.size());
                    index_of[id] = (uint32_t)reads.size();
                    reads.push_back(VqReadRef{(uint64_t)seq.size(), (uint32_t)cur_seq.size(), 0});
                    seq += cur_seq;
                    for (char c : l) {
                        const int q = (int)(unsigned char)c - 33;
                        if (q < 0 || q >= NQ) fail(HLMI_EINVAL, "%s: read %llu: quality character outside '!'..'~'", fastq, (unsigned long long)id);
                        qual.push_back((char)q);
                    }
                    break;
                }
            }
        }
    }
    if (!n) return;
    // p per quality pair, in the reference's order of operations (EdgeCalculator.cpp:41-49, :62-66)
    std::vector<double> P(NQ), lm((size_t)NQ * NQ), lx((size_t)NQ * NQ);
    std::vector<uint8_t> bm((size_t)NQ * NQ), bx((size_t)NQ * NQ);
    for (int q = 0; q < NQ; ++q) P[q] = pow(10, -q / 10.0);
    for (int q1 = 0; q1 < NQ; ++q1)
        for (int q2 = 0; q2 < NQ; ++q2) {
            const double p1 = P[q1], p2 = P[q2];
            const double pm = (1 - p1) * (1 - p2) + (p1 * p2) / 3.0;
            const double px = p1 * (1 - p2) / 3.0 + p2 * (1 - p1) / 3.0 + (2 / 9.0) * p1 * p2;
            lm[(size_t)q1 * NQ + q2] = log(pm); bm[(size_t)q1 * NQ + q2] = pm < mismatch;
            lx[(size_t)q1 * NQ + q2] = log(px); bx[(size_t)q1 * NQ + q2] = px < mismatch;
        }
    std::vector<uint32_t> i1(n), i2(n), p1(n);
    std::vector<uint8_t> ori(n);
    for (uint64_t k = 0; k < n; ++k) {
        if (ov[k].type1 != 's' || ov[k].type2 != 's') fail(HLMI_EINVAL, "overlap %llu is not single-single", (unsigned long long)k);
        auto a = index_of.find(ov[k].id1), b = index_of.find(ov[k].id2);
        if (a == index_of.end() || b == index_of.end())
            fail(HLMI_EINVAL, "overlap %llu names a read that is not in %s", (unsigned long long)k, fastq);
        i1[k] = a->second; i2[k] = b->second; p1[k] = ov[k].pos1;
        ori[k] = (uint8_t)((ov[k].ori1 == '+' ? 1 : 0) | (ov[k].ori2 == '+' ? 2 : 0));
    }
    DBuf<uint8_t> d_seq, d_qual, d_ori, d_bm, d_bx;
    DBuf<VqReadRef> d_reads;
    DBuf<uint32_t> d_i1, d_i2, d_p1;
    DBuf<double> d_lm, d_lx;
    d_seq.upload((const uint8_t *)seq.data(), seq.size()); d_qual.upload((const uint8_t *)qual.data(), qual.size());
    d_reads.upload(reads); d_i1.upload(i1); d_i2.upload(i2); d_p1.upload(p1); d_ori.upload(ori);
    d_lm.upload(lm); d_lx.upload(lx); d_bm.upload(bm); d_bx.upload(bx);
    DBuf<VqScoreOut> d_out(n);
    hipLaunchKernelGGL(vq_score_kernel, grid1(n), dim3(WG), 0, stream(), d_seq.p, d_qual.p, d_reads.p, d_i1.p, d_i2.p, d_p1.p, d_ori.p,
                       (size_t)n, d_lm.p, d_lx.p, d_bm.p, d_bx.p, min_read_len, d_out.p);
    HIP_CHECK(hipGetLastError());
    const std::vector<VqScoreOut> h = d_out.download(n);
    for (uint64_t k = 0; k < n; ++k) {
        if (h[k].status == 2) fail(HLMI_EINVAL, "overlap %llu: a base that is not A, C, G, T or N", (unsigned long long)k);
        pos3[k] = (int64_t)reads[i1[k]].len - (int64_t)p1[k] - (int64_t)reads[i2[k]].len;
        if (h[k].status == 1 || h[k].len == 0) { score[k] = 0; mismatch_rate[k] = 1.0; continue; }
        const double total_len = (double)h[k].len;
        mismatch_rate[k] = (double)(float)h[k].mismatches / total_len;         // float(mismatch_count)/total_len
        score[k] = exp((1.0 / total_len) * h[k].total);
    }
}

}  // namespace hlmi
