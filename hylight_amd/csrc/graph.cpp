// graph.cpp - overlap-graph build = miniasm 0.3-r179 as HyLight runs it
// (`miniasm -d D -n N -e E -c C -f reads.fa in.paf > out.gfa`, script/HyLight.py:137,140,171).
//
// Restated stage by stage from the reference (tools/miniasm):
//   a9   PAF rows -> hits          paf.c:34-67, sdict.c:27-45, hit.c:70-107
//   a10  per-read coverage window  hit.c:109-160
//   a11  hit clipping / filtering  hit.c:162-223
//   a12  containment removal       hit.c:24-36,225-256, sdict.c:69-86
//   a13  hit -> arc, graph build   miniasm.h:86-104, asm.c:9-39, asg.c:27-80
//   a14  transitive reduction      asg.c:148-193  -> HIP kernel (graph_kernels.hip)
//        multi / asymmetric arcs   asg.c:104-145
//   a15  tips, bubbles, short overlaps, internal, bi-loops   asg.c:83-101,204-433  (host: results depend on
//        ascending-vertex in-place deletion order)
//   a16  unitigs, sequences, GFA   asm.c:77-286
// Output must be byte-identical to the reference binary, including the order in which its unstable
// in-place MSD radix sort (ksort.h:132-184) leaves equal keys, so that sort is restated exactly.
#include "graph.h"

#include <chrono>

#include <algorithm>
#include <array>
#include <cctype>
#include <cerrno>
#include <cmath>
#include <deque>

#include "paf_io.h"

namespace hlmi {

namespace {

struct Opt {
    int min_span = 2000, min_match = 100, min_dp = 3;
    float min_iden = .05f;
    int max_hang = 1000, min_ovlp = 2000;
    float int_frac = .8f;
    int gap_fuzz = 1000, n_rounds = 2, bub_dist = 50000, max_ext = 4;
    float min_drop = .5f, max_drop = .7f, final_drop = .8f;     // common.c:5-23
};

struct Hit {                       // miniasm.h:29-34 (32 bytes)
    uint64_t qns;
    uint32_t qe, tn, ts, te;
    uint32_t ml : 31, rev : 1;
    uint32_t bl : 31, del : 1;
};
struct Sub {                       // miniasm.h:38-40
    uint32_t s : 31, del : 1, e;
};
struct SdSeq {
    std::string name;
    uint32_t len;
    bool del = false, aux = false;
};
struct Dict {
    std::vector<SdSeq> seq;
    std::unordered_map<std::string, uint32_t> h;
    std::string key_;                                  // scratch for look-ups by view
    uint32_t put(std::string_view name, uint32_t len) {
        key_.assign(name);
        auto it = h.find(key_);
        if (it != h.end()) return it->second;
        uint32_t id = (uint32_t)seq.size();
        seq.push_back(SdSeq{key_, len});
        h.emplace(key_, id);
        return id;
    }
    int32_t get(const std::string &name) const {
        auto it = h.find(name);
        return it == h.end() ? -1 : (int32_t)it->second;
    }
};

// ---- the reference's in-place radix sort: American-flag permutation on the top byte, recursion on the
// ---- lower bytes, insertion sort for <= 64 elements.  Equal keys end up in the same order as ksort.h leaves them.
constexpr int RS_MIN = 64;
template <class T, class Key>
void rs_insertion(T *beg, T *end, Key key) {
    for (T *i = beg + 1; i < end; ++i)
        if (key(*i) < key(*(i - 1))) {
            T tmp = *i, *j;
            for (j = i; j > beg && key(tmp) < key(*(j - 1)); --j) *j = *(j - 1);
            *j = tmp;
        }
}
template <class T, class Key>
void rs_pass(T *beg, T *end, int shift, Key key) {
    struct Bk { T *b, *e; };
    Bk bk[256];
    for (auto &k : bk) k.b = k.e = beg;
    for (T *i = beg; i != end; ++i) ++bk[key(*i) >> shift & 255].e;
    for (int k = 1; k < 256; ++k) { bk[k].e += bk[k - 1].e - beg; bk[k].b = bk[k - 1].e; }
    for (int k = 0; k < 256;) {
        if (bk[k].b != bk[k].e) {
            int l = (int)(key(*bk[k].b) >> shift & 255);
            if (l != k) {
                T tmp = *bk[k].b, swap;
                do {
                    swap = tmp; tmp = *bk[l].b; *bk[l].b++ = swap;
                    l = (int)(key(tmp) >> shift & 255);
                } while (l != k);
                *bk[k].b++ = tmp;
            } else ++bk[k].b;
        } else ++k;
    }
    bk[0].b = beg;
    for (int k = 1; k < 256; ++k) bk[k].b = bk[k - 1].e;
    if (shift) {
        int s2 = shift > 8 ? shift - 8 : 0;
        for (auto &k : bk) {
            if (k.e - k.b > RS_MIN) rs_pass(k.b, k.e, s2, key);
            else if (k.e - k.b > 1) rs_insertion(k.b, k.e, key);
        }
    }
}
template <class T, class Key>
void radix_sort64(T *beg, T *end, Key key) {
    if (end - beg <= RS_MIN) rs_insertion(beg, end, key);
    else rs_pass(beg, end, 56, key);
}

// ---- a9 ------------------------------------------------------------------------------------------
struct PafRow { uint32_t ql, qs, qe, tl, ts, te, ml, bl; bool rev; std::string_view qn, tn; };
// strtol(s, nullptr, 10) on a token that is not NUL-terminated: leading white space, optional sign, digits up to the
// first other character, saturation at LONG_MIN / LONG_MAX
long strtol_view(std::string_view t) {
    size_t i = 0;
    while (i < t.size() && (t[i] == ' ' || (t[i] >= '\t' && t[i] <= '\r'))) ++i;
    bool neg = false;
    if (i < t.size() && (t[i] == '+' || t[i] == '-')) { neg = t[i] == '-'; ++i; }
    unsigned long long v = 0;
    const unsigned long long lim = neg ? (unsigned long long)LONG_MAX + 1ull : (unsigned long long)LONG_MAX;
    bool sat = false;
    for (; i < t.size() && t[i] >= '0' && t[i] <= '9'; ++i) {
        if (!sat) {
            v = v * 10 + (unsigned long long)(t[i] - '0');
            if (v > lim) { v = lim; sat = true; }
        }
    }
    return neg ? (long)(0ull - v) : (long)v;
}

std::vector<Hit> read_hits(const char *fn, const Opt &o, Dict &d) {
    const std::string file = read_file(fn);
    const std::string_view data(file);
    std::vector<Hit> hits;
    PafRow r{};
    size_t pos = 0, N = data.size();
    while (pos < N) {
        size_t e = data.find('\n', pos);
        if (e == std::string_view::npos) e = N;
        size_t le = e;
        if (le - pos > 1 && data[le - 1] == '\r') --le;
        // fields (paf.c:34-61): strtol on each numeric column, row skipped when it has < 10 of them
        int t = 0;
        size_t p = pos;
        while (true) {
            size_t f = data.find('\t', p);
            if (f == std::string_view::npos || f > le) f = le;
            const std::string_view tok = data.substr(p, f - p);
            const long v = t >= 1 && t <= 10 && t != 4 && t != 5 ? strtol_view(tok) : 0;
            switch (t) {
                case 0: r.qn = tok; break;
                case 1: r.ql = (uint32_t)v; break;
                case 2: r.qs = (uint32_t)v; break;
                case 3: r.qe = (uint32_t)v; break;
                case 4: r.rev = !tok.empty() && tok[0] == '-'; break;
                case 5: r.tn = tok; break;
                case 6: r.tl = (uint32_t)v; break;
                case 7: r.ts = (uint32_t)v; break;
                case 8: r.te = (uint32_t)v; break;
                case 9: r.ml = (uint32_t)v & 0x7fffffffu; break;
                case 10: r.bl = (uint32_t)v; break;
                default: break;
            }
            ++t;
            if (f >= le) break;
            p = f + 1;
        }
        pos = e + 1;
        if (t < 10) continue;
        // hit.c:85 - unsigned differences, int thresholds
        if (r.qe - r.qs < (uint32_t)o.min_span || r.te - r.ts < (uint32_t)o.min_span || (int)r.ml < o.min_match) continue;
        Hit h{};
        h.qns = (uint64_t)d.put(r.qn, r.ql) << 32 | r.qs;
        h.qe = r.qe;
        h.tn = d.put(r.tn, r.tl);
        h.ts = r.ts; h.te = r.te; h.rev = r.rev; h.ml = r.ml; h.bl = r.bl & 0x7fffffffu; h.del = 0;
        hits.push_back(h);
        if ((uint32_t)(h.qns >> 32) != h.tn) {      // bi_dir = 1 (main.c:35)
            Hit m{};
            m.qns = (uint64_t)h.tn << 32 | r.ts;
            m.qe = r.te;
            m.tn = (uint32_t)(h.qns >> 32);
            m.ts = r.qs; m.te = r.qe; m.rev = r.rev; m.ml = r.ml; m.bl = r.bl & 0x7fffffffu; m.del = 0;
            hits.push_back(m);
        }
    }
    radix_sort64(hits.data(), hits.data() + hits.size(), [](const Hit &h) { return h.qns; });
    return hits;
}

// ---- a10 -----------------------------------------------------------------------------------------
std::vector<Sub> hit_sub(int min_dp, float min_iden, const std::vector<Hit> &a, size_t n, size_t n_sub) {
    std::vector<Sub> sub(n_sub, Sub{0, 0, 0});
    std::vector<uint32_t> b;
    for (size_t i = 1, last = 0; i <= n; ++i) {
        if (i != n && a[i].qns >> 32 == a[i - 1].qns >> 32) continue;
        const uint32_t qid = (uint32_t)(a[i - 1].qns >> 32);
        b.clear();
        for (size_t j = last; j < i; ++j) {
            if (a[j].tn == qid || (float)(int)a[j].ml < (float)(int)a[j].bl * min_iden) continue;
            const uint32_t qs = (uint32_t)a[j].qns, qe = a[j].qe;
            if (qe > qs) { b.push_back(qs << 1); b.push_back(qe << 1 | 1); }
        }
        std::sort(b.begin(), b.end());
        uint32_t max_s = 0, max_e = 0, max2_s = 0, max2_e = 0;
        size_t start = 0;
        int dp = 0;
        for (size_t j = 0; j < b.size(); ++j) {
            const int old = dp;
            if (b[j] & 1) --dp; else ++dp;
            if (old < min_dp && dp >= min_dp) start = b[j] >> 1;
            else if (old >= min_dp && dp < min_dp) {
                const int len = (int)((b[j] >> 1) - start);
                if ((uint32_t)len > max_e - max_s) { max2_s = max_s; max2_e = max_e; max_s = (uint32_t)start & 0x7fffffffu; max_e = b[j] >> 1; }
                else if ((uint32_t)len > max2_e - max2_s) { max2_s = (uint32_t)start & 0x7fffffffu; max2_e = b[j] >> 1; }
            }
        }
        if (max_e - max_s > 0) { sub[qid].s = max_s; sub[qid].e = max_e; sub[qid].del = 0; }
        else sub[qid].del = 1;
        last = i;
    }
    return sub;
}

// ---- a11 -----------------------------------------------------------------------------------------
size_t hit_cut(const std::vector<Sub> &reg, int min_span, size_t n, std::vector<Hit> &a) {
    size_t m = 0;
    for (size_t i = 0; i < n; ++i) {
        Hit *p = &a[i];
        const Sub *rq = &reg[p->qns >> 32], *rt = &reg[p->tn];
        if (rq->del || rt->del) continue;
        const uint32_t q0 = (uint32_t)p->qns;
        int qs, qe, ts, te;
        if (p->rev) {
            qs = p->te < rt->e ? q0 : q0 + (p->te - rt->e);
            qe = p->ts > rt->s ? p->qe : p->qe - (rt->s - p->ts);
            ts = p->qe < rq->e ? p->ts : p->ts + (p->qe - rq->e);
            te = q0 > rq->s ? p->te : p->te - (rq->s - q0);
        } else {
            qs = p->ts > rt->s ? q0 : q0 + (rt->s - p->ts);
            qe = p->te < rt->e ? p->qe : p->qe - (p->te - rt->e);
            ts = q0 > rq->s ? p->ts : p->ts + (rq->s - q0);
            te = p->qe < rq->e ? p->te : p->te - (p->qe - rq->e);
        }
        // hit.c:181-184: the 31-bit field `s` promotes to int (signed compare), the 32-bit `e` forces an
        // unsigned compare
        const int sq = (int)rq->s, st2 = (int)rt->s;
        qs = (qs > sq ? qs : sq) - sq;
        qe = (int)(((uint32_t)qe < rq->e ? (uint32_t)qe : rq->e) - (uint32_t)sq);
        ts = (ts > st2 ? ts : st2) - st2;
        te = (int)(((uint32_t)te < rt->e ? (uint32_t)te : rt->e) - (uint32_t)st2);
        if (qe - qs >= min_span && te - ts >= min_span) {
            p->qns = p->qns >> 32 << 32 | (uint32_t)qs;
            p->qe = (uint32_t)qe; p->ts = (uint32_t)ts; p->te = (uint32_t)te;
            a[m++] = *p;
        }
    }
    return m;
}

constexpr int HT_INT = -1, HT_QCONT = -2, HT_TCONT = -3, HT_SHORT = -4;

// miniasm.h:86-104 with its mixed signed / unsigned arithmetic kept
int hit2arc(const Hit *h, int ql, int tl, int max_hang, float int_frac, int min_ovlp, Arc *p) {
    int32_t tl5, tl3, ext5, ext3, qs = (int32_t)(uint32_t)h->qns;
    uint32_t u, v, l;
    if (h->rev) { tl5 = (int32_t)((uint32_t)tl - h->te); tl3 = (int32_t)h->ts; }
    else { tl5 = (int32_t)h->ts; tl3 = (int32_t)((uint32_t)tl - h->te); }
    ext5 = qs < tl5 ? qs : tl5;
    const uint32_t qr = (uint32_t)ql - h->qe;                  // ql - h->qe is unsigned in the reference
    ext3 = qr < (uint32_t)tl3 ? (int32_t)qr : tl3;
    const uint32_t span = h->qe - (uint32_t)qs;
    if (ext5 > max_hang || ext3 > max_hang || (float)span < (float)(span + (uint32_t)ext5 + (uint32_t)ext3) * int_frac) return HT_INT;
    if (qs <= tl5 && qr <= (uint32_t)tl3) return HT_QCONT;
    else if (qs >= tl5 && qr >= (uint32_t)tl3) return HT_TCONT;
    else if (qs > tl5) { u = 0; v = !!h->rev; l = (uint32_t)(qs - tl5); }
    else { u = 1; v = !h->rev; l = qr - (uint32_t)tl3; }
    if (span + (uint32_t)ext5 + (uint32_t)ext3 < (uint32_t)min_ovlp || h->te - h->ts + (uint32_t)ext5 + (uint32_t)ext3 < (uint32_t)min_ovlp) return HT_SHORT;
    u |= (uint32_t)(h->qns >> 32) << 1; v |= h->tn << 1;
    p->ul = (uint64_t)u << 32 | l; p->v = v; p->ol = ((uint32_t)ql - l) & 0x7fffffffu; p->del = 0;
    return (int)l;
}

size_t hit_flt(const std::vector<Sub> &sub, int max_hang, int min_ovlp, size_t n, std::vector<Hit> &a) {
    size_t m = 0;
    Arc t;
    for (size_t i = 0; i < n; ++i) {
        const Hit *h = &a[i];
        const Sub *sq = &sub[h->qns >> 32], *st = &sub[h->tn];
        if (sq->del || st->del) continue;
        const int r = hit2arc(h, (int)(sq->e - sq->s), (int)(st->e - st->s), max_hang, .5f, min_ovlp, &t);
        if (r >= 0 || r == HT_QCONT || r == HT_TCONT) a[m++] = *h;
    }
    return m;
}

// ---- a12 -----------------------------------------------------------------------------------------
size_t hit_contained(const Opt &o, Dict &d, std::vector<Sub> &sub, size_t n, std::vector<Hit> &a) {
    Arc t;
    const size_t old_n = d.seq.size();
    for (size_t i = 0; i < n; ++i) {
        const Hit *h = &a[i];
        Sub *sq = &sub[h->qns >> 32], *st = &sub[h->tn];
        const int r = hit2arc(h, (int)(sq->e - sq->s), (int)(st->e - st->s), o.max_hang, o.int_frac, o.min_ovlp, &t);
        if (r == HT_QCONT) sq->del = 1;
        else if (r == HT_TCONT) st->del = 1;
    }
    for (size_t i = 0; i < old_n; ++i) if (sub[i].del) d.seq[i].del = true;
    for (auto &s : d.seq) s.aux = false;                       // ma_hit_mark_unused
    for (size_t i = 0; i < n; ++i) d.seq[a[i].qns >> 32].aux = d.seq[a[i].tn].aux = true;
    for (auto &s : d.seq) { if (!s.aux) s.del = true; else s.aux = false; }
    std::vector<int32_t> map(old_n, -1);                       // sd_squeeze
    size_t j = 0;
    d.h.clear();
    for (size_t i = 0; i < old_n; ++i) {
        if (d.seq[i].del) continue;
        if (j != i) d.seq[j] = std::move(d.seq[i]);
        map[i] = (int32_t)j++;
    }
    d.seq.resize(j);
    for (size_t i = 0; i < j; ++i) d.h.emplace(d.seq[i].name, (uint32_t)i);
    for (size_t i = 0; i < old_n; ++i) if (map[i] >= 0) sub[map[i]] = sub[i];
    size_t m = 0;
    for (size_t i = 0; i < n; ++i) {
        const int32_t qn = map[a[i].qns >> 32], tn = map[a[i].tn];
        if (qn >= 0 && tn >= 0) {
            a[i].qns = (uint64_t)qn << 32 | (uint32_t)a[i].qns;
            a[i].tn = (uint32_t)tn;
            a[m++] = a[i];
        }
    }
    return m;
}

// ---- string graph (asg.c) --------------------------------------------------------------------------
struct Graph {
    std::vector<Arc> arc;
    std::vector<GSeq> seq;
    std::vector<uint64_t> idx;
    bool is_srt = false, is_symm = false, has_idx = false;
    uint32_t n_arc(uint32_t v) const { return (uint32_t)idx[v]; }
    Arc *arcs(uint32_t v) { return &arc[idx[v] >> 32]; }
    const Arc *arcs(uint32_t v) const { return &arc[idx[v] >> 32]; }
};

void g_cleanup(Graph &g) {                                     // asg.c:57-80
    size_t n = 0;
    for (size_t e = 0; e < g.arc.size(); ++e) {
        const uint32_t u = (uint32_t)(g.arc[e].ul >> 32), v = g.arc[e].v;
        if (!g.arc[e].del && !g.seq[u >> 1].del && !g.seq[v >> 1].del) g.arc[n++] = g.arc[e];
    }
    if (n < g.arc.size()) g.has_idx = false;
    g.arc.resize(n);
    if (!g.is_srt) {
        radix_sort64(g.arc.data(), g.arc.data() + g.arc.size(), [](const Arc &a) { return a.ul; });
        g.is_srt = true;
    }
    if (!g.has_idx) {
        g.idx.assign(g.seq.size() * 2, 0);
        for (size_t i = 1, last = 0; i <= n; ++i)
            if (i == n || g.arc[i - 1].ul >> 32 != g.arc[i].ul >> 32) {
                g.idx[g.arc[i - 1].ul >> 32] = (uint64_t)last << 32 | (i - last);
                last = i;
            }
        g.has_idx = true;
    }
}

void g_arc_del(Graph &g, uint32_t v, uint32_t w, bool del) {  // asg.h:53-59
    Arc *av = g.arcs(v);
    for (uint32_t i = 0, nv = g.n_arc(v); i < nv; ++i) if (av[i].v == w) av[i].del = del;
}
void g_seq_del(Graph &g, uint32_t s) {                          // asg.h:62-76
    g.seq[s].del = 1;
    for (uint32_t k = 0; k < 2; ++k) {
        const uint32_t v = s << 1 | k;
        Arc *av = g.arcs(v);
        for (uint32_t i = 0, nv = g.n_arc(v); i < nv; ++i) { av[i].del = 1; g_arc_del(g, av[i].v ^ 1, v ^ 1, true); }
    }
}

void g_del_multi(Graph &g) {                                   // asg.c:104-121
    const uint32_t n_vtx = (uint32_t)g.seq.size() * 2;
    std::vector<uint32_t> cnt(n_vtx, 0);
    uint32_t n_multi = 0;
    for (uint32_t v = 0; v < n_vtx; ++v) {
        Arc *av = g.arcs(v);
        const int32_t nv = (int32_t)g.n_arc(v);
        if (nv < 2) continue;
        for (int32_t i = nv - 1; i >= 0; --i) ++cnt[av[i].v];
        for (int32_t i = nv - 1; i >= 0; --i) if (--cnt[av[i].v] != 0) { av[i].del = 1; ++n_multi; }
    }
    if (n_multi) g_cleanup(g);
}
void g_del_asymm(Graph &g) {                                   // asg.c:124-138
    uint32_t n_asymm = 0;
    for (size_t e = 0; e < g.arc.size(); ++e) {
        const uint32_t v = g.arc[e].v ^ 1, u = (uint32_t)(g.arc[e].ul >> 32) ^ 1;
        const Arc *av = g.arcs(v);
        uint32_t i, nv = g.n_arc(v);
        for (i = 0; i < nv; ++i) if (av[i].v == u) break;
        if (i == nv) { g.arc[e].del = 1; ++n_asymm; }
    }
    if (n_asymm) g_cleanup(g);
}
void g_symm(Graph &g) { g_del_multi(g); g_del_asymm(g); g.is_symm = true; }

int g_del_short(Graph &g, float drop_ratio) {                  // asg.c:83-101
    const uint32_t n_vtx = (uint32_t)g.seq.size() * 2;
    uint32_t n_short = 0;
    for (uint32_t v = 0; v < n_vtx; ++v) {
        Arc *av = g.arcs(v);
        const uint32_t nv = g.n_arc(v);
        if (nv < 2) continue;
        // (uint32_t)(ol * ratio + .499): with `-n 1` the caller's ratio is NaN (0.2f / 0 * 0, main.c:168)
        // and x86-64 converts NaN to 0x80000000'00000000 -> low word 0
        const double x = (double)((float)(int)av[0].ol * drop_ratio) + .499;
        const uint32_t thres = std::isnan(x) ? 0u : (uint32_t)(int64_t)x;
        uint32_t i;
        for (i = nv - 1; i >= 1 && av[i].ol < thres; --i) {}
        for (i = i + 1; i < nv; ++i) { av[i].del = 1; ++n_short; }
    }
    if (n_short) { g_cleanup(g); g_symm(g); }
    return (int)n_short;
}

constexpr int ET_MERGEABLE = 0, ET_TIP = 1, ET_MULTI_OUT = 2, ET_MULTI_NEI = 3;
int g_is_utg_end(const Graph &g, uint32_t v, uint64_t *lw) {     // asg.c:204-221
    const Arc *av = g.arcs(v ^ 1);
    uint32_t nv0 = g.n_arc(v ^ 1), nv = 0;
    int i0 = -1;
    for (uint32_t i = 0; i < nv0; ++i) if (!av[i].del) { i0 = (int)i; ++nv; }
    if (nv == 0) return ET_TIP;
    if (nv > 1) return ET_MULTI_OUT;
    if (lw) *lw = av[i0].ul << 32 | av[i0].v;
    const uint32_t w = av[i0].v ^ 1;
    const Arc *aw = g.arcs(w);
    uint32_t nw = 0;
    for (uint32_t i = 0, nw0 = g.n_arc(w); i < nw0; ++i) if (!aw[i].del) ++nw;
    return nw != 1 ? ET_MULTI_NEI : ET_MERGEABLE;
}
int g_extend(const Graph &g, uint32_t v, int max_ext, std::vector<uint64_t> &a) {   // asg.c:223-236
    int ret;
    uint64_t lw = 0;
    a.clear();
    a.push_back(v);
    do {
        ret = g_is_utg_end(g, v ^ 1, &lw);
        if (ret != 0) break;
        a.push_back(lw);
        v = (uint32_t)lw;
    } while (--max_ext > 0);
    return ret;
}
int g_cut_tip(Graph &g, int max_ext) {                          // asg.c:238-254
    std::vector<uint64_t> a;
    uint32_t cnt = 0;
    for (uint32_t v = 0, n_vtx = (uint32_t)g.seq.size() * 2; v < n_vtx; ++v) {
        if (g.seq[v >> 1].del) continue;
        if (g_is_utg_end(g, v, nullptr) != ET_TIP) continue;
        if (g_extend(g, v, max_ext, a) == ET_MERGEABLE) continue;
        for (uint64_t x : a) g_seq_del(g, (uint32_t)x >> 1);
        ++cnt;
    }
    if (cnt) g_cleanup(g);
    return (int)cnt;
}
int g_cut_internal(Graph &g, int max_ext) {                     // asg.c:256-272
    std::vector<uint64_t> a;
    uint32_t cnt = 0;
    for (uint32_t v = 0, n_vtx = (uint32_t)g.seq.size() * 2; v < n_vtx; ++v) {
        if (g.seq[v >> 1].del) continue;
        if (g_is_utg_end(g, v, nullptr) != ET_MULTI_NEI) continue;
        if (g_extend(g, v, max_ext, a) != ET_MULTI_NEI) continue;
        for (uint64_t x : a) g_seq_del(g, (uint32_t)x >> 1);
        ++cnt;
    }
    if (cnt) g_cleanup(g);
    return (int)cnt;
}
int g_cut_biloop(Graph &g, int max_ext) {                       // asg.c:274-306
    std::vector<uint64_t> a;
    uint32_t cnt = 0;
    for (uint32_t v = 0, n_vtx = (uint32_t)g.seq.size() * 2; v < n_vtx; ++v) {
        if (g.seq[v >> 1].del) continue;
        if (g_is_utg_end(g, v, nullptr) != ET_MULTI_NEI) continue;
        if (g_extend(g, v, max_ext, a) != ET_MULTI_OUT) continue;
        const uint32_t x = (uint32_t)a.back() ^ 1;
        uint32_t w = UINT32_MAX, ov = 0, ox = 0;
        const Arc *av = g.arcs(v ^ 1);
        for (uint32_t i = 0, nv = g.n_arc(v ^ 1); i < nv; ++i) if (!av[i].del) w = av[i].v ^ 1;
        if (w == UINT32_MAX) fail(HLMI_EINVAL, "miniasm: bi-loop without neighbour (corrupt graph)");
        const Arc *aw = g.arcs(w);
        for (uint32_t i = 0, nw = g.n_arc(w); i < nw; ++i) {
            if (aw[i].del) continue;
            if (aw[i].v == x) ox = aw[i].ol;
            if (aw[i].v == v) ov = aw[i].ol;
        }
        if (ov == 0 && ox == 0) continue;
        if (ov > ox) { g_arc_del(g, w, x, true); g_arc_del(g, x ^ 1, w ^ 1, true); ++cnt; }
    }
    if (cnt) g_cleanup(g);
    return (int)cnt;
}

// bubble popping, asg.c:312-433
struct BInfo { uint32_t p, d, c; uint32_t r : 31, s : 1; };
struct BBuf { std::vector<BInfo> a; std::vector<uint32_t> S, T, b, e; };
uint32_t g_count_out(const Graph &g, uint32_t v) {
    uint32_t n = 0;
    const Arc *av = g.arcs(v);
    for (uint32_t i = 0, nv = g.n_arc(v); i < nv; ++i) if (!av[i].del) ++n;
    return n;
}
void g_bub_backtrack(Graph &g, uint32_t v0, BBuf &b) {
    for (uint32_t x : b.b) g.seq[x >> 1].del = 1;
    for (uint32_t x : b.e) {
        Arc *a = &g.arc[x];
        a->del = 1;
        g_arc_del(g, a->v ^ 1, (uint32_t)(a->ul >> 32) ^ 1, true);
    }
    uint32_t v = b.S[0];
    do {
        const uint32_t u = b.a[v].p;
        g.seq[v >> 1].del = 0;
        g_arc_del(g, u, v, false);
        g_arc_del(g, v ^ 1, u ^ 1, false);
        v = u;
    } while (v != v0);
}
uint64_t g_bub_pop1(Graph &g, uint32_t v0, int max_dist, BBuf &b) {
    uint32_t n_pending = 0;
    uint64_t n_pop = 0;
    if (g.seq[v0 >> 1].del) return 0;
    if ((uint32_t)g.idx[v0] < 2) return 0;
    b.S.clear(); b.T.clear(); b.b.clear(); b.e.clear();
    b.a[v0].c = b.a[v0].d = 0;
    b.S.push_back(v0);
    bool reset = false;
    do {
        const uint32_t v = b.S.back();
        b.S.pop_back();
        const uint32_t d = b.a[v].d, c = b.a[v].c, nv = g.n_arc(v);
        const Arc *av = g.arcs(v);
        uint32_t i;
        for (i = 0; i < nv; ++i) {
            const uint32_t w = av[i].v, l = (uint32_t)av[i].ul;
            BInfo *t = &b.a[w];
            if (w == v0) { reset = true; break; }
            if (av[i].del) continue;
            b.e.push_back((uint32_t)(g.idx[v] >> 32) + i);
            if (d + l > (uint32_t)max_dist) break;
            if (t->s == 0) {
                b.b.push_back(w);
                t->p = v; t->s = 1; t->d = d + l;
                t->r = g_count_out(g, w ^ 1);
                ++n_pending;
            } else {
                if (c + 1 > t->c || (c + 1 == t->c && d + l > t->d)) t->p = v;
                if (c + 1 > t->c) t->c = c + 1;
                if (d + l < t->d) t->d = d + l;
            }
            if (--(t->r) == 0) {
                if (g.n_arc(w)) b.S.push_back(w); else b.T.push_back(w);
                --n_pending;
            }
        }
        if (reset || i < nv || b.S.empty()) { reset = true; break; }
    } while (b.S.size() > 1 || n_pending);
    if (!reset) {
        g_bub_backtrack(g, v0, b);
        n_pop = 1 | (uint64_t)b.T.size() << 32;
    }
    for (uint32_t x : b.b) { BInfo *t = &b.a[x]; t->s = 0; t->c = 0; t->d = 0; }
    return n_pop;
}
int g_pop_bubble(Graph &g, int max_dist) {
    const uint32_t n_vtx = (uint32_t)g.seq.size() * 2;
    uint64_t n_pop = 0;
    if (!g.is_symm) g_symm(g);
    BBuf b;
    b.a.assign(n_vtx, BInfo{0, 0, 0, 0, 0});
    for (uint32_t v = 0; v < n_vtx; ++v) {
        const uint32_t nv = g.n_arc(v);
        const Arc *av = g.arcs(v);
        if (nv < 2 || g.seq[v >> 1].del) continue;
        uint32_t n_arc = 0;
        for (uint32_t i = 0; i < nv; ++i) if (!av[i].del) ++n_arc;
        if (n_arc > 1) n_pop += g_bub_pop1(g, v, max_dist, b);
    }
    if (n_pop) g_cleanup(g);
    return (int)n_pop;
}

// ---- a13 -----------------------------------------------------------------------------------------
Graph sg_gen(const Opt &o, const Dict &d, const std::vector<Sub> &sub, size_t n_hits, const std::vector<Hit> &hit) {
    Graph g;
    g.seq.resize(d.seq.size());
    for (size_t i = 0; i < d.seq.size(); ++i) {
        g.seq[i].len = (sub[i].e - sub[i].s) & 0x7fffffffu;
        g.seq[i].del = (sub[i].del || d.seq[i].del) ? 1 : 0;
    }
    for (size_t i = 0; i < n_hits; ++i) {
        Arc t;
        const Hit *h = &hit[i];
        const uint32_t qn = (uint32_t)(h->qns >> 32);
        const int r = hit2arc(h, (int)(sub[qn].e - sub[qn].s), (int)(sub[h->tn].e - sub[h->tn].s), o.max_hang, o.int_frac,
                              o.min_ovlp, &t);
        if (r >= 0) {
            if (qn == h->tn) {
                if ((uint32_t)h->qns == h->ts && h->qe == h->te && h->rev) g.seq[qn].del = 1;
                continue;
            }
            g.arc.push_back(t);
        } else if (r == HT_QCONT) g.seq[qn].del = 1;
    }
    g_cleanup(g);
    return g;
}

// ---- a16 -----------------------------------------------------------------------------------------
struct Utg {
    uint32_t len = 0;
    bool circ = false;
    uint32_t start = 0, end = 0;
    std::vector<uint64_t> a;
    std::string s;
    bool has_seq = false;
};
struct UGraph { std::vector<Utg> u; Graph g; };

UGraph ug_gen(Graph &g) {                                       // asm.c:117-206
    const uint32_t n_vtx = (uint32_t)g.seq.size() * 2;
    std::vector<int32_t> mark(n_vtx, 0);
    UGraph ug;
    std::deque<uint64_t> q;
    auto cnt = [&](uint32_t v) { return (uint32_t)g.idx[v]; };
    auto first = [&](uint32_t v) -> const Arc & { return g.arc[g.idx[v] >> 32]; };
    for (uint32_t v = 0; v < n_vtx; ++v) {
        if (g.seq[v >> 1].del || cnt(v) == 0 || mark[v]) continue;
        mark[v] = 1;
        q.clear();
        uint32_t start = v, end = v ^ 1, len = 0, w = v, x, l;
        while (true) {
            if (cnt(w) != 1) break;
            x = first(w).v;
            if (cnt(x ^ 1) != 1) break;
            mark[x] = mark[w ^ 1] = 1;
            l = (uint32_t)first(w).ul;
            q.push_back((uint64_t)w << 32 | l);
            end = x ^ 1; len += l;
            w = x;
            if (x == v) break;
        }
        bool circular = false;
        if (start != (end ^ 1) || q.empty()) {
            l = g.seq[end >> 1].len;
            q.push_back((uint64_t)(end ^ 1) << 32 | l);
            len += l;
        } else {
            start = end = UINT32_MAX;
            circular = true;
        }
        if (!circular) {
            x = v;
            while (true) {
                if (cnt(x ^ 1) != 1) break;
                w = first(x ^ 1).v ^ 1;
                if (cnt(w) != 1) break;
                mark[x] = mark[w ^ 1] = 1;
                l = (uint32_t)first(w).ul;
                q.push_front((uint64_t)w << 32 | l);
                start = w; len += l;
                x = w;
            }
        }
        if (start != UINT32_MAX) mark[start] = mark[end] = 1;
        Utg p;
        p.start = start; p.end = end; p.len = len & 0x7fffffffu; p.circ = (start == UINT32_MAX);
        p.a.assign(q.begin(), q.end());
        ug.u.push_back(std::move(p));
    }
    for (uint32_t v = 0; v < n_vtx; ++v) mark[v] = -1;
    for (size_t i = 0; i < ug.u.size(); ++i) {
        if (ug.u[i].circ) continue;
        mark[ug.u[i].start] = (int32_t)(i << 1 | 0);
        mark[ug.u[i].end] = (int32_t)(i << 1 | 1);
    }
    for (size_t i = 0; i < g.arc.size(); ++i) {
        const Arc *p = &g.arc[i];
        if (p->del) continue;
        if (mark[(uint32_t)(p->ul >> 32) ^ 1] >= 0 && mark[p->v] >= 0) {
            const uint32_t u = (uint32_t)mark[(uint32_t)(p->ul >> 32) ^ 1] ^ 1;
            int l = (int)ug.u[u >> 1].len - (int)p->ol;
            if (l < 0) l = 1;
            Arc qa;
            qa.ol = p->ol; qa.del = 0;
            qa.ul = (uint64_t)u << 32 | (uint32_t)l;
            qa.v = (uint32_t)mark[p->v];
            ug.g.arc.push_back(qa);
        }
    }
    ug.g.seq.resize(ug.u.size());
    for (size_t i = 0; i < ug.u.size(); ++i) { ug.g.seq[i].len = ug.u[i].len; ug.g.seq[i].del = 0; }
    g_cleanup(ug.g);
    return ug;
}

char comp_base(int c) {                                         // asm.c:220-229 (IUPAC complement table)
    static const char up[] = "TVGHEFCDIJMLKNOPQYSAABWXRZ";
    if (c >= 'A' && c <= 'Z') return up[c - 'A'];
    if (c >= 'a' && c <= 'z') return (char)(up[c - 'a'] + 32);
    if (c == 96) return 64;
    return (char)c;
}

void ug_seq(UGraph &ug, const Dict &d, const std::vector<Sub> &sub, const char *fn) {   // asm.c:232-286
    struct Intv { uint32_t utg, ori, start, len; };
    std::vector<Intv> tmp(d.seq.size(), Intv{0, 0, 0, 0});
    for (size_t i = 0; i < ug.u.size(); ++i) {
        Utg &u = ug.u[i];
        u.s.assign(u.len, 'N');
        u.has_seq = true;
        uint32_t l = 0;
        for (uint64_t x : u.a) {
            Intv &t = tmp[x >> 33];
            t.utg = (uint32_t)i; t.ori = (uint32_t)(x >> 32) & 1; t.start = l; t.len = (uint32_t)x;
            l += t.len;
        }
    }
    // only the reads that sit on a unitig bring their bases along (a few hundred of the read set)
    SeqSet reads;
    std::string key;
    const std::function<bool(std::string_view)> on_unitig = [&](std::string_view name) {
        key.assign(name);
        const int32_t id = d.get(key);
        return id >= 0 && tmp[id].len != 0;
    };
    read_seqs_subset(fn, &on_unitig, reads);
    for (size_t r = 0; r < reads.size(); ++r) {
        const int32_t id = d.get(reads.names[r]);
        if (id < 0 || tmp[id].len == 0) continue;
        const Intv &t = tmp[id];
        Utg &u = ug.u[t.utg];
        const uint32_t rl = reads.len(r);
        if (sub[id].e - sub[id].s > rl) fail(HLMI_EINVAL, "read %s is shorter in %s than in the PAF", reads.names[r].c_str(), fn);
        const char *s = reads.bases.data() + reads.off[r] + sub[id].s;
        const uint32_t sl = sub[id].e - sub[id].s;
        for (uint32_t i = 0; i < t.len; ++i) {
            if (t.start + i >= u.s.size()) break;
            if (!t.ori) u.s[t.start + i] = s[i];
            else {
                const int c = (uint8_t)s[sl - 1 - i];
                u.s[t.start + i] = c >= 128 ? 'N' : comp_base(c);
            }
        }
    }
}

void appendf(std::string &out, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    int n = vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (n < (int)sizeof buf) { out.append(buf, n); return; }
    std::vector<char> big(n + 1);
    va_start(ap, fmt);
    vsnprintf(big.data(), big.size(), fmt, ap);
    va_end(ap);
    out.append(big.data(), n);
}

void ug_print(const UGraph &ug, const Dict &d, const std::vector<Sub> &sub, std::string &out) {   // asm.c:77-112
    char name[32];
    for (size_t i = 0; i < ug.u.size(); ++i) {
        const Utg &p = ug.u[i];
        snprintf(name, sizeof name, "utg%.6d%c", (int)i + 1, "lc"[p.circ]);
        out += "S\t"; out += name; out += '\t';
        if (p.has_seq) out += p.s; else out += '*';
        appendf(out, "\tLN:i:%d\n", (int)p.len);
        uint32_t l = 0;
        for (uint64_t a : p.a) {
            const uint32_t x = (uint32_t)(a >> 33);
            appendf(out, "a\t%s\t%d\t%s:%d-%d\t%c\t%d\n", name, (int)l, d.seq[x].name.c_str(), (int)sub[x].s + 1, (int)sub[x].e,
                    "+-"[a >> 32 & 1], (int)(uint32_t)a);
            l += (uint32_t)a;
        }
    }
    for (const Arc &a : ug.g.arc) {
        const uint32_t u = (uint32_t)(a.ul >> 32), v = a.v;
        appendf(out, "L\tutg%.6d%c\t%c\tutg%.6d%c\t%c\t%dM\tSD:i:%d\n", (int)(u >> 1) + 1, "lc"[ug.u[u >> 1].circ], "+-"[u & 1],
                (int)(v >> 1) + 1, "lc"[ug.u[v >> 1].circ], "+-"[v & 1], (int)a.ol, (int)(uint32_t)a.ul);
    }
    for (size_t i = 0; i < ug.u.size(); ++i) {
        const Utg &u = ug.u[i];
        if (u.start == UINT32_MAX) {
            appendf(out, "x\tutg%.6dc\t%d\t%d\n", (int)i + 1, (int)u.len, (int)u.a.size());
        } else {
            const uint32_t c0 = (uint32_t)ug.g.idx[i << 1 | 0], c1 = (uint32_t)ug.g.idx[i << 1 | 1];
            appendf(out, "x\tutg%.6dl\t%d\t%d\t%d\t%d\t%s:%d-%d\t%c\t%s:%d-%d\t%c\n", (int)i + 1, (int)u.len, (int)u.a.size(), (int)c1,
                    (int)c0, d.seq[u.start >> 1].name.c_str(), (int)sub[u.start >> 1].s + 1, (int)sub[u.start >> 1].e,
                    "+-"[u.start & 1], d.seq[u.end >> 1].name.c_str(), (int)sub[u.end >> 1].s + 1, (int)sub[u.end >> 1].e,
                    "+-"[u.end & 1]);
        }
    }
}

void write_text(const char *path, const std::string &s) {
    FILE *f = fopen(path, "wb");
    if (!f) fail(HLMI_EIO, "cannot write %s: %s", path, strerror(errno));
    fwrite(s.data(), 1, s.size(), f);
    if (fclose(f) != 0) fail(HLMI_EIO, "write error on %s", path);
}

}  // namespace

void miniasm_run(const char *paf, const char *reads_fa, int bub_dist, int n_rounds_arg, int max_ext, int min_dp,
                 const char *outfmt, const char *out_path) {
    Opt o;
    o.bub_dist = bub_dist;                 // -d
    o.n_rounds = n_rounds_arg - 1;         // -n (main.c:60)
    o.max_ext = max_ext;                   // -e
    o.min_dp = min_dp;                     // -c
    o.min_ovlp = o.min_span;               // main.c:74
    const std::string fmt = outfmt;
    if (fmt != "ug" && fmt != "sg" && fmt != "paf" && fmt != "bed") fail(HLMI_EINVAL, "outfmt must be ug, sg, paf or bed");
    stat_reset();
    const auto clk = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_prev = clk();
    const auto lap = [&](const char *name) { const double t = clk(); stat_set(name, t - t_prev); t_prev = t; };
    Dict d;
    std::vector<Hit> hit = read_hits(paf, o, d);
    lap("t_graph_read_paf_s");
    size_t n_hits = hit.size();
    // Step 2: 1-pass read selection (main.c:119-126)
    std::vector<Sub> sub = hit_sub(o.min_dp, o.min_iden, hit, n_hits, d.seq.size());
    n_hits = hit_cut(sub, o.min_span, n_hits, hit);
    n_hits = hit_flt(sub, (int)(o.max_hang * 1.5), (int)(o.min_ovlp * .5), n_hits, hit);
    // Step 3: 2-pass read selection (main.c:128-142)
    {
        std::vector<Sub> sub2 = hit_sub(o.min_dp, o.min_iden, hit, n_hits, d.seq.size());
        n_hits = hit_cut(sub2, o.min_span, n_hits, hit);
        for (size_t i = 0; i < d.seq.size(); ++i) {           // ma_sub_merge, hit.c:218-223
            sub[i].e = sub[i].s + sub2[i].e;
            sub[i].s = (sub[i].s + sub2[i].s) & 0x7fffffffu;
        }
    }
    n_hits = hit_contained(o, d, sub, n_hits, hit);
    hit.resize(n_hits);
    lap("t_graph_select_s");
    std::string out;
    if (fmt == "bed") {
        for (size_t i = 0; i < d.seq.size(); ++i)
            if (!d.seq[i].del && sub[i].s != sub[i].e) appendf(out, "%s\t%d\t%d\n", d.seq[i].name.c_str(), (int)sub[i].s, (int)sub[i].e);
    } else if (fmt == "paf") {
        for (const Hit &p : hit) {
            const Sub *rq = &sub[p.qns >> 32], *rt = &sub[p.tn];
            appendf(out, "%s:%d-%d\t%d\t%d\t%d\t%c\t%s:%d-%d\t%d\t%d\t%d\t%d\t%d\t255\n", d.seq[p.qns >> 32].name.c_str(), (int)rq->s + 1,
                    (int)rq->e, (int)(rq->e - rq->s), (int)(uint32_t)p.qns, (int)p.qe, "+-"[p.rev], d.seq[p.tn].name.c_str(),
                    (int)rt->s + 1, (int)rt->e, (int)(rt->e - rt->s), (int)p.ts, (int)p.te, (int)p.ml, (int)p.bl);
        }
    } else {
        Graph sg = sg_gen(o, d, sub, n_hits, hit);
        // Step 4.1 transitive reduction on the GPU (asg.c:148-193), then cleanup + symm on the host
        lap("t_graph_build_s");
        if (arc_del_trans_device(sg.arc, sg.seq, sg.idx, o.gap_fuzz)) { g_cleanup(sg); g_symm(sg); }
        lap("t_graph_reduce_s");
        g_cut_tip(sg, o.max_ext);                               // 4.2
        g_pop_bubble(sg, o.bub_dist);
        for (int i = 0; i <= o.n_rounds; ++i) {                 // 4.3 (main.c:167-173, float arithmetic incl. the NaN of -n 1)
            const float r = o.min_drop + (o.max_drop - o.min_drop) / (float)o.n_rounds * (float)i;
            if (g_del_short(sg, r) != 0) { g_cut_tip(sg, o.max_ext); g_pop_bubble(sg, o.bub_dist); }
        }
        g_cut_internal(sg, 1);                                  // 4.4
        g_cut_biloop(sg, o.max_ext);
        g_cut_tip(sg, o.max_ext);
        g_pop_bubble(sg, o.bub_dist);
        if (g_del_short(sg, o.final_drop) != 0) { g_cut_tip(sg, o.max_ext); g_pop_bubble(sg, o.bub_dist); }   // 4.5
        lap("t_graph_clean_s");
        if (fmt == "ug") {
            UGraph ug = ug_gen(sg);
            if (reads_fa) ug_seq(ug, d, sub, reads_fa);
            lap("t_graph_unitig_seq_s");
            ug_print(ug, d, sub, out);
        } else {
            for (const Arc &p : sg.arc) {                       // ma_sg_print, asm.c:41-55
                const Sub *sq = &sub[p.ul >> 33], *st = &sub[p.v >> 1];
                appendf(out, "L\t%s:%d-%d\t%c\t%s:%d-%d\t%c\t%d:\tL1:i:%d\n", d.seq[p.ul >> 33].name.c_str(), (int)sq->s + 1, (int)sq->e,
                        "+-"[p.ul >> 32 & 1], d.seq[p.v >> 1].name.c_str(), (int)st->s + 1, (int)st->e, "+-"[p.v & 1], (int)p.ol,
                        (int)(uint32_t)p.ul);
            }
        }
    }
    write_text(out_path, out);
    lap("t_graph_write_s");
}

// ---- a18: sfo2overlaps.py, --num_pairs 0 branch ----------------------------------------------------------
void sfo2overlaps_run(const char *in_sfo, const char *out_savage, int num_singles, int num_pairs) {
    (void)num_singles;
    if (num_pairs != 0) fail(HLMI_ESTATE, "sfo2overlaps: paired-end branch (--num_pairs > 0) is not on HyLight's path "
                                           "(HyLight.py:317 passes 0)");
    std::string data = read_file(in_sfo);
    struct Row { long long ia, ib; std::string line; };
    std::vector<Row> rows;
    size_t pos = 0;
    auto split_ws = [](const std::string &l) {
        std::vector<std::string> f;
        size_t p = 0;
        while (p < l.size()) {
            while (p < l.size() && isspace((unsigned char)l[p])) ++p;
            size_t e = p;
            while (e < l.size() && !isspace((unsigned char)l[e])) ++e;
            if (e > p) f.emplace_back(l, p, e - p);
            p = e;
        }
        return f;
    };
    while (pos < data.size()) {
        size_t e = data.find('\n', pos);
        if (e == std::string::npos) e = data.size();
        std::string line = data.substr(pos, e - pos);
        pos = e + 1;
        std::vector<std::string> f = split_ws(line);
        if (f.size() != 8) fail(HLMI_EINVAL, "%s: SFO row needs 8 fields", in_sfo);
        long long ia = atoll(f[0].c_str()), ib = atoll(f[1].c_str());
        std::string body;
        if (ia > ib) {                                          // sfo2overlaps.py:41-47,112-122
            std::vector<std::string> g;
            if (f[2] == "I") g = {f[1], f[0], f[2], f[4], f[3], f[6], f[5], f[7]};
            else g = {f[1], f[0], f[2], std::to_string(-atoll(f[3].c_str())), std::to_string(-atoll(f[4].c_str())), f[6], f[5], f[7]};
            for (size_t i = 0; i < g.size(); ++i) { if (i) body += '\t'; body += g[i]; }
            std::swap(ia, ib);
        } else body = line;
        rows.push_back(Row{ia, ib, std::to_string(ia) + "\t" + std::to_string(ib) + "\t" + body});
    }
    // sort -k1,1n -k2,2n -k3,3n -k4,4n | uniq   (fields 3,4 are the SFO ids again)
    auto num = [&](const Row &r, int k) { return atoll(split_ws(r.line)[k].c_str()); };
    std::vector<std::array<long long, 4>> keys(rows.size());
    for (size_t i = 0; i < rows.size(); ++i) keys[i] = {num(rows[i], 0), num(rows[i], 1), num(rows[i], 2), num(rows[i], 3)};
    std::vector<uint32_t> idx(rows.size());
    for (size_t i = 0; i < idx.size(); ++i) idx[i] = (uint32_t)i;
    std::sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) {
        if (keys[a] != keys[b]) return keys[a] < keys[b];
        return rows[a].line < rows[b].line;
    });
    std::vector<std::string> out;
    const std::string *prev = nullptr;
    for (uint32_t i : idx) {
        const std::string &l = rows[i].line;
        if (prev && *prev == l) continue;
        prev = &l;
        std::vector<std::string> c = split_ws(l);
        if (c.size() != 10) fail(HLMI_EINVAL, "sfo2overlaps: internal row needs 10 fields");
        const long long ida = atoll(c[0].c_str()), idb = atoll(c[1].c_str());
        if (ida == idb) continue;
        const long long oha = atoll(c[5].c_str()), ohb = atoll(c[6].c_str()), ola = atoll(c[7].c_str()), olb = atoll(c[8].c_str());
        const char ori = c[4] == "N" ? '+' : '-';
        const long long ovlen = std::min(ola, olb);
        long long lena, lenb, pos1;
        std::string id1, id2;
        char ori1, ori2;
        if (oha >= 0) {
            lena = ola + oha + (ohb >= 0 ? 0 : -ohb);
            lenb = ohb >= 0 ? olb + ohb : olb;
            id1 = c[0]; id2 = c[1]; pos1 = oha; ori1 = '+'; ori2 = ori;
        } else {
            lena = ohb >= 0 ? ola : ola - ohb;
            lenb = -oha + olb + (ohb >= 0 ? ohb : 0);
            id1 = c[1]; id2 = c[0]; pos1 = -oha; ori1 = ori; ori2 = '+';
        }
        const long long minlen = std::min(lena, lenb);
        if (minlen <= 0) fail(HLMI_EINVAL, "sfo2overlaps: non-positive read length");
        // Python round(): half to even on the exact double 100*ovlen/minlen
        const double x = (double)(100 * ovlen) / (double)minlen;
        long long perc = (long long)std::nearbyint(x);          // FE_TONEAREST = ties to even
        if (perc > 100) perc = 100;
        char buf[256];
        snprintf(buf, sizeof buf, "%s\t%s\t%lld\t-\t-\t%c\t%c\t%lld\t-\t%lld\t-\ts\ts", id1.c_str(), id2.c_str(), pos1, ori1, ori2, perc, ovlen);
        if (out.empty() || out.back() != buf) out.emplace_back(buf);
    }
    write_lines(out_savage, out);
}

}  // namespace hlmi
