// graph_host.cpp - the order-dependent half of the overlap-graph build and the driver of
// the whole stage (hlmi_miniasm = `miniasm -d D -n N -e E -c C -f reads in.paf`, script/HyLight.py:137,140,171).
//
// What runs here and why it is not a kernel:
//   * reference_sort_order(): miniasm orders overlaps and arcs with an in-place, unstable radix sort; where keys are
//     equal, the order it leaves is a product of that algorithm's data movement and shows in the output (which of two
//     arcs of equal length comes first decides `-p sg` / `-p paf` dumps and, through the cleaners below, in rare cases
//     the graph).  The permutation is therefore computed by running the same movement on (key, index) pairs; the
//     records themselves are permuted on the device (graph_dev.hip).
//   * the graph cleaners (tips, bubbles, short overlaps, internal sequences, bi-loops): each walks the vertices in
//     ascending order and deletes in place, so what a later vertex sees depends on what earlier ones removed
//     (asg.c:238-306, 360-433).  They run on the reduced graph, which is small.
//   * unitig construction, unitig sequences and the GFA text (asm.c:77-286).
//
// The algorithms below are those of miniasm 0.3-r179 (tools/miniasm, MIT license, Copyright (c) 2015 Broad Institute)
// and of klib's ksort.h (MIT license, Copyright (c) 2008, 2011 Attractive Chaos): byte-identical output requires the
// same procedures, restated here over this library's own data structures.  Permission notice of both: "Permission is
// hereby granted, free of charge, to any person obtaining a copy of this software and associated documentation files
// (the "Software"), to deal in the Software without restriction ... The above copyright notice and this permission
// notice shall be included in all copies or substantial portions of the Software."
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <deque>
#include <functional>
#include <numeric>

#include "graph.h"
#include "paf_io.h"

namespace hlmi {

// =================================================================================================
// order of the reference's in-place sort (ksort.h:132-184): most significant byte first, every level distributes the
// records over 256 buckets by following displacement cycles inside the array, buckets of at most 64 records are
// finished by a (stable) insertion sort.
// =================================================================================================
namespace {
struct KeyIdx { uint64_t key; uint32_t idx; };
constexpr ptrdiff_t SMALL_RUN = 64;

void insertion_pass(KeyIdx *lo, KeyIdx *hi) {
    for (KeyIdx *i = lo + 1; i < hi; ++i) {
        if (!(i->key < (i - 1)->key)) continue;
        const KeyIdx moving = *i;
        KeyIdx *j = i;
        for (; j > lo && moving.key < (j - 1)->key; --j) *j = *(j - 1);
        *j = moving;
    }
}

void bucket_pass(KeyIdx *lo, KeyIdx *hi, int shift) {
    KeyIdx *fill[256], *end[256];            // next free slot / end of every bucket's region
    size_t count[256] = {0};
    for (const KeyIdx *p = lo; p != hi; ++p) ++count[p->key >> shift & 255];
    KeyIdx *at = lo;
    for (int b = 0; b < 256; ++b) { fill[b] = at; at += count[b]; end[b] = at; }
    for (int b = 0; b < 256;) {
        if (fill[b] == end[b]) { ++b; continue; }
        int home = (int)(fill[b]->key >> shift & 255);
        if (home == b) { ++fill[b]; continue; }
        // the record at the head of bucket b belongs elsewhere: carry it there, pick up the record it displaces, and
        // so on until a record for bucket b turns up
        KeyIdx carried = *fill[b];
        do {
            std::swap(carried, *fill[home]);
            ++fill[home];
            home = (int)(carried.key >> shift & 255);
        } while (home != b);
        *fill[b]++ = carried;
    }
    if (!shift) return;
    const int next = shift > 8 ? shift - 8 : 0;
    KeyIdx *from = lo;
    for (int b = 0; b < 256; ++b) {
        KeyIdx *to = end[b];
        if (to - from > SMALL_RUN) bucket_pass(from, to, next);
        else if (to - from > 1) insertion_pass(from, to);
        from = to;
    }
}
}  // namespace

void reference_sort_order(const std::vector<uint64_t> &keys, std::vector<uint32_t> &perm) {
    std::vector<KeyIdx> a(keys.size());
    for (size_t i = 0; i < keys.size(); ++i) a[i] = KeyIdx{keys[i], (uint32_t)i};
    if ((ptrdiff_t)a.size() <= SMALL_RUN) insertion_pass(a.data(), a.data() + a.size());
    else bucket_pass(a.data(), a.data() + a.size(), 56);
    perm.resize(a.size());
    for (size_t i = 0; i < a.size(); ++i) perm[i] = a[i].idx;
}

namespace {

// =================================================================================================
// string graph on the host: arcs sorted by (source vertex, length), vertex v = read << 1 | end
// =================================================================================================
class StringGraph {
  public:
    std::vector<Arc> arc;
    std::vector<uint32_t> seq;              // length | deleted << 31
    std::vector<uint64_t> idx;              // per vertex: first arc << 32 | arcs
    bool sorted = false, symmetric = false;

    uint32_t n_vtx() const { return (uint32_t)seq.size() * 2; }
    uint32_t deg(uint32_t v) const { return (uint32_t)idx[v]; }
    uint32_t first(uint32_t v) const { return (uint32_t)(idx[v] >> 32); }
    Arc *out(uint32_t v) { return arc.data() + first(v); }
    const Arc *out(uint32_t v) const { return arc.data() + first(v); }
    bool read_gone(uint32_t v) const { return seq[v >> 1] >> 31; }
    void drop_read(uint32_t r) { seq[r] |= 0x80000000u; }
    void keep_read(uint32_t r) { seq[r] &= 0x7fffffffu; }
    uint32_t live_deg(uint32_t v) const {
        uint32_t n = 0;
        for (uint32_t i = 0, d = deg(v); i < d; ++i) n += !out(v)[i].del();
        return n;
    }

    // asg.c:57-80: forget deleted arcs and arcs of deleted reads, sort once, index
    void rebuild() {
        size_t n = 0;
        for (const Arc &a : arc)
            if (!a.del() && !read_gone(a.src()) && !read_gone(a.v)) arc[n++] = a;
        const bool shrunk = n < arc.size();
        arc.resize(n);
        if (!sorted) {
            std::vector<uint64_t> keys(n);
            for (size_t i = 0; i < n; ++i) keys[i] = arc[i].ul;
            std::vector<uint32_t> perm;
            reference_sort_order(keys, perm);
            std::vector<Arc> s(n);
            for (size_t i = 0; i < n; ++i) s[i] = arc[perm[i]];
            arc.swap(s);
            sorted = true;
        }
        if (shrunk || idx.size() != (size_t)n_vtx()) {
            idx.assign(n_vtx(), 0);
            for (size_t b = 0; b < n;) {
                size_t e = b + 1;
                while (e < n && arc[e].src() == arc[b].src()) ++e;
                idx[arc[b].src()] = (uint64_t)b << 32 | (e - b);
                b = e;
            }
        }
    }
    void mark_arcs(uint32_t v, uint32_t w, bool del) {                 // every arc v -> w (asg.h:53-59)
        for (uint32_t i = 0, d = deg(v); i < d; ++i)
            if (out(v)[i].v == w) out(v)[i].set_del(del);
    }
    void remove_read(uint32_t r) {                                      // the read, its arcs and their partners (asg.h:62-76)
        drop_read(r);
        for (uint32_t v = r << 1; v <= (r << 1 | 1u); ++v)
            for (uint32_t i = 0, d = deg(v); i < d; ++i) {
                out(v)[i].set_del(true);
                mark_arcs(out(v)[i].v ^ 1u, v ^ 1u, true);
            }
    }
    // asg.c:104-145
    void make_symmetric() {
        size_t removed = 0;
        for (uint32_t v = 0; v < n_vtx(); ++v) {                        // several arcs to one neighbour: the first stays
            const uint32_t d = deg(v);
            if (d < 2) continue;
            Arc *a = out(v);
            for (uint32_t i = 1; i < d; ++i)
                for (uint32_t k = 0; k < i; ++k)
                    if (a[k].v == a[i].v) { a[i].set_del(true); ++removed; break; }
        }
        if (removed) rebuild();
        removed = 0;
        for (Arc &a : arc) {                                            // u -> v without v^1 -> u^1
            const uint32_t back_from = a.v ^ 1u, back_to = a.src() ^ 1u;
            bool paired = false;
            for (uint32_t i = 0, d = deg(back_from); i < d && !paired; ++i) paired = out(back_from)[i].v == back_to;
            if (!paired) { a.set_del(true); ++removed; }
        }
        if (removed) rebuild();
        symmetric = true;
    }
};

// ---- a15: cleaners -------------------------------------------------------------------------------------------------
// asg.c:83-101: at every vertex, arcs whose overlap is below ratio x (longest overlap of the vertex) go - the arcs are
// sorted by length, i.e. by descending overlap, so the losers are a suffix
size_t drop_short_overlaps(StringGraph &g, float ratio) {
    size_t dropped = 0;
    for (uint32_t v = 0; v < g.n_vtx(); ++v) {
        const uint32_t d = g.deg(v);
        if (d < 2) continue;
        Arc *a = g.out(v);
        // (uint32_t)(ol * ratio + .499) as the reference evaluates it: float product widened to double; a NaN ratio
        // (`-n 1` makes main.c:168 divide 0.2f by 0) converts to 0 on x86-64
        const double x = (double)((float)(int)a[0].ol() * ratio) + .499;
        const uint32_t floor_ol = std::isnan(x) ? 0u : (uint32_t)(int64_t)x;
        uint32_t last_kept = d - 1;
        while (last_kept >= 1 && a[last_kept].ol() < floor_ol) --last_kept;
        for (uint32_t i = last_kept + 1; i < d; ++i) { a[i].set_del(true); ++dropped; }
    }
    if (dropped) { g.rebuild(); g.make_symmetric(); }
    return dropped;
}

enum class End { Mergeable, Tip, ManyOut, ManyIn };       // asg.c:200-203
// what lies behind vertex v (asg.c:204-221): looks at the live arcs leaving v^1
End look_back(const StringGraph &g, uint32_t v, uint64_t *step) {
    const Arc *a = g.out(v ^ 1u);
    int only = -1;
    uint32_t live = 0;
    for (uint32_t i = 0, d = g.deg(v ^ 1u); i < d; ++i)
        if (!a[i].del()) { only = (int)i; ++live; }
    if (live == 0) return End::Tip;
    if (live > 1) return End::ManyOut;
    if (step) *step = a[only].ul << 32 | a[only].v;
    return g.live_deg(a[only].v ^ 1u) == 1 ? End::Mergeable : End::ManyIn;
}
// follow single links from v for at most max_ext steps (asg.c:223-236); path[0] = v
End walk(const StringGraph &g, uint32_t v, int max_ext, std::vector<uint64_t> &path) {
    path.assign(1, v);
    End e;
    uint64_t step = 0;
    do {
        e = look_back(g, v ^ 1u, &step);
        if (e != End::Mergeable) break;
        path.push_back(step);
        v = (uint32_t)step;
    } while (--max_ext > 0);
    return e;
}
// asg.c:238-272: short dead ends (start = Tip) and short pieces between two branch points (start = ManyIn)
size_t cut_short_paths(StringGraph &g, int max_ext, End start) {
    std::vector<uint64_t> path;
    size_t cut = 0;
    for (uint32_t v = 0; v < g.n_vtx(); ++v) {
        if (g.read_gone(v) || look_back(g, v, nullptr) != start) continue;
        const End stop = walk(g, v, max_ext, path);
        if (start == End::Tip ? stop == End::Mergeable : stop != End::ManyIn) continue;
        for (uint64_t x : path) g.remove_read((uint32_t)x >> 1);
        ++cut;
    }
    if (cut) g.rebuild();
    return cut;
}
// asg.c:274-306: a short path that leaves a branch vertex and comes back to a neighbour of it
size_t cut_biloops(StringGraph &g, int max_ext) {
    std::vector<uint64_t> path;
    size_t cut = 0;
    for (uint32_t v = 0; v < g.n_vtx(); ++v) {
        if (g.read_gone(v) || look_back(g, v, nullptr) != End::ManyIn) continue;
        if (walk(g, v, max_ext, path) != End::ManyOut) continue;
        const uint32_t x = (uint32_t)path.back() ^ 1u;
        uint32_t w = UINT32_MAX;
        for (uint32_t i = 0, d = g.deg(v ^ 1u); i < d; ++i)
            if (!g.out(v ^ 1u)[i].del()) w = g.out(v ^ 1u)[i].v ^ 1u;
        if (w == UINT32_MAX) fail(HLMI_EINVAL, "overlap graph: bi-loop start without a neighbour");
        uint32_t ol_v = 0, ol_x = 0;
        for (uint32_t i = 0, d = g.deg(w); i < d; ++i) {
            const Arc &a = g.out(w)[i];
            if (a.del()) continue;
            if (a.v == x) ol_x = a.ol();
            if (a.v == v) ol_v = a.ol();
        }
        if (ol_v == 0 && ol_x == 0) continue;
        if (ol_v > ol_x) { g.mark_arcs(w, x, true); g.mark_arcs(x ^ 1u, w ^ 1u, true); ++cut; }
    }
    if (cut) g.rebuild();
    return cut;
}

// asg.c:312-433: bubbles.  From a branching vertex the graph is explored in topological order (a vertex is expanded
// once all its live incoming arcs have been seen); if the exploration narrows down to a single vertex again within
// max_dist, everything visited is deleted except the path with the most reads.
class BubblePopper {
    struct Visit { uint32_t parent = 0, dist = 0, reads = 0, waiting = 0; bool seen = false; };
    StringGraph &g;
    std::vector<Visit> at;
    std::vector<uint32_t> ready, dead_ends, touched_vtx, touched_arc;

    void keep_best_path(uint32_t source) {                              // asg.c:337-356
        for (uint32_t x : touched_vtx) g.drop_read(x >> 1);
        for (uint32_t e : touched_arc) {
            Arc &a = g.arc[e];
            a.set_del(true);
            g.mark_arcs(a.v ^ 1u, a.src() ^ 1u, true);
        }
        uint32_t v = ready[0];
        do {
            const uint32_t u = at[v].parent;
            g.keep_read(v >> 1);
            g.mark_arcs(u, v, false);
            g.mark_arcs(v ^ 1u, u ^ 1u, false);
            v = u;
        } while (v != source);
    }
    uint64_t pop_from(uint32_t source, uint32_t max_dist) {
        if (g.read_gone(source) || g.deg(source) < 2) return 0;
        ready.assign(1, source);
        dead_ends.clear(); touched_vtx.clear(); touched_arc.clear();
        at[source].reads = at[source].dist = 0;
        uint32_t pending = 0;
        bool closed = true;
        do {
            const uint32_t v = ready.back();
            ready.pop_back();
            const uint32_t d = at[v].dist, c = at[v].reads, nv = g.deg(v);
            uint32_t i = 0;
            for (; i < nv; ++i) {
                const Arc &a = g.out(v)[i];
                const uint32_t w = a.v, l = a.len();
                if (w == source) { closed = false; break; }             // a cycle through the source
                if (a.del()) continue;
                touched_arc.push_back(g.first(v) + i);
                if (d + l > max_dist) break;                            // too far: i < nv ends the attempt below
                Visit &t = at[w];
                if (!t.seen) {
                    touched_vtx.push_back(w);
                    t.seen = true; t.parent = v; t.dist = d + l;
                    t.waiting = g.live_deg(w ^ 1u);
                    ++pending;
                } else {
                    if (c + 1 > t.reads || (c + 1 == t.reads && d + l > t.dist)) t.parent = v;
                    if (c + 1 > t.reads) t.reads = c + 1;
                    if (d + l < t.dist) t.dist = d + l;
                }
                if (--t.waiting == 0) {
                    (g.deg(w) ? ready : dead_ends).push_back(w);
                    --pending;
                }
            }
            if (!closed || i < nv || ready.empty()) { closed = false; break; }
        } while (ready.size() > 1 || pending);
        uint64_t popped = 0;
        if (closed) {
            keep_best_path(source);
            popped = 1 | (uint64_t)dead_ends.size() << 32;
        }
        for (uint32_t x : touched_vtx) at[x] = Visit();
        return popped;
    }

  public:
    explicit BubblePopper(StringGraph &graph) : g(graph), at(graph.n_vtx()) {}
    uint64_t run(int max_dist) {
        uint64_t popped = 0;
        for (uint32_t v = 0; v < g.n_vtx(); ++v) {
            if (g.deg(v) < 2 || g.read_gone(v)) continue;
            if (g.live_deg(v) > 1) popped += pop_from(v, (uint32_t)max_dist);
        }
        return popped;
    }
};
void pop_bubbles(StringGraph &g, int max_dist) {
    if (!g.symmetric) g.make_symmetric();
    if (BubblePopper(g).run(max_dist)) g.rebuild();
}

// ---- a16: unitigs ----------------------------------------------------------------------------------------------------
struct Unitig {
    uint32_t len = 0, start = UINT32_MAX, end = UINT32_MAX;       // start == UINT32_MAX: circular
    std::vector<uint64_t> reads;                                   // vertex << 32 | bases this read contributes
    std::string bases;
    bool has_bases = false;
    bool circular() const { return start == UINT32_MAX; }
};
struct UnitigGraph { std::vector<Unitig> utg; StringGraph links; };

// asm.c:117-206: maximal non-branching paths, found by walking forward then backward from every unvisited vertex
UnitigGraph build_unitigs(const StringGraph &g) {
    const uint32_t n = g.n_vtx();
    std::vector<int32_t> tag(n, 0);
    UnitigGraph ug;
    std::deque<uint64_t> path;
    auto single = [&](uint32_t v) { return g.deg(v) == 1; };
    auto next = [&](uint32_t v) -> const Arc & { return g.arc[g.first(v)]; };
    for (uint32_t v = 0; v < n; ++v) {
        if (g.read_gone(v) || g.deg(v) == 0 || tag[v]) continue;
        tag[v] = 1;
        path.clear();
        uint32_t start = v, end = v ^ 1u, len = 0;
        for (uint32_t w = v;;) {                                    // forward
            if (!single(w)) break;
            const uint32_t x = next(w).v;
            if (!single(x ^ 1u)) break;
            tag[x] = tag[w ^ 1u] = 1;
            const uint32_t l = next(w).len();
            path.push_back((uint64_t)w << 32 | l);
            end = x ^ 1u; len += l;
            w = x;
            if (x == v) break;
        }
        bool circular = false;
        if (start != (end ^ 1u) || path.empty()) {
            const uint32_t l = g.seq[end >> 1] & 0x7fffffffu;
            path.push_back((uint64_t)(end ^ 1u) << 32 | l);
            len += l;
        } else circular = true;
        if (!circular) {
            for (uint32_t x = v;;) {                                // backward
                if (!single(x ^ 1u)) break;
                const uint32_t w = next(x ^ 1u).v ^ 1u;
                if (!single(w)) break;
                tag[x] = tag[w ^ 1u] = 1;
                const uint32_t l = next(w).len();
                path.push_front((uint64_t)w << 32 | l);
                start = w; len += l;
                x = w;
            }
            tag[start] = tag[end] = 1;
        }
        Unitig u;
        if (!circular) { u.start = start; u.end = end; }
        u.len = len & 0x7fffffffu;
        u.reads.assign(path.begin(), path.end());
        ug.utg.push_back(std::move(u));
    }
    // links between unitig ends (asm.c:176-205)
    std::fill(tag.begin(), tag.end(), -1);
    for (size_t i = 0; i < ug.utg.size(); ++i) {
        if (ug.utg[i].circular()) continue;
        tag[ug.utg[i].start] = (int32_t)(i << 1);
        tag[ug.utg[i].end] = (int32_t)(i << 1 | 1);
    }
    for (const Arc &a : g.arc) {
        if (a.del()) continue;
        const int32_t from = tag[a.src() ^ 1u], to = tag[a.v];
        if (from < 0 || to < 0) continue;
        const uint32_t u = (uint32_t)from ^ 1u;
        int l = (int)ug.utg[u >> 1].len - (int)a.ol();
        if (l < 0) l = 1;
        ug.links.arc.push_back(Arc{(uint64_t)u << 32 | (uint32_t)l, (uint32_t)to, a.ol()});
    }
    ug.links.seq.resize(ug.utg.size());
    for (size_t i = 0; i < ug.utg.size(); ++i) ug.links.seq[i] = ug.utg[i].len;
    ug.links.rebuild();
    return ug;
}

char complement(int c) {                                            // asm.c:220-229
    static const char up[] = "TVGHEFCDIJMLKNOPQYSAABWXRZ";
    if (c >= 'A' && c <= 'Z') return up[c - 'A'];
    if (c >= 'a' && c <= 'z') return (char)(up[c - 'a'] + 32);
    if (c == 96) return 64;
    return (char)c;
}

// asm.c:232-286: every read on a unitig pastes the first `bases it contributes` of its window (reverse-complemented
// on the '-' strand)
void fill_unitig_bases(UnitigGraph &ug, const GraphState &st, const char *reads_path) {
    struct Slot { uint32_t utg = 0, rev = 0, at = 0, n = 0; };
    std::vector<Slot> slot(st.name.size());
    for (size_t i = 0; i < ug.utg.size(); ++i) {
        Unitig &u = ug.utg[i];
        u.bases.assign(u.len, 'N');
        u.has_bases = true;
        uint32_t at = 0;
        for (uint64_t x : u.reads) {
            slot[x >> 33] = Slot{(uint32_t)i, (uint32_t)(x >> 32) & 1u, at, (uint32_t)x};
            at += (uint32_t)x;
        }
    }
    std::unordered_map<std::string, uint32_t> id_of;
    for (uint32_t r = 0; r < st.name.size(); ++r)
        if (slot[r].n) id_of.emplace(st.read_name(r), r);
    // only the reads that sit on a unitig bring their bases along
    std::string key;
    const std::function<bool(std::string_view)> wanted = [&](std::string_view name) {
        key.assign(name);
        return id_of.count(key) != 0;
    };
    SeqSet reads;
    read_seqs_subset(reads_path, &wanted, reads);
    for (size_t k = 0; k < reads.size(); ++k) {
        const auto it = id_of.find(reads.names[k]);
        if (it == id_of.end()) continue;
        const uint32_t r = it->second;
        const Slot &s = slot[r];
        const ReadWin &w = st.win[r];
        const uint32_t wl = w.e - w.s;
        if (wl > reads.len(k)) fail(HLMI_EINVAL, "read %s is shorter in %s than in the PAF", reads.names[k].c_str(), reads_path);
        const char *b = reads.bases.data() + reads.off[k] + w.s;
        std::string &dst = ug.utg[s.utg].bases;
        for (uint32_t i = 0; i < s.n && s.at + i < dst.size(); ++i) {
            if (!s.rev) dst[s.at + i] = b[i];
            else {
                const int c = (uint8_t)b[wl - 1 - i];
                dst[s.at + i] = c >= 128 ? 'N' : complement(c);
            }
        }
    }
}

// ---- text --------------------------------------------------------------------------------------------------------------
void put(std::string &out, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    const int n = vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (n < (int)sizeof buf) { out.append(buf, (size_t)n); return; }
    std::vector<char> big((size_t)n + 1);
    va_start(ap, fmt);
    vsnprintf(big.data(), big.size(), fmt, ap);
    va_end(ap);
    out.append(big.data(), (size_t)n);
}
// "<name>:<start+1>-<end>" of a read's window, as every dump prints it
std::string window_tag(const GraphState &st, uint32_t r) {
    std::string s = st.read_name(r);
    put(s, ":%d-%d", (int)st.win[r].s + 1, (int)st.win[r].e);
    return s;
}

void print_gfa(const UnitigGraph &ug, const GraphState &st, std::string &out) {        // asm.c:77-112
    char name[32];
    for (size_t i = 0; i < ug.utg.size(); ++i) {
        const Unitig &u = ug.utg[i];
        snprintf(name, sizeof name, "utg%.6d%c", (int)i + 1, "lc"[u.circular()]);
        out += "S\t"; out += name; out += '\t';
        if (u.has_bases) out += u.bases; else out += '*';
        put(out, "\tLN:i:%d\n", (int)u.len);
        uint32_t at = 0;
        for (uint64_t x : u.reads) {
            put(out, "a\t%s\t%d\t%s\t%c\t%d\n", name, (int)at, window_tag(st, (uint32_t)(x >> 33)).c_str(), "+-"[x >> 32 & 1], (int)(uint32_t)x);
            at += (uint32_t)x;
        }
    }
    for (const Arc &a : ug.links.arc) {
        const uint32_t u = a.src(), v = a.v;
        put(out, "L\tutg%.6d%c\t%c\tutg%.6d%c\t%c\t%dM\tSD:i:%d\n", (int)(u >> 1) + 1, "lc"[ug.utg[u >> 1].circular()], "+-"[u & 1],
            (int)(v >> 1) + 1, "lc"[ug.utg[v >> 1].circular()], "+-"[v & 1], (int)a.ol(), (int)a.len());
    }
    for (size_t i = 0; i < ug.utg.size(); ++i) {
        const Unitig &u = ug.utg[i];
        if (u.circular()) { put(out, "x\tutg%.6dc\t%d\t%d\n", (int)i + 1, (int)u.len, (int)u.reads.size()); continue; }
        put(out, "x\tutg%.6dl\t%d\t%d\t%d\t%d\t%s\t%c\t%s\t%c\n", (int)i + 1, (int)u.len, (int)u.reads.size(),
            (int)ug.links.deg((uint32_t)i << 1 | 1u), (int)ug.links.deg((uint32_t)i << 1), window_tag(st, u.start >> 1).c_str(),
            "+-"[u.start & 1], window_tag(st, u.end >> 1).c_str(), "+-"[u.end & 1]);
    }
}

void write_text(const char *path, const std::string &s) {
    std::vector<std::string_view> one;
    // write_lines adds the newline of the last line itself
    if (!s.empty()) one.emplace_back(s.data(), s.size() - (s.back() == '\n' ? 1 : 0));
    write_lines(path, one);
}

}  // namespace

void miniasm_run(const char *paf, const char *reads_fa, int bub_dist, int n_rounds_arg, int max_ext, int min_dp,
                 const char *outfmt, const char *out_path) {
    GraphOpt o;
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

    GraphState st;
    graph_device(paf, o, fmt, st);                                  // a9-a14
    lap("t_graph_device_s");
    std::string out;
    if (fmt == "bed") {                                             // main.c:14-22
        for (uint32_t r = 0; r < st.name.size(); ++r)
            if (!st.win[r].del && st.win[r].s != st.win[r].e)
                put(out, "%s\t%d\t%d\n", st.read_name(r).c_str(), (int)st.win[r].s, (int)st.win[r].e);
    } else if (fmt == "paf") {                                      // main.c:24-30
        for (const Ovl &h : st.ovl)
            put(out, "%s\t%d\t%d\t%d\t%c\t%s\t%d\t%d\t%d\t%d\t%d\t255\n", window_tag(st, h.q).c_str(), (int)(st.win[h.q].e - st.win[h.q].s),
                (int)h.qs, (int)h.qe, "+-"[h.ml_rev >> 31], window_tag(st, h.t).c_str(), (int)(st.win[h.t].e - st.win[h.t].s), (int)h.ts,
                (int)h.te, (int)(h.ml_rev & 0x7fffffffu), (int)h.bl);
    } else if (st.have_graph) {
        StringGraph g;
        g.arc.swap(st.arc);
        g.seq.swap(st.seq_len);
        g.sorted = true;
        g.symmetric = st.symmetric;
        g.rebuild();
        // main.c:160-187 (steps 4.2-4.5)
        cut_short_paths(g, o.max_ext, End::Tip);
        pop_bubbles(g, o.bub_dist);
        for (int i = 0; i <= o.n_rounds; ++i) {                     // float arithmetic as written there, NaN of `-n 1` included
            const float r = o.min_drop + (o.max_drop - o.min_drop) / (float)o.n_rounds * (float)i;
            if (drop_short_overlaps(g, r)) { cut_short_paths(g, o.max_ext, End::Tip); pop_bubbles(g, o.bub_dist); }
        }
        cut_short_paths(g, 1, End::ManyIn);
        cut_biloops(g, o.max_ext);
        cut_short_paths(g, o.max_ext, End::Tip);
        pop_bubbles(g, o.bub_dist);
        if (drop_short_overlaps(g, o.final_drop)) { cut_short_paths(g, o.max_ext, End::Tip); pop_bubbles(g, o.bub_dist); }
        lap("t_graph_clean_s");
        if (fmt == "ug") {
            UnitigGraph ug = build_unitigs(g);
            if (reads_fa) fill_unitig_bases(ug, st, reads_fa);
            lap("t_graph_unitig_seq_s");
            print_gfa(ug, st, out);
        } else {                                                    // asm.c:41-55
            for (const Arc &a : g.arc)
                put(out, "L\t%s\t%c\t%s\t%c\t%d:\tL1:i:%d\n", window_tag(st, a.src() >> 1).c_str(), "+-"[a.src() & 1],
                    window_tag(st, a.v >> 1).c_str(), "+-"[a.v & 1], (int)a.ol(), (int)a.len());
        }
    }
    write_text(out_path, out);
    lap("t_graph_write_s");
}

}  // namespace hlmi
