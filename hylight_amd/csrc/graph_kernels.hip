// graph_kernels.hip - a14: Myers transitive reduction of the string graph on the GPU
// (tools/miniasm/asg.c:148-193, fuzz = gap_fuzz = 1000).
//
// The reference keeps one byte of mark per vertex and walks the vertices sequentially; the marks of
// vertex v only ever concern v's own out-neighbours, and the pass reads arc targets / lengths and
// seq[].del and writes only v's own arc del flags, so vertices are independent.  Here every vertex
// gets one thread and a per-vertex state strip (one byte per out-arc, in a scratch array parallel to
// the arc array) instead of the global mark array: state 1 = neighbour, 2 = reachable through another
// neighbour within L = len(longest arc) + fuzz.  Arc order inside a vertex (ascending length) and the
// "skip neighbours already marked 2" rule are kept, so the result is identical to the reference.
// HBM-bound pointer chasing over 16-byte arcs: reads ~ sum_v sum_{w in N(v)} deg(w) arcs.
#include "graph.h"

namespace hlmi {

namespace {
__global__ void del_trans_kernel(const Arc *arc, const GSeq *seq, const uint64_t *idx, uint32_t n_vtx, int fuzz,
                                 uint8_t *state, uint8_t *del, uint32_t *n_reduced) {
    uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n_vtx) return;
    const uint32_t nv = (uint32_t)idx[v];
    if (nv == 0) return;
    const uint64_t b = idx[v] >> 32;
    const Arc *av = arc + b;
    uint8_t *st = state + b;
    uint32_t red = 0;
    if (seq[v >> 1].del) {
        for (uint32_t i = 0; i < nv; ++i) del[b + i] = 1;
        atomicAdd(n_reduced, nv);
        return;
    }
    // mark[w] = 1 for every neighbour: all arcs to the same w share one mark -> keep it on the first arc to w
    for (uint32_t i = 0; i < nv; ++i) st[i] = 1;
    const uint32_t L = (uint32_t)av[nv - 1].ul + (uint32_t)fuzz;
    for (uint32_t i = 0; i < nv; ++i) {
        const uint32_t w = av[i].v;
        // mark of w = state of the first arc v->w
        uint32_t fw = i;
        for (uint32_t k = 0; k < i; ++k) if (av[k].v == w) { fw = k; break; }
        if (st[fw] != 1) continue;
        const uint32_t nw = (uint32_t)idx[w];
        const Arc *aw = arc + (idx[w] >> 32);
        const uint32_t li = (uint32_t)av[i].ul;
        for (uint32_t j = 0; j < nw && (uint32_t)aw[j].ul + li <= L; ++j) {
            const uint32_t x = aw[j].v;
            for (uint32_t k = 0; k < nv; ++k)
                if (av[k].v == x) { st[k] = 2; break; }      // first arc to x carries the mark
        }
    }
    // asg.c:181-184 resets mark[w] while it deletes, so of several arcs v->w only the first one is reduced
    for (uint32_t i = 0; i < nv; ++i) {
        bool first = true;
        for (uint32_t k = 0; k < i; ++k) if (av[k].v == av[i].v) { first = false; break; }
        if (first && st[i] == 2) { del[b + i] = 1; ++red; }
    }
    if (red) atomicAdd(n_reduced, red);
}
}  // namespace

uint32_t arc_del_trans_device(std::vector<Arc> &arc, const std::vector<GSeq> &seq, const std::vector<uint64_t> &idx,
                              int fuzz) {
    require_device();
    const size_t n = arc.size();
    const uint32_t n_vtx = (uint32_t)seq.size() * 2;
    if (!n || !n_vtx) return 0;
    DBuf<Arc> d_arc;
    DBuf<GSeq> d_seq;
    DBuf<uint64_t> d_idx;
    d_arc.upload(arc);
    d_seq.upload(seq);
    d_idx.upload(idx);
    DBuf<uint8_t> state(n), del(n);
    DBuf<uint32_t> cnt(1);
    state.zero();
    del.zero();
    cnt.zero();
    {
        KTimer kt("arc_del_trans");
        hipLaunchKernelGGL(del_trans_kernel, dim3(cdiv(n_vtx, 256)), dim3(256), 0, stream(), d_arc.p, d_seq.p, d_idx.p, n_vtx,
                           fuzz, state.p, del.p, cnt.p);
    }
    HIP_CHECK(hipGetLastError());
    std::vector<uint8_t> h = del.download(n);
    const uint32_t red = download_one(cnt.p);
    for (size_t i = 0; i < n; ++i) if (h[i]) arc[i].del = 1;
    ktimer_flush();
    return red;
}

}  // namespace hlmi
