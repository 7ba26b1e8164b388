// sketch.hip - S1 of the overlapper spec: homopolymer-compressed (k,w)-minimizers on the GPU.
//
// Published algorithm: Li 2016 (minimap) Alg. 1 "compute minimizers" with the HPC k-mers of
// Li 2018 section 2.1; restated sequentially in oracle/ava_oracle.c:oracle_sketch_codes.
// Data-parallel formulation (every step is a coalesced streaming pass, HBM bound):
//   1. slot ends   : a "slot" is one homopolymer run (or one base without -H) or one ambiguous
//                    base; flag the last base of every slot                      (1 B read/base)
//   2. compaction  : slot -> base index                                          (scan + scatter)
//   3. k-mer pass  : per slot, the k preceding run symbols -> forward / reverse 2-bit words,
//                    invertible hash of the canonical one, span in original bases
//   4. window pass : slot j is a minimizer iff it attains the minimum of some window of w
//                    consecutive valid slots that contains it (all ties kept)
//   5. compaction  : minimizers in (read, position) order, 16 B each
#include "ava.h"
#include "dev_prims.h"

namespace hlmi {

namespace {
constexpr int WG = 256;
inline dim3 grid1(size_t n) { return dim3(cdiv(n ? n : 1, WG)); }

__global__ void encode_kernel(uint8_t *b, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint8_t c = b[i], v = 4;
    switch (c) {
        case 'A': case 'a': v = 0; break;
        case 'C': case 'c': v = 1; break;
        case 'G': case 'g': v = 2; break;
        case 'T': case 't': case 'U': case 'u': v = 3; break;
        default: v = 4;
    }
    b[i] = v;
}

__global__ void mark_read_last_kernel(const uint64_t *off, size_t n_reads, uint8_t *flag) {
    size_t r = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    if (off[r + 1] > off[r]) flag[off[r + 1] - 1] = 1;
}

__global__ void slot_flag_kernel(const uint8_t *codes, size_t n, int hpc, uint8_t *flag) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint8_t c = codes[i];
    bool end = flag[i] || c > 3 || !hpc || (i + 1 < n && codes[i + 1] != c);
    flag[i] = end ? 1 : 0;
}

__device__ __forceinline__ size_t lower_bound_u32(const uint32_t *a, size_t n, uint64_t v) {
    size_t lo = 0, hi = n;
    while (lo < hi) {
        size_t m = (lo + hi) >> 1;
        if ((uint64_t)a[m] < v) lo = m + 1; else hi = m;
    }
    return lo;
}

// first slot of every read (and of the virtual read n): number of slots that end before off[r]
__global__ void read_slot0_kernel(const uint32_t *slot_pos, size_t ns, const uint64_t *off, size_t n_reads,
                                  uint32_t *rslot0) {
    size_t r = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (r > n_reads) return;
    rslot0[r] = (uint32_t)lower_bound_u32(slot_pos, ns, off[r]);
}

__global__ void slot_read_kernel(const uint32_t *rslot0, size_t n_reads, size_t ns, uint32_t *slot_rid) {
    size_t s = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (s >= ns) return;
    // last read r with rslot0[r] <= s
    size_t lo = 0, hi = n_reads;   // invariant: rslot0[lo] <= s < rslot0[hi]
    while (hi - lo > 1) {
        size_t m = (lo + hi) >> 1;
        if (rslot0[m] <= s) lo = m; else hi = m;
    }
    slot_rid[s] = (uint32_t)lo;
}

__device__ __forceinline__ uint64_t hash64(uint64_t key, uint64_t mask) {   // Li 2016 section 2.2
    key = (~key + (key << 21)) & mask;
    key = key ^ key >> 24;
    key = ((key + (key << 3)) + (key << 8)) & mask;
    key = key ^ key >> 14;
    key = ((key + (key << 2)) + (key << 4)) & mask;
    key = key ^ key >> 28;
    key = (key + (key << 31)) & mask;
    return key;
}

// The symbols of the workgroup's slots and of the w + k - 2 before them sit in LDS (one gather per slot instead of up
// to w + k - 1 dependent pairs of reads per slot).
__global__ __launch_bounds__(WG) void kmer_kernel(const uint8_t *codes, const uint64_t *off, const uint32_t *slot_pos,
                                                   const uint32_t *slot_rid, const uint32_t *rslot0, size_t ns, int k, int w, int hpc,
                                                   uint64_t *sx, uint32_t *spz, uint8_t *sl) {
    __shared__ uint8_t s_c[WG + 96];            // w + k - 1 <= 91
    const int need = w + k - 1;
    const size_t j0 = blockIdx.x * (size_t)WG;
    for (int i = threadIdx.x; i < WG + need - 1; i += WG) {
        const long long si = (long long)j0 - (need - 1) + i;
        s_c[i] = si >= 0 && (size_t)si < ns ? codes[slot_pos[si]] : (uint8_t)4;
    }
    __syncthreads();
    size_t s = j0 + threadIdx.x;
    if (s >= ns) return;
    const uint32_t r = slot_rid[s];
    const size_t s0 = rslot0[r];
    int l = 0;                                  // consecutive symbol slots ending at s, capped at w+k-1
    uint64_t fwd = 0, rev = 0;
    const uint64_t mask = (1ULL << 2 * k) - 1;
    const int shift = 2 * (k - 1);
    while (l < need && s >= s0 + (size_t)l) {
        uint8_t c = s_c[threadIdx.x + need - 1 - l];
        if (c > 3) break;
        if (l < k) {                            // symbol at distance l from the k-mer's last symbol
            fwd |= (uint64_t)c << (2 * l);
            rev |= (uint64_t)(3 ^ c) << (shift - 2 * l);
        }
        ++l;
    }
    uint64_t x = ~0ull;
    uint32_t pz = 0;
    if (l >= k) {
        const uint64_t pos = slot_pos[s];
        int64_t prev_end = (s >= s0 + (size_t)k) ? (int64_t)slot_pos[s - k] : (int64_t)off[r] - 1;
        int64_t span = hpc ? (int64_t)pos - prev_end : k;
        fwd &= mask;
        if (span < 256 && fwd != rev) {
            int z = fwd < rev ? 0 : 1;
            x = hash64(z ? rev : fwd, mask) << 8 | (uint64_t)span;
            pz = (uint32_t)(pos - off[r]) << 1 | (uint32_t)z;
        }
    }
    sx[s] = x;
    spz[s] = pz;
    sl[s] = (uint8_t)l;
}

// A slot is picked when its key is the minimum of some fully valid window of w slots that contains it.  One tile of
// keys per workgroup in LDS (WG slots + w - 1 on both sides): the window minima once per window end, then every slot
// looks at the w windows it belongs to - 2 w LDS reads per slot instead of w * w reads of global memory.
__global__ __launch_bounds__(WG) void pick_kernel(const uint64_t *sx, const uint8_t *sl, const uint32_t *slot_rid,
                                                   const uint32_t *rslot0, size_t ns, int k, int w, uint8_t *pick) {
    __shared__ uint64_t s_x[WG + 128];            // keys of slots j0 - (w-1) .. j0 + WG + w - 2   (w <= 64)
    __shared__ uint64_t s_wm[WG + 64];            // window minima of the windows ending at j0 .. j0 + WG + w - 2
    const size_t j0 = blockIdx.x * (size_t)WG;
    const long long lo = (long long)j0 - (w - 1);
    for (int i = threadIdx.x; i < WG + 2 * (w - 1); i += WG) {
        const long long s = lo + i;
        s_x[i] = s >= 0 && (size_t)s < ns ? sx[s] : ~0ull;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < WG + w - 1; i += WG) {            // window ending at slot j0 + i = tile index i + w - 1
        const size_t s = j0 + (size_t)i;
        uint64_t mn = ~0ull;
        if (s < ns && (int)sl[s] >= w + k - 1)                      // window [s-w+1, s] fully valid
            for (int t = 0; t < w; ++t) { const uint64_t v = s_x[i + w - 1 - t]; mn = v < mn ? v : mn; }
        else mn = 0;                                                // (never equals a live key: see below)
        s_wm[i] = mn;
    }
    __syncthreads();
    const size_t j = j0 + threadIdx.x;
    if (j >= ns) return;
    const uint64_t xj = s_x[threadIdx.x + w - 1];
    bool sel = false;
    if (xj != ~0ull) {
        const size_t send = rslot0[slot_rid[j] + 1];   // one past the read's last slot
        for (int d = 0; d < w && j + (size_t)d < send && !sel; ++d) {
            const uint64_t m = s_wm[threadIdx.x + d];
            sel = m == xj && (int)sl[j + (size_t)d] >= w + k - 1;
        }
    }
    pick[j] = sel ? 1 : 0;
}

__global__ void emit_mz_kernel(const uint32_t *midx, size_t n, const uint64_t *sx, const uint32_t *spz,
                               const uint32_t *slot_rid, uint32_t rid_base, Mz *out) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t s = midx[i];
    out[i].x = sx[s];
    out[i].y = (uint64_t)(rid_base + slot_rid[s]) << 32 | spz[s];
}

__global__ void read_count_kernel(const uint32_t *midx, size_t n_mz, const uint32_t *rslot0, size_t n_reads,
                                  uint32_t *counts) {
    size_t r = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    counts[r] = (uint32_t)(lower_bound_u32(midx, n_mz, rslot0[r + 1]) - lower_bound_u32(midx, n_mz, rslot0[r]));
}
}  // namespace

static void finish_upload(const char *bases, const std::vector<uint64_t> &off, DevReads &out) {
    out.n = off.size() - 1;
    out.total = off.back();
    out.h_off = off;
    out.alloc_codes();
    if (out.total) {
        HIP_CHECK(hipMemcpyAsync(out.codes(), bases, out.total, hipMemcpyHostToDevice, stream()));
        // (a launch holds fewer than 2^32 work-items per dimension: a grid for 5 Gbases wraps silently and leaves the
        //  tail as ASCII - found by the full-size C5 run, round 3)
        for (uint64_t o = 0; o < out.total; o += 1ull << 30) {
            const uint64_t len = std::min<uint64_t>(1ull << 30, out.total - o);
            hipLaunchKernelGGL(encode_kernel, grid1(len), dim3(WG), 0, stream(), out.codes() + o, len);
        }
        HIP_CHECK(hipGetLastError());
    }
    out.off.upload(off);
    sync();
}

void upload_reads(const SeqSet &s, size_t lo, size_t hi, DevReads &out) {
    std::vector<uint64_t> off(hi - lo + 1);
    for (size_t i = lo; i <= hi; ++i) off[i - lo] = s.off[i] - s.off[lo];
    finish_upload(s.bases.data() + s.off[lo], off, out);       // (a contiguous stretch of the file's bases: no copy)
}

void upload_reads(const SeqSet &s, const std::vector<uint32_t> &ids, DevReads &out) {
    std::vector<uint64_t> off(ids.size() + 1, 0);
    std::string bases;
    for (size_t i = 0; i < ids.size(); ++i) {
        bases.append(s.bases, s.off[ids[i]], s.off[ids[i] + 1] - s.off[ids[i]]);
        off[i + 1] = bases.size();
    }
    finish_upload(bases.data(), off, out);
}

__global__ void copy_reads_kernel(const uint8_t *src, const uint64_t *src_off, const uint32_t *ids, const uint64_t *dst_off,
                                  uint8_t *dst) {
    const uint32_t r = blockIdx.x;
    const uint64_t s0 = src_off[ids[r]], d0 = dst_off[r], len = dst_off[r + 1] - d0;
    for (uint64_t i = threadIdx.x; i < len; i += blockDim.x) dst[d0 + i] = src[s0 + i];
}

void subset_reads_device(const DevReads &all, const std::vector<uint32_t> &ids, DevReads &out) {
    std::vector<uint64_t> off(ids.size() + 1, 0);
    for (size_t i = 0; i < ids.size(); ++i) off[i + 1] = off[i] + (all.h_off[ids[i] + 1] - all.h_off[ids[i]]);
    out.n = ids.size();
    out.total = off.back();
    out.h_off = off;
    out.alloc_codes();
    out.off.upload(off);
    if (!ids.empty()) {
        DBuf<uint32_t> d_ids;
        d_ids.upload(ids);
        hipLaunchKernelGGL(copy_reads_kernel, dim3((unsigned)ids.size()), dim3(WG), 0, stream(), all.codes(), all.off.p,
                           d_ids.p, out.off.p, out.codes());
        HIP_CHECK(hipGetLastError());
        sync();
    }
}

// shared body: returns the compacted slot indices and fills per-read counts
static int64_t sketch_core(const DevReads &r, int k, int w, int hpc, uint32_t rid_base, Mz *d_out, int64_t cap,
                           uint32_t *d_counts, DBuf<Mz> *own_out) {
    if (!(k & 1) || k < 3 || k > 28 || w < 1 || w > 64) fail(HLMI_EINVAL, "sketch needs odd k in [3,28], w in [1,64]");
    const size_t n = r.total;
    if (!r.n) return 0;
    if (n >= (1ull << 32)) fail(HLMI_EINVAL, "sketch of %zu bases in one piece (the kernels index bases with 32 bits: the job sketches in parts)", n);
    if (!n) {
        HIP_CHECK(hipMemsetAsync(d_counts, 0, r.n * 4, stream()));
        sync();                  // the caller reads the counts from another stream (torch) right after the call
        return 0;
    }
    DBuf<uint8_t> flag(n);
    flag.zero();
    hipLaunchKernelGGL(mark_read_last_kernel, grid1(r.n), dim3(WG), 0, stream(), r.off.p, r.n, flag.p);
    hipLaunchKernelGGL(slot_flag_kernel, grid1(n), dim3(WG), 0, stream(), r.codes(), n, hpc, flag.p);
    DBuf<uint32_t> slot_pos(n);
    const size_t ns = select_flagged_indices(flag.p, slot_pos.p, n);
    flag.release();
    KTimer kt_all("sketch_kmer_window");
    DBuf<uint32_t> rslot0(r.n + 1), slot_rid(ns);
    hipLaunchKernelGGL(read_slot0_kernel, grid1(r.n + 1), dim3(WG), 0, stream(), slot_pos.p, ns, r.off.p, r.n, rslot0.p);
    hipLaunchKernelGGL(slot_read_kernel, grid1(ns), dim3(WG), 0, stream(), rslot0.p, r.n, ns, slot_rid.p);
    DBuf<uint64_t> sx(ns);
    DBuf<uint32_t> spz(ns);
    DBuf<uint8_t> sl(ns), pick(ns);
    hipLaunchKernelGGL(kmer_kernel, grid1(ns), dim3(WG), 0, stream(), r.codes(), r.off.p, slot_pos.p, slot_rid.p,
                       rslot0.p, ns, k, w, hpc, sx.p, spz.p, sl.p);
    hipLaunchKernelGGL(pick_kernel, grid1(ns), dim3(WG), 0, stream(), sx.p, sl.p, slot_rid.p, rslot0.p, ns, k, w, pick.p);
    HIP_CHECK(hipGetLastError());
    DBuf<uint32_t> midx(ns);
    const size_t nm = select_flagged_indices(pick.p, midx.p, ns);
    if (own_out) {
        own_out->alloc(nm ? nm : 1);
        d_out = own_out->p;
    } else if ((int64_t)nm > cap) {
        fail(HLMI_EINVAL, "sketch buffer too small: need %zu entries, have %lld", nm, (long long)cap);
    }
    hipLaunchKernelGGL(emit_mz_kernel, grid1(nm), dim3(WG), 0, stream(), midx.p, nm, sx.p, spz.p, slot_rid.p, rid_base,
                       d_out);
    hipLaunchKernelGGL(read_count_kernel, grid1(r.n), dim3(WG), 0, stream(), midx.p, nm, rslot0.p, r.n, d_counts);
    HIP_CHECK(hipGetLastError());
    sync();
    stat_add("sketch_bases", (double)n);
    stat_add("sketch_minimizers", (double)nm);
    return (int64_t)nm;
}

void sketch_device(const DevReads &r, int k, int w, int hpc, uint32_t rid_base, DevSketch &out) {
    out.counts.alloc(r.n ? r.n : 1);
    out.n = (size_t)sketch_core(r, k, w, hpc, rid_base, nullptr, 0, out.counts.p, &out.mz);
}

int64_t sketch_device_into(const DevReads &r, int k, int w, int hpc, uint32_t rid_base, Mz *d_out, int64_t cap,
                           uint32_t *d_counts) {
    return sketch_core(r, k, w, hpc, rid_base, d_out, cap, d_counts, nullptr);
}

}  // namespace hlmi
