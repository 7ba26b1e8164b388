// dev_prims.hip - rocPRIM-backed implementation of dev_prims.h
#include "dev_prims.h"

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

#include <utility>

#include "common.h"
#include "wave_ops.h"

namespace hlmi {

// Temporaries are pooled device buffers (common.h): releasing one while kernels of this stream still use it is
// safe, the next user is ordered behind them on the same stream - no host synchronisation in here unless a
// value goes back to the host.
namespace {
constexpr int WG = 256;

template <typename K, typename V>
void sort_pairs_impl(K *keys, V *vals, size_t n, int b0, int b1) {
    if (n < 2) return;
    DBuf<K> k2(n);
    DBuf<V> v2(n);
    rocprim::double_buffer<K> dk(keys, k2.p);
    rocprim::double_buffer<V> dv(vals, v2.p);
    size_t tmp_bytes = 0;
    HIP_CHECK(rocprim::radix_sort_pairs(nullptr, tmp_bytes, dk, dv, n, b0, b1, stream()));
    DBuf<char> tmp(tmp_bytes ? tmp_bytes : 1);
    HIP_CHECK(rocprim::radix_sort_pairs(tmp.p, tmp_bytes, dk, dv, n, b0, b1, stream()));
    if (dk.current() != keys)
        HIP_CHECK(hipMemcpyAsync(keys, dk.current(), n * sizeof(K), hipMemcpyDeviceToDevice, stream()));
    if (dv.current() != vals)
        HIP_CHECK(hipMemcpyAsync(vals, dv.current(), n * sizeof(V), hipMemcpyDeviceToDevice, stream()));
}

// per 64 flags: how many are set
__global__ __launch_bounds__(WG) void flag_count_kernel(const uint8_t *flags, size_t n, uint32_t *cnt) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const unsigned long long m = __ballot(i < n && flags[i] != 0);
    if ((threadIdx.x & 63) == 0 && (i >> 6) < (n + 63) / 64) cnt[i >> 6] = (uint32_t)__popcll(m);
}
__global__ __launch_bounds__(WG) void flag_scatter_kernel(const uint8_t *flags, size_t n, const uint32_t *off, uint32_t *out_idx,
                                                           uint32_t *total) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const bool f = i < n && flags[i] != 0;
    const unsigned long long m = __ballot(f);
    if (f) out_idx[off[i >> 6] + __popcll(m & ((1ull << (threadIdx.x & 63)) - 1ull))] = (uint32_t)i;
    if (i == n - 1) *total = off[i >> 6] + (uint32_t)__popcll(m);      // the wave that holds the last element
}
}  // namespace

// (64-bit key + 32-bit value, the SNP event sort: 1024 x 6 per sort block measured 9.9 ms against 10.9 ms with the
// default 512 x 16; 1024 x 8, 512 x 8 and 512 x 12 were slower)
using PairsOnesweep = rocprim::radix_sort_config<
    rocprim::default_config, rocprim::default_config,
    rocprim::radix_sort_onesweep_config<rocprim::kernel_config<1024, 16>, rocprim::kernel_config<1024, 6>, 8,
                                        rocprim::block_radix_rank_algorithm::match>>;

void sort_pairs_u64_u32(uint64_t *k, uint32_t *v, size_t n, int b0, int b1) { sort_pairs_impl(k, v, n, b0, b1); }
// (64-bit key + 64-bit value: the index and the anchor batches that do not fit one word; same shape as PairsOnesweep)
void sort_pairs_u64_u64(uint64_t *keys, uint64_t *vals, size_t n, int b0, int b1) {
    if (n < 2) return;
    DBuf<uint64_t> k2(n), v2(n);
    rocprim::double_buffer<uint64_t> dk(keys, k2.p), dv(vals, v2.p);
    size_t tmp_bytes = 0;
    HIP_CHECK(rocprim::radix_sort_pairs<PairsOnesweep>(nullptr, tmp_bytes, dk, dv, n, b0, b1, stream()));
    DBuf<char> tmp(tmp_bytes ? tmp_bytes : 1);
    HIP_CHECK(rocprim::radix_sort_pairs<PairsOnesweep>(tmp.p, tmp_bytes, dk, dv, n, b0, b1, stream()));
    if (dk.current() != keys) HIP_CHECK(hipMemcpyAsync(keys, dk.current(), n * 8, hipMemcpyDeviceToDevice, stream()));
    if (dv.current() != vals) HIP_CHECK(hipMemcpyAsync(vals, dv.current(), n * 8, hipMemcpyDeviceToDevice, stream()));
}
void sort_pairs_u32_u32(uint32_t *k, uint32_t *v, size_t n, int b0, int b1) { sort_pairs_impl(k, v, n, b0, b1); }
void sort_pairs_u32_u32(DBuf<uint32_t> &keys, DBuf<uint32_t> &vals, size_t n, int b0, int b1) {
    if (n < 2) return;
    DBuf<uint32_t> k2(keys.n), v2(vals.n);
    rocprim::double_buffer<uint32_t> dk(keys.p, k2.p), dv(vals.p, v2.p);
    size_t tmp_bytes = 0;
    HIP_CHECK(rocprim::radix_sort_pairs(nullptr, tmp_bytes, dk, dv, n, b0, b1, stream()));
    DBuf<char> tmp(tmp_bytes ? tmp_bytes : 1);
    HIP_CHECK(rocprim::radix_sort_pairs(tmp.p, tmp_bytes, dk, dv, n, b0, b1, stream()));
    if (dk.current() != keys.p) std::swap(keys, k2);
    if (dv.current() != vals.p) std::swap(vals, v2);
}

// (small key + 64-bit value: the library's own gfx950 shape measured 113.9 ms per C3 step against 119.3 with PairsOnesweep;
// 1024 x 8, 512 x 12, 512 x 16, 256 x 16 and 1024 x 4 were slower)
template <typename K>
static void sort_small_key_pairs(DBuf<K> &keys, DBuf<uint64_t> &vals, size_t n, int b0, int b1) {
    if (n < 2) return;
    DBuf<K> k2(keys.n);
    DBuf<uint64_t> v2(vals.n);
    rocprim::double_buffer<K> dk(keys.p, k2.p);
    rocprim::double_buffer<uint64_t> dv(vals.p, v2.p);
    size_t tmp_bytes = 0;
    HIP_CHECK(rocprim::radix_sort_pairs(nullptr, tmp_bytes, dk, dv, n, b0, b1, stream()));
    DBuf<char> tmp(tmp_bytes ? tmp_bytes : 1);
    HIP_CHECK(rocprim::radix_sort_pairs(tmp.p, tmp_bytes, dk, dv, n, b0, b1, stream()));
    if (dk.current() != keys.p) std::swap(keys, k2);
    if (dv.current() != vals.p) std::swap(vals, v2);
}
void sort_pairs_u16_u64(DBuf<uint16_t> &k, DBuf<uint64_t> &v, size_t n, int b0, int b1) { sort_small_key_pairs(k, v, n, b0, b1); }
void sort_pairs_u32_u64(DBuf<uint32_t> &k, DBuf<uint64_t> &v, size_t n, int b0, int b1) { sort_small_key_pairs(k, v, n, b0, b1); }

void sort_keys_u64(uint64_t *keys, size_t n, int b0, int b1) {
    if (n < 2) return;
    DBuf<uint64_t> k2(n);
    rocprim::double_buffer<uint64_t> dk(keys, k2.p);
    size_t tmp_bytes = 0;
    HIP_CHECK(rocprim::radix_sort_keys(nullptr, tmp_bytes, dk, n, b0, b1, stream()));
    DBuf<char> tmp(tmp_bytes ? tmp_bytes : 1);
    HIP_CHECK(rocprim::radix_sort_keys(tmp.p, tmp_bytes, dk, n, b0, b1, stream()));
    if (dk.current() != keys)
        HIP_CHECK(hipMemcpyAsync(keys, dk.current(), n * sizeof(uint64_t), hipMemcpyDeviceToDevice, stream()));
}

void sort_pairs_u64_u32(DBuf<uint64_t> &keys, DBuf<uint32_t> &vals, size_t n, int b0, int b1) {
    if (n < 2) return;
    DBuf<uint64_t> k2(keys.n);
    DBuf<uint32_t> v2(vals.n);
    rocprim::double_buffer<uint64_t> dk(keys.p, k2.p);
    rocprim::double_buffer<uint32_t> dv(vals.p, v2.p);
    size_t tmp_bytes = 0;
    HIP_CHECK(rocprim::radix_sort_pairs<PairsOnesweep>(nullptr, tmp_bytes, dk, dv, n, b0, b1, stream()));
    DBuf<char> tmp(tmp_bytes ? tmp_bytes : 1);
    HIP_CHECK(rocprim::radix_sort_pairs<PairsOnesweep>(tmp.p, tmp_bytes, dk, dv, n, b0, b1, stream()));
    if (dk.current() != keys.p) std::swap(keys, k2);
    if (dv.current() != vals.p) std::swap(vals, v2);
}

// Onesweep shape for the big keys-only sorts (the anchor batches): 1024 x 8 keys per sort block, 1024 x 16 per histogram
// block measured 18.1 ms per C2 step against 20.1 ms with the library's gfx950 default (512 x 8); 512 x 16, 256 x 16,
// 1024 x 4 and 256 x 32 were slower.
using KeysOnesweep = rocprim::radix_sort_config<
    rocprim::default_config, rocprim::default_config,
    rocprim::radix_sort_onesweep_config<rocprim::kernel_config<1024, 16>, rocprim::kernel_config<1024, 8>, 8,
                                        rocprim::block_radix_rank_algorithm::match>>;

void sort_keys_u64(DBuf<uint64_t> &keys, size_t n, int b0, int b1) {
    if (n < 2) return;
    DBuf<uint64_t> k2(keys.n);
    rocprim::double_buffer<uint64_t> dk(keys.p, k2.p);
    size_t tmp_bytes = 0;
    HIP_CHECK(rocprim::radix_sort_keys<KeysOnesweep>(nullptr, tmp_bytes, dk, n, b0, b1, stream()));
    DBuf<char> tmp(tmp_bytes ? tmp_bytes : 1);
    HIP_CHECK(rocprim::radix_sort_keys<KeysOnesweep>(tmp.p, tmp_bytes, dk, n, b0, b1, stream()));
    if (dk.current() != keys.p) std::swap(keys, k2);      // the result sits in the other buffer: keep that one
}

void exclusive_scan_u32(const uint32_t *in, uint32_t *out, size_t n) {
    if (!n) return;
    size_t tmp_bytes = 0;
    HIP_CHECK(rocprim::exclusive_scan(nullptr, tmp_bytes, in, out, 0u, n, rocprim::plus<uint32_t>(), stream()));
    DBuf<char> tmp(tmp_bytes ? tmp_bytes : 1);
    HIP_CHECK(rocprim::exclusive_scan(tmp.p, tmp_bytes, in, out, 0u, n, rocprim::plus<uint32_t>(), stream()));
}

void exclusive_scan_u32_to_u64(const uint32_t *in, uint64_t *out, size_t n) {
    if (!n) return;
    size_t tmp_bytes = 0;
    auto in64 = rocprim::make_transform_iterator(in, [] __device__(uint32_t v) { return (uint64_t)v; });
    HIP_CHECK(rocprim::exclusive_scan(nullptr, tmp_bytes, in64, out, (uint64_t)0, n, rocprim::plus<uint64_t>(), stream()));
    DBuf<char> tmp(tmp_bytes ? tmp_bytes : 1);
    HIP_CHECK(rocprim::exclusive_scan(tmp.p, tmp_bytes, in64, out, (uint64_t)0, n, rocprim::plus<uint64_t>(), stream()));
}

// Three small launches instead of rocprim::select: set bits per 64 flags (ballot), their exclusive scan, scatter by
// ballot rank.  The flags are read twice (1 B each); on the 5e7-element head arrays of the anchor batches this is
// several times quicker than the look-back partition, and most calls of a step are such selections.
// the same two passes with the predicate "key[i] >> shift differs from key[i-1] >> shift" (first element: true)
// computed from the sorted keys themselves: no flag array in between
__global__ __launch_bounds__(WG) void head_count_kernel(const uint64_t *key, size_t n, int shift, uint32_t *cnt,
                                                         unsigned long long *mask) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const bool f = i < n && (i == 0 || (key[i] >> shift) != (key[i - 1] >> shift));
    const unsigned long long m = __ballot(f);
    if ((threadIdx.x & 63) == 0 && (i >> 6) < (n + 63) / 64) { cnt[i >> 6] = (uint32_t)__popcll(m); mask[i >> 6] = m; }
}
template <typename K>
__global__ __launch_bounds__(WG) void head_count_split_kernel(const K *skey, const uint64_t *val, size_t n, int shift, uint32_t *cnt,
                                                               unsigned long long *mask) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const bool f = i < n && (i == 0 || skey[i] != skey[i - 1] || (val[i] >> shift) != (val[i - 1] >> shift));
    const unsigned long long m = __ballot(f);
    if ((threadIdx.x & 63) == 0 && (i >> 6) < (n + 63) / 64) { cnt[i >> 6] = (uint32_t)__popcll(m); mask[i >> 6] = m; }
}
// bit i of the low 16 bits of x -> bit 4 i
__device__ __forceinline__ unsigned long long spread16x4(unsigned long long x) {
    x &= 0xffffull;
    x = (x | x << 24) & 0x000000ff000000ffull;
    x = (x | x << 12) & 0x000f000f000f000full;
    x = (x | x << 6) & 0x0303030303030303ull;
    x = (x | x << 3) & 0x1111111111111111ull;
    return x;
}
// four elements per thread, flags f0..f3 of a thread's elements as ballots b0..b3 (bit planes by lane): the word of the
// wave's elements 64 j .. 64 j + 63 interleaves the planes' 16-bit slices j
__device__ __forceinline__ unsigned long long planes_word(unsigned long long b0, unsigned long long b1, unsigned long long b2,
                                                          unsigned long long b3, int j) {
    const int sh = 16 * j;
    return spread16x4(b0 >> sh) | spread16x4(b1 >> sh) << 1 | spread16x4(b2 >> sh) << 2 | spread16x4(b3 >> sh) << 3;
}
// The 16-bit form, four elements per thread: three wide loads instead of sixteen narrow ones (the predecessor of a
// thread's first element comes from the lane below, lane 0 fetches its own).  The four ballots are bit planes by lane;
// word j of the wave's 256 flags interleaves their 16-bit slices j (lanes 0 .. 3 build one word each).
__global__ __launch_bounds__(WG) void head_count_split4_kernel(const uint16_t *skey, const uint64_t *val, size_t n, int shift,
                                                                uint32_t *cnt, unsigned long long *mask) {
    const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const int lane = (int)(threadIdx.x & 63);
    const size_t i0 = 4 * t;
    uint64_t v[4] = {0, 0, 0, 0};
    uint32_t s[4] = {0, 0, 0, 0};
    if (i0 + 3 < n) {
        const ulonglong2 a = ((const ulonglong2 *)val)[2 * t], b = ((const ulonglong2 *)val)[2 * t + 1];
        const uint2 ss = ((const uint2 *)skey)[t];
        v[0] = a.x >> shift; v[1] = a.y >> shift; v[2] = b.x >> shift; v[3] = b.y >> shift;
        s[0] = ss.x & 0xffffu; s[1] = ss.x >> 16; s[2] = ss.y & 0xffffu; s[3] = ss.y >> 16;
    } else {
        for (int k = 0; k < 4; ++k) if (i0 + k < n) { v[k] = val[i0 + k] >> shift; s[k] = skey[i0 + k]; }
    }
    uint32_t pv_lo = (uint32_t)wave_shr1((int)(uint32_t)v[3], 0), pv_hi = (uint32_t)wave_shr1((int)(uint32_t)(v[3] >> 32), 0);
    uint32_t ps = (uint32_t)wave_shr1((int)s[3], 0);
    if (lane == 0 && i0 > 0 && i0 < n) { const uint64_t x = val[i0 - 1] >> shift; pv_lo = (uint32_t)x; pv_hi = (uint32_t)(x >> 32); ps = skey[i0 - 1]; }
    const uint64_t pv = (uint64_t)pv_hi << 32 | pv_lo;
    const bool f0 = i0 < n && (i0 == 0 || s[0] != ps || v[0] != pv);
    const bool f1 = i0 + 1 < n && (s[1] != s[0] || v[1] != v[0]);
    const bool f2 = i0 + 2 < n && (s[2] != s[1] || v[2] != v[1]);
    const bool f3 = i0 + 3 < n && (s[3] != s[2] || v[3] != v[2]);
    const unsigned long long b0 = __ballot(f0), b1 = __ballot(f1), b2 = __ballot(f2), b3 = __ballot(f3);
    if (lane < 4) {
        const unsigned long long m = planes_word(b0, b1, b2, b3, lane);
        const size_t w = (t - (size_t)lane) / 16 + (size_t)lane;
        if (w < (n + 63) / 64) { cnt[w] = (uint32_t)__popcll(m); mask[w] = m; }
    }
}
// second pass: the ballots of the first (8 B per 64 keys) instead of the keys again
__global__ __launch_bounds__(WG) void head_scatter_kernel(const unsigned long long *mask, size_t n, const uint32_t *off,
                                                           uint32_t *out_idx, uint32_t *total) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long m = mask[i >> 6];
    const int lane = (int)(threadIdx.x & 63);
    if ((m >> lane) & 1ull) out_idx[off[i >> 6] + __popcll(m & ((1ull << lane) - 1ull))] = (uint32_t)i;
    if (i == n - 1) *total = off[i >> 6] + (uint32_t)__popcll(m);
}
// the same with one thread per 64 keys: for sparse heads (anchor groups of hundreds: one set bit in six words)
__global__ __launch_bounds__(WG) void head_scatter_words_kernel(const unsigned long long *mask, size_t nw, const uint32_t *off,
                                                                 uint32_t *out_idx, uint32_t *total) {
    const size_t w = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (w >= nw) return;
    unsigned long long m = mask[w];
    uint32_t o = off[w];
    if (w == nw - 1) *total = o + (uint32_t)__popcll(m);
    while (m) {
        out_idx[o++] = (uint32_t)(w * 64 + (size_t)(__ffsll((long long)m) - 1));
        m &= m - 1;
    }
}

size_t select_run_heads_u64(const uint64_t *key, size_t n, int shift, uint32_t *out_idx) {
    if (!n) return 0;
    if (n >= (1ull << 32)) fail(HLMI_EINVAL, "select_run_heads_u64: more than 2^32 elements");
    const size_t nw = (n + 63) / 64;
    DBuf<uint32_t> cnt(nw), off(nw), total(1);
    DBuf<unsigned long long> mask(nw);
    const dim3 grid(cdiv(n, WG));
    hipLaunchKernelGGL(head_count_kernel, grid, dim3(WG), 0, stream(), key, n, shift, cnt.p, mask.p);
    exclusive_scan_u32(cnt.p, off.p, nw);
    hipLaunchKernelGGL(head_scatter_kernel, grid, dim3(WG), 0, stream(), mask.p, n, off.p, out_idx, total.p);
    HIP_CHECK(hipGetLastError());
    return (size_t)download_one(total.p);
}

size_t select_run_heads_split(const void *skey, int key_bytes, const uint64_t *val, size_t n, int val_shift, uint32_t *out_idx) {
    if (!n) return 0;
    if (n >= (1ull << 32)) fail(HLMI_EINVAL, "select_run_heads_split: more than 2^32 elements");
    const size_t nw = (n + 63) / 64;
    DBuf<uint32_t> cnt(nw), off(nw), total(1);
    DBuf<unsigned long long> mask(nw);
    const dim3 grid(cdiv(n, WG));
    if (key_bytes == 2)
        hipLaunchKernelGGL(head_count_split4_kernel, dim3(cdiv(cdiv(n, (size_t)4), WG)), dim3(WG), 0, stream(), (const uint16_t *)skey, val, n,
                           val_shift, cnt.p, mask.p);
    else
        hipLaunchKernelGGL(head_count_split_kernel<uint32_t>, grid, dim3(WG), 0, stream(), (const uint32_t *)skey, val, n, val_shift, cnt.p, mask.p);
    exclusive_scan_u32(cnt.p, off.p, nw);
    // sparse heads: one thread per mask word; dense ones (short reads: groups of a few anchors): one per key
    uint32_t h_last[2];
    HIP_CHECK(hipMemcpyAsync(&h_last[0], off.p + (nw - 1), 4, hipMemcpyDeviceToHost, stream()));
    HIP_CHECK(hipMemcpyAsync(&h_last[1], cnt.p + (nw - 1), 4, hipMemcpyDeviceToHost, stream()));
    sync();
    const size_t n_heads = (size_t)h_last[0] + h_last[1];
    if (n_heads * 16 < n)
        hipLaunchKernelGGL(head_scatter_words_kernel, dim3(cdiv(nw, WG)), dim3(WG), 0, stream(), mask.p, nw, off.p, out_idx, total.p);
    else
        hipLaunchKernelGGL(head_scatter_kernel, grid, dim3(WG), 0, stream(), mask.p, n, off.p, out_idx, total.p);
    HIP_CHECK(hipGetLastError());
    return n_heads;
}

void select_flagged_indices_async(const uint8_t *flags, uint32_t *out_idx, size_t n, uint32_t *d_count) {
    if (!n) { HIP_CHECK(hipMemsetAsync(d_count, 0, 4, stream())); return; }
    if (n >= (1ull << 32)) fail(HLMI_EINVAL, "select_flagged_indices: more than 2^32 elements");
    const size_t nw = (n + 63) / 64;
    DBuf<uint32_t> cnt(nw), off(nw);
    const dim3 grid(cdiv(n, WG));
    hipLaunchKernelGGL(flag_count_kernel, grid, dim3(WG), 0, stream(), flags, n, cnt.p);
    exclusive_scan_u32(cnt.p, off.p, nw);
    hipLaunchKernelGGL(flag_scatter_kernel, grid, dim3(WG), 0, stream(), flags, n, off.p, out_idx, d_count);
    HIP_CHECK(hipGetLastError());
}

// class c - 1 owns cnt / mask / off entries [(c - 1) nw, c nw): one scan serves the four selections;
// four elements per thread (one 32-bit load instead of four byte loads; lane 4 k + j builds word j of class k + 1)
__global__ __launch_bounds__(WG) void class_count4_kernel(const uint8_t *cls, size_t n, size_t nw, uint32_t *cnt, unsigned long long *mask) {
    const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const int lane = (int)(threadIdx.x & 63);
    const size_t i0 = 4 * t;
    uint32_t c4 = 0;
    if (i0 + 3 < n) c4 = ((const uint32_t *)cls)[t];
    else for (int k = 0; k < 4; ++k) if (i0 + k < n) c4 |= (uint32_t)cls[i0 + k] << (8 * k);
    unsigned long long b[4][4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) b[k][e] = __ballot(((c4 >> (8 * e)) & 0xffu) == (uint32_t)(k + 1));
    if (lane < 16) {
        const int k = lane >> 2, j = lane & 3;
        unsigned long long m = 0;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) if (kk == k) m = planes_word(b[kk][0], b[kk][1], b[kk][2], b[kk][3], j);
        const size_t w = (t - (size_t)lane) / 16 + (size_t)j;
        if (w < nw) { cnt[(size_t)k * nw + w] = (uint32_t)__popcll(m); mask[(size_t)k * nw + w] = m; }
    }
}
struct Out4 { uint32_t *p[4]; };
// scatter by mask word: one thread per (class, 64 elements) - the class lists are sparse (a few set bits per word), the
// byte array is not read again
__global__ __launch_bounds__(WG) void class_scatter_words_kernel(size_t nw, const uint32_t *cnt, const unsigned long long *mask,
                                                                  const uint32_t *off, Out4 out, uint32_t *d_counts) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < 4) {                                          // the four totals
        const size_t k = i;
        const uint32_t end = k < 3 ? off[(k + 1) * nw] : off[4 * nw - 1] + cnt[4 * nw - 1];
        d_counts[k] = end - off[k * nw];
    }
    if (i >= 4 * nw) return;
    const size_t k = i / nw, w = i - k * nw;
    unsigned long long m = mask[i];
    uint32_t o = off[i] - off[k * nw];
    uint32_t *dst = out.p[k];
    while (m) {
        dst[o++] = (uint32_t)(w * 64 + (size_t)(__ffsll((long long)m) - 1));
        m &= m - 1;
    }
}
void select_classes4_async(const uint8_t *cls, size_t n, uint32_t *out1, uint32_t *out2, uint32_t *out3, uint32_t *out4,
                           uint32_t *d_counts) {
    if (!n) { HIP_CHECK(hipMemsetAsync(d_counts, 0, 16, stream())); return; }
    if (n >= (1ull << 32)) fail(HLMI_EINVAL, "select_classes4: more than 2^32 elements");
    const size_t nw = (n + 63) / 64;
    DBuf<uint32_t> cnt(4 * nw), off(4 * nw);
    DBuf<unsigned long long> mask(4 * nw);
    hipLaunchKernelGGL(class_count4_kernel, dim3(cdiv(cdiv(n, (size_t)4), WG)), dim3(WG), 0, stream(), cls, n, nw, cnt.p, mask.p);
    exclusive_scan_u32(cnt.p, off.p, 4 * nw);
    hipLaunchKernelGGL(class_scatter_words_kernel, dim3(cdiv(4 * nw, WG)), dim3(WG), 0, stream(), nw, cnt.p, mask.p, off.p,
                       Out4{{out1, out2, out3, out4}}, d_counts);
    HIP_CHECK(hipGetLastError());
}

size_t select_flagged_indices(const uint8_t *flags, uint32_t *out_idx, size_t n) {
    if (!n) return 0;
    DBuf<uint32_t> total(1);
    select_flagged_indices_async(flags, out_idx, n, total.p);
    return (size_t)download_one(total.p);
}

}  // namespace hlmi
