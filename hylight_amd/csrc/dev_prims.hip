// dev_prims.hip - rocPRIM-backed implementation of dev_prims.h
#include "dev_prims.h"

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_select.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

#include "common.h"

namespace hlmi {

namespace {
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
    sync();  // temporaries die here
}
}  // namespace

void sort_pairs_u64_u32(uint64_t *k, uint32_t *v, size_t n, int b0, int b1) { sort_pairs_impl(k, v, n, b0, b1); }
void sort_pairs_u64_u64(uint64_t *k, uint64_t *v, size_t n, int b0, int b1) { sort_pairs_impl(k, v, n, b0, b1); }
void sort_pairs_u32_u32(uint32_t *k, uint32_t *v, size_t n, int b0, int b1) { sort_pairs_impl(k, v, n, b0, b1); }

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
    sync();
}

void exclusive_scan_u32(const uint32_t *in, uint32_t *out, size_t n) {
    if (!n) return;
    size_t tmp_bytes = 0;
    HIP_CHECK(rocprim::exclusive_scan(nullptr, tmp_bytes, in, out, 0u, n, rocprim::plus<uint32_t>(), stream()));
    DBuf<char> tmp(tmp_bytes ? tmp_bytes : 1);
    HIP_CHECK(rocprim::exclusive_scan(tmp.p, tmp_bytes, in, out, 0u, n, rocprim::plus<uint32_t>(), stream()));
    sync();
}

void exclusive_scan_u32_to_u64(const uint32_t *in, uint64_t *out, size_t n) {
    if (!n) return;
    size_t tmp_bytes = 0;
    auto in64 = rocprim::make_transform_iterator(in, [] __device__(uint32_t v) { return (uint64_t)v; });
    HIP_CHECK(rocprim::exclusive_scan(nullptr, tmp_bytes, in64, out, (uint64_t)0, n, rocprim::plus<uint64_t>(), stream()));
    DBuf<char> tmp(tmp_bytes ? tmp_bytes : 1);
    HIP_CHECK(rocprim::exclusive_scan(tmp.p, tmp_bytes, in64, out, (uint64_t)0, n, rocprim::plus<uint64_t>(), stream()));
    sync();
}

size_t select_flagged_indices(const uint8_t *flags, uint32_t *out_idx, size_t n) {
    if (!n) return 0;
    DBuf<size_t> cnt(1);
    size_t tmp_bytes = 0;
    rocprim::counting_iterator<uint32_t> iota(0);
    HIP_CHECK(rocprim::select(nullptr, tmp_bytes, iota, flags, out_idx, cnt.p, n, stream()));
    DBuf<char> tmp(tmp_bytes ? tmp_bytes : 1);
    HIP_CHECK(rocprim::select(tmp.p, tmp_bytes, iota, flags, out_idx, cnt.p, n, stream()));
    return download_one(cnt.p);
}

}  // namespace hlmi
