// dev_prims.h - device-wide sort / scan / compaction primitives (rocPRIM behind a narrow API;
// everything runs on hlmi::stream()).  Radix sorts are stable (LSD), which the callers rely on
// to build multi-word orders out of successive passes.
#pragma once
#include <cstddef>
#include <cstdint>

namespace hlmi {

// in-place (keys, vals) sort by the key bits [begin_bit, end_bit)
void sort_pairs_u64_u32(uint64_t *keys, uint32_t *vals, size_t n, int begin_bit = 0, int end_bit = 64);
void sort_pairs_u64_u64(uint64_t *keys, uint64_t *vals, size_t n, int begin_bit = 0, int end_bit = 64);
void sort_pairs_u32_u32(uint32_t *keys, uint32_t *vals, size_t n, int begin_bit = 0, int end_bit = 32);
void sort_keys_u64(uint64_t *keys, size_t n, int begin_bit = 0, int end_bit = 64);
// same, without the copy back when the last pass lands in the temporary: `keys` then takes that buffer over
template <typename T> struct DBuf;
void sort_keys_u64(DBuf<uint64_t> &keys, size_t n, int begin_bit = 0, int end_bit = 64);
void sort_pairs_u64_u32(DBuf<uint64_t> &keys, DBuf<uint32_t> &vals, size_t n, int begin_bit = 0, int end_bit = 64);
void sort_pairs_u32_u32(DBuf<uint32_t> &keys, DBuf<uint32_t> &vals, size_t n, int begin_bit = 0, int end_bit = 32);
// small key + 64-bit value (anchor batches: the (target, strand) bits apart from the rest of the anchor)
void sort_pairs_u16_u64(DBuf<uint16_t> &keys, DBuf<uint64_t> &vals, size_t n, int begin_bit = 0, int end_bit = 16);
void sort_pairs_u32_u64(DBuf<uint32_t> &keys, DBuf<uint64_t> &vals, size_t n, int begin_bit = 0, int end_bit = 32);

// out[i] = sum_{j<i} in[j]; returns nothing, total = out[n-1] + in[n-1] (use scan_total)
void exclusive_scan_u32(const uint32_t *in, uint32_t *out, size_t n);
void exclusive_scan_u32_to_u64(const uint32_t *in, uint64_t *out, size_t n);
// writes the indices i with flags[i] != 0 (ascending) to out_idx; returns their count (the only call here that waits
// for the device: the count goes back to the host)
size_t select_flagged_indices(const uint8_t *flags, uint32_t *out_idx, size_t n);
// same, the count goes to *d_count (device): several selections can then share one trip to the host
void select_flagged_indices_async(const uint8_t *flags, uint32_t *out_idx, size_t n, uint32_t *d_count);
// four selections in one go: out_idx[c - 1] receives the indices with cls[i] == c (c = 1..4, ascending), their counts go
// to d_counts[0..3] (device)
void select_classes4_async(const uint8_t *cls, size_t n, uint32_t *out1, uint32_t *out2, uint32_t *out3, uint32_t *out4,
                           uint32_t *d_counts);
// indices i where key[i] >> shift starts a new run (keys grouped): the group boundaries of a sorted / grouped array
size_t select_run_heads_u64(const uint64_t *key, size_t n, int shift, uint32_t *out_idx);
// same for a key kept in two arrays: a new run starts where skey[i] (key_bytes = 2 or 4 wide) or val[i] >> val_shift changes
size_t select_run_heads_split(const void *skey, int key_bytes, const uint64_t *val, size_t n, int val_shift, uint32_t *out_idx);
// number of significant bits of the maximum key value helper
inline int bits_for(uint64_t max_value) {
    int b = 1;
    while (b < 64 && (max_value >> b)) ++b;
    return b;
}

}  // namespace hlmi
