// graph.h - miniasm overlap-graph build and sfo2overlaps (a18).
#pragma once
#include <vector>

#include "common.h"

namespace hlmi {

// string-graph arc / vertex records; same field widths as tools/miniasm/asg.h:7-15 so that the
// unstable in-place radix sort moves identical 16-byte records
struct Arc {
    uint64_t ul;              // (vertex << 32) | arc length
    uint32_t v;               // target vertex
    uint32_t ol : 31, del : 1;
};
struct GSeq {
    uint32_t len : 31, del : 1;
};

// a14 on the GPU: Myers transitive reduction (tools/miniasm/asg.c:148-193).  arcs sorted by ul with
// idx[v] = start << 32 | count.  Sets Arc::del, returns the number of reduced arcs.
uint32_t arc_del_trans_device(std::vector<Arc> &arc, const std::vector<GSeq> &seq, const std::vector<uint64_t> &idx,
                              int fuzz);

void miniasm_run(const char *paf, const char *reads_fa, int bub_dist, int n_rounds_arg, int max_ext, int min_dp,
                 const char *outfmt, const char *out_path);
void sfo2overlaps_run(const char *in_sfo, const char *out_savage, int num_singles, int num_pairs);

}  // namespace hlmi
