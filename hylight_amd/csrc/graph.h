// graph.h - miniasm overlap-graph build and sfo2overlaps (a18).
#pragma once
#include "common.h"

namespace hlmi {
void miniasm_run(const char *paf, const char *reads_fa, int bub_dist, int n_rounds_arg, int max_ext, int min_dp,
                 const char *outfmt, const char *out_path);
void sfo2overlaps_run(const char *in_sfo, const char *out_savage, int num_singles, int num_pairs);
}  // namespace hlmi
