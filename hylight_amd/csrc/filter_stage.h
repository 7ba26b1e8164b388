// filter_stage.h - device implementation of the PAF filter chain.
#pragma once
#include <vector>

#include "common.h"

namespace hlmi {

struct FilterCfg {
    // v4 window filter as the worker invokes it: `-len 30 -oh 3`, default -iden 0.6 (slr2:51, v4:19-21)
    int v4_min_len = 30;
    double v4_min_iden = 0.6;
    int v4_min_o = 3;
    // pass 2 (slr2:18-28 as called from utils.py:54)
    int len_over = 3000;
    int mc = 2;
    double thre = 0.0025;
    int min_o = 4;
    bool long_mode = true;
    uint32_t chunk_id_bound = 0;   // 1 + largest PafRec::chunk value that can occur (0: number of chunks of the call)
    bool reference_order = true;   // rows come back in the reference's write order; false: any order (the stage sorts
                                   // everything by score and text afterwards, so the order here cannot show)
};

// Rows kept by pass 2 BEFORE the score2 >= iden test (that test needs the "%.4f" text and is
// applied by the host formatter), in the order the reference writes them: per chunk, in the
// intermediate sort order of slr2:57.  x_digit_sum[i] belongs to rows[i].
struct FilterOut {
    std::vector<uint32_t> rows;
    std::vector<uint32_t> x_digit_sum;
    size_t n_after_v4 = 0, n_events = 0, n_pairs = 0;
};

// recs: n rows in stream order, grouped by chunk (chunk_row_start has n_chunks+1 entries).
void filter_stage_device(const PafRec *d_recs, size_t n, const uint32_t *d_ops,
                         const std::vector<uint64_t> &chunk_row_start, const FilterCfg &cfg, FilterOut &out);

// a4/a17 alone: keep[i] = 1 for rows the window filter prints.  variant 3 or 4.
void window_filter_device(const PafRec *d_recs, size_t n, const std::vector<uint64_t> &chunk_row_start,
                          int variant, int min_len, double min_iden, int min_o, uint8_t *d_keep);

// filter_ovlp_inline.py: keep[i] = 1 for the longest overlap of each pair per window;
// first_of[i] = row index of the pair's first surviving row (the position the kept row is printed at)
void ovlp_inline_device(const PafRec *d_recs, size_t n, int min_len, double min_iden, int o, double r, uint8_t *d_keep,
                        uint32_t *d_first_of);

// CIGAR ops per chunk id (sizes the chunk groups a caller filters at a time)
std::vector<uint64_t> ops_per_chunk(const PafRec *d_recs, size_t n, uint32_t n_chunk_ids);

// host copies of the rows recs[idx[i]]
std::vector<PafRec> download_rows(const PafRec *d_recs, const std::vector<uint32_t> &idx);

}  // namespace hlmi
