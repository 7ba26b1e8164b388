// row_text.h - final rows of a stage as text, formatted on the device (row_text.hip).
#pragma once
#include <string>
#include <vector>

#include "common.h"
#include "paf_io.h"

namespace hlmi {

struct DevNames {                    // the names of a job's reads by name id (= strcmp rank), resident in HBM
    DBuf<char> text;
    DBuf<uint64_t> off;
    size_t n = 0;
    void upload(const std::vector<std::string> &names_by_id);
};

struct RowText {
    std::string text;                // the rows one after the other, no separators
    std::vector<uint64_t> at;        // row i: text[at[i] .. at[i] + len[i])
    std::vector<uint32_t> len;       // 0: dropped by the identity test (slr2:146); ROW_TEXT_TO_HOST: the host formatter takes it
    std::vector<uint32_t> key;       // column 12 x 10^4 (paf_io.h: sort_scored_lines), 0xffffffff: not plain digits
};
constexpr uint32_t ROW_TEXT_TO_HOST = 0xffffffffu;

// rows d_recs[idx[i]] with the X digit sums xsum[i] (FilterOut) -> the 14 columns of filter_overlap_slr2.py:142-151
void format_rows_device(const PafRec *d_recs, const std::vector<uint32_t> &idx, const std::vector<uint32_t> &xsum, const DevNames &names,
                        double iden, RowText &out);

}  // namespace hlmi
