// paf_io.h - host-side text I/O for the path: PAF parsing, FASTA/FASTQ reading, final row
// formatting and the GNU-sort orders the reference pipes its text through.
#pragma once
#include <functional>
#include <string>
#include <string_view>
#include <vector>

#include "common.h"

namespace hlmi {

// Name dictionary: equal names <=> equal ids (the reference compares qseqid == sseqid as
// strings and keys its dicts by "a:b" of the sorted name pair, filter_overlap_slr2.py:88).
struct NameDict {
    std::vector<std::string> names;
    std::unordered_map<std::string, uint32_t> index;
    uint32_t put(std::string_view s);
};

struct PafText {
    std::string data;                 // whole file
    std::vector<size_t> line_off;     // start of each line
    std::vector<uint32_t> line_len;   // length without the newline
    std::vector<PafRec> recs;         // one per line
    std::vector<uint32_t> ops;        // concatenated CIGAR ops of the last field
    NameDict dict;
    std::string_view line(size_t i) const { return std::string_view(data).substr(line_off[i], line_len[i]); }
};

// Reads and parses a PAF file.  Rows need >= 11 tab-separated columns with integer columns
// 2-4 and 7-11 (anything else is HLMI_EINVAL - the reference's int() would raise there).
// chunk is set to 0 and `tie` to the rank of the whole line in byte order (GNU sort's
// last-resort comparison under LC_ALL=C).
void read_paf(const char *path, PafText &out, bool need_tie_rank);

// Sequence file reader with the record detection of kseq (FASTA '>' / FASTQ '@', multi-line).
struct SeqSet {
    std::vector<std::string> names;
    std::string bases;               // concatenated, as in the file (no case folding)
    std::vector<uint64_t> off;       // size n+1
    std::vector<uint32_t> first_line;  // 0-based line number of each record's header line
    uint64_t n_lines = 0;            // `wc -l` of the file (utils.py:44)
    size_t size() const { return names.size(); }
    uint32_t len(size_t i) const { return (uint32_t)(off[i + 1] - off[i]); }
};
void read_seqs(const char *path, SeqSet &out);
// the same scan, but only the records `want` accepts bring their bases along (the others keep their name and have
// length 0; n_lines is not counted)
void read_seqs_subset(const char *path, const std::function<bool(std::string_view)> *want, SeqSet &out);

// Final 14-column row of filter_overlap_slr2.py:142-151 (with the trailing TAB); returns
// false when the row is dropped by the identity test `float(score2) < iden` (slr2:146).
// "%.4f" of v, the bytes printf writes, into dst (>= 64 bytes); returns the length (tests/capi/fixed4_check.cpp)
size_t format_fixed4(double v, char *dst);
// sort_key (optional): column 12 as an integer (value x 10^4) for sort_scored_lines, 0xffffffff when it is not plain digits.
constexpr uint32_t SCORE_KEY_LIMIT = 1u << 22;                // 419.4304
bool format_scored_row(const PafRec &r, const std::string &qname, const std::string &tname,
                       uint32_t x_digit_sum, double iden, std::string &out, uint32_t *sort_key = nullptr);

// `sort -k12 -nr` (utils.py:54,69): numeric descending on column 12, ties by reversed
// whole-line byte order.  Lines carry no newline.
void sort_scored_lines(std::vector<std::string> &lines);
// the same on views into text the caller keeps alive (the stage formats its rows into a few large buffers: one
// allocation per thread instead of one per line)
// (keys: the sort_key of every line, from format_scored_row - spares the parse of column 12)
void sort_scored_lines(std::vector<std::string_view> &lines, const std::vector<uint32_t> *keys = nullptr);

void write_lines(const char *path, const std::vector<std::string> &lines);
void write_lines(const char *path, const std::vector<std::string_view> &lines);
std::string read_file(const char *path);

}  // namespace hlmi
